"""Noise for the device-sharing experiment: one kernel class of the training backward at a time, each looped for
SF_NOISE_SECONDS on the op entry points of the C ABI, with wall-clock stamps — a victim process (tools/head_det.py with SF_DET_TS=1)
running beside it shows which class disturbs the pooling head's forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
import streamformer_amd._native as nat
from streamformer_amd.training import StreamformerTrainer

SEC = float(os.environ.get("SF_NOISE_SECONDS", "22"))
MODES = os.environ.get("SF_NOISE_MODES", "idle,wgrad_lora,wgrad_big,attn_bwd_spatial,attn_bwd_temporal,ln_bwd,head_bwd,layers_bwd,layers_bwd_noside").split(",")
dev = torch.device("cuda:0")
st = nat.current_stream_handle(dev)
M, D, I = 8 * 16 * 196, 768, 3072
g = torch.Generator().manual_seed(1)
rb = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)
dy3, x, dyI = rb(M, 3 * D), rb(M, D), rb(M, I)
u32 = rb(M, 32)
out_a, out_b, out_c = torch.zeros(3 * D, 32, device=dev), torch.zeros(32, D, device=dev), torch.zeros(D, I, device=dev)
qkv_s, o_s, do_s = rb(128, 196, 3 * D), rb(128, 196, D), rb(128, 196, D)
dq_s = torch.empty_like(qkv_s)
qkv_t, o_t, do_t = rb(8, 16, 196, 3 * D), rb(8, 16, 196, D), rb(8, 16, 196, D)
dq_t = torch.empty_like(qkv_t)
xf, dyf, gam = torch.randn(M, D, generator=g).to(dev), torch.randn(M, D, generator=g).to(dev), torch.ones(D, device=dev)
gin, dxf, dg, db = torch.zeros(M, D, device=dev), torch.empty(M, D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)

cfg = sa.siglip_base(add_lora_spatial=True)
tr = StreamformerTrainer(cfg, sa.make_state_dict(cfg, seed=0, lora=True), ["retrieval", "localization"], freeze_spatial=True, device="cuda:0")
xin = torch.randn(8, 16, 3, 224, 224, generator=g).cuda()
lhs, pooler = tr.forward(xin)
gp = torch.randn_like(pooler) * 1e-3
nst = len(tr.stage_ranges)


def stages(first, last):
    nat.check(nat.lib.sf_trainer_backward(tr._h, gp.data_ptr(), None, tr.grads.data_ptr(), first, last, tr._ws.data_ptr(), tr._ws.numel(), tr._stream()))


def one(mode):
    if mode == "idle":
        time.sleep(0.02)
    elif mode == "wgrad_lora":          # sf_wgrad_kernel (128^2 tiles, 64 KB of LDS) + its reduce: the LoRA factor gradients
        nat.check(nat.lib.sf_op_wgrad(dy3.data_ptr(), 3 * D, u32.data_ptr(), 32, M, 3 * D, 32, 1.0, 0, out_a.data_ptr(), 32, 0, st))
        nat.check(nat.lib.sf_op_wgrad(u32.data_ptr(), 32, x.data_ptr(), D, M, 32, D, 1.0, 0, out_b.data_ptr(), D, 0, st))
    elif mode == "wgrad_big":           # sf_wgrad256_kernel (grouped path, 128 KB of LDS)
        nat.check(nat.lib.sf_op_wgrad(x.data_ptr(), D, dyI.data_ptr(), I, M, D, I, 1.0, 0, out_c.data_ptr(), I, 0, st))
    elif mode == "attn_bwd_spatial":
        nat.check(nat.lib.sf_op_attention_bwd(qkv_s.data_ptr(), o_s.data_ptr(), do_s.data_ptr(), dq_s.data_ptr(), 0, 128, 196, 1, 12, 0, st))
    elif mode == "attn_bwd_temporal" or mode.startswith("tbwd:"):      # tbwd:k = lab variant k of the kernel (SF_LIB=lab, SF_TBWD_LAB)
        nat.check(nat.lib.sf_op_attention_bwd(qkv_t.data_ptr(), o_t.data_ptr(), do_t.data_ptr(), dq_t.data_ptr(), 1, 8 * 196, 16, 196, 12, 1, st))
    elif mode == "ln_bwd":
        nat.check(nat.lib.sf_op_layernorm_bwd(xf.data_ptr(), dyf.data_ptr(), gam.data_ptr(), gin.data_ptr(), dxf.data_ptr(), dg.data_ptr(), db.data_ptr(), M, D, 1e-6, st))
    elif mode == "head_bwd":            # stage 0: the pooling head's backward kernels + the head's small GEMMs
        stages(0, 0)
    elif mode in ("layers_bwd", "layers_bwd_noside"):
        stages(0, nst - 1)


for mode in MODES:
    if mode.startswith("tbwd:"): os.environ["SF_TBWD_LAB"] = mode.split(":")[1]
    if mode == "layers_bwd_noside":
        os.environ["SF_TRAIN_SIDE_STREAM"] = "0"
        tr2 = StreamformerTrainer(cfg, sa.make_state_dict(cfg, seed=0, lora=True), ["retrieval", "localization"], freeze_spatial=True, device="cuda:0")
        nat.lib.sf_reload_switches()
        tr = tr2
        lhs, pooler = tr.forward(xin)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    print(f"[noise] {mode} starting", flush=True)
    while time.time() - t0 < SEC:
        for _ in range(8): one(mode)
        torch.cuda.synchronize()
        n += 8
    print(f"[noise] {mode} from {t0:.2f} to {time.time():.2f} ({n} calls)", flush=True)
