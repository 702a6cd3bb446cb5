"""A ViT-L/16-shaped StreamFormer encoder (D = 1024, 16 heads, I = 4096, L = 24: outside every SigLIP-base-only fast path — no N = 768 panel tiles,
no plane-form residual stream) on the generic kernels: frames/s of the 8-clip forward in both compute modes, error against the CPU oracle on one clip,
and (under rocprofv3 --kernel-trace) its kernel table.   python tools/vitl_forward.py [clips] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd.configuration import StreamformerConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = StreamformerConfig(image_size=224, patch_size=16, num_frames=16, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                         intermediate_size=4096, enable_causal_temporal=True)
sd = sa.make_state_dict(cfg, seed=0)
x = torch.randn(B, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
gf_per_clip = 24 * 2 * 3136 * (2 * 1024 * 3072 + 2 * 1024 * 1024 + 1024 * 1024 + 2 * 1024 * 4096) / 1e9      # the eight Linear layers of a block (fused temporal pair counted as two)
want = None
if os.environ.get("SF_VITL_ORACLE", "1") == "1":
    from oracle import streamformer_oracle as O
    x1 = torch.randn(1, 16, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    want = O.forward(sd, cfg, x1)
for mode in ("bf16", "fp32"):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    for _ in range(3):
        out = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    line = f"ViT-L/16-shaped encoder, {B} clips x 16 x 224^2, {mode}: {1e3 * dt:.2f} ms/step = {B * 16 / dt:.0f} frames/s ({B * gf_per_clip / dt / 1e3:.0f} TFLOP/s of Linear-layer work)"
    if want is not None:
        o1 = m(x1.cuda())
        line += (f"; one clip vs the CPU oracle: max-abs last_hidden_state {float((o1.last_hidden_state.cpu() - want['last_hidden_state']).abs().max()):.3e}, "
                 f"pooler_output {float((o1.pooler_output.cpu() - want['pooler_output']).abs().max()):.3e}")
    print(line, flush=True)
    del m
    torch.cuda.empty_cache()
