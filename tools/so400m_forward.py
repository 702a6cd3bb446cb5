"""A SigLIP-so400m-shaped StreamFormer encoder (hidden 1152, 16 heads of 72, intermediate 4304, 27 layers, 14 x 14 patches on 224 x 224 = 256 tokens
per frame) on the generic-width path of round 6: frames/s of the forward in both compute modes, error against the CPU oracle on one 4-frame clip,
and (under rocprofv3 --kernel-trace) its kernel table.   python tools/so400m_forward.py [clips] [steps] [layers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd.configuration import StreamformerConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = int(sys.argv[3]) if len(sys.argv) > 3 else 27
D, I, N = 1152, 4304, 256
cfg = StreamformerConfig(image_size=224, patch_size=14, num_frames=16, hidden_size=D, num_hidden_layers=L, num_attention_heads=16,
                         intermediate_size=I, enable_causal_temporal=True)
sd = sa.make_state_dict(cfg, seed=0)
x = torch.randn(B, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
gf_per_clip = L * 2 * 16 * N * (2 * D * 3 * D + 3 * D * D + 2 * D * I) / 1e9      # the eight Linear layers of a block
want = None
if os.environ.get("SF_SO400M_ORACLE", "1") == "1":
    from oracle import streamformer_oracle as O
    x1 = torch.randn(1, 4, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    want = O.forward(sd, cfg, x1)
for mode in ("bf16", "fp32"):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda").eval()
    for _ in range(2):
        out = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    line = f"so400m-shaped encoder ({L} layers), {B} clips x 16 x 224^2, {mode}: {1e3 * dt:.2f} ms/step = {B * 16 / dt:.0f} frames/s ({B * gf_per_clip / dt / 1e3:.0f} TFLOP/s of Linear-layer work)"
    if want is not None:
        o1 = m(x1.cuda())
        line += (f"; one 4-frame clip vs the CPU oracle: max-abs last_hidden_state {float((o1.last_hidden_state.cpu() - want['last_hidden_state']).abs().max()):.3e}, "
                 f"pooler_output {float((o1.pooler_output.cpu() - want['pooler_output']).abs().max()):.3e}")
    print(line, flush=True)
    del m
