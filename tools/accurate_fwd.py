"""Forward time of the fp32-accurate (bf16x3) mode at the BASELINE batch, and its error against the bf16 mode's
inputs (random weights, seed 0): `python tools/accurate_fwd.py [steps]`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    for _ in range(3): out = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): out = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"accurate forward: {dt*1e3:.2f} ms/step = {8*16/dt:.0f} frames/s; checksum {out.last_hidden_state.double().abs().mean().item():.9f}")
