"""Forward time at a given batch / clip length: python tools/fwd_time.py B T  (SF_MODE=bf16|fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
B, T = int(sys.argv[1]), int(sys.argv[2])
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(B, T, 3, 224, 224).cuda()
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"B={B} T={T} (M={B*T*196}): {dt*1e3:.3f} ms = {B*T/dt:.0f} frames/s")
