import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
for B in (3, 5, 6, 7):
    x = torch.randn(B, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        for _ in range(4): out = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): out = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"fill={os.environ.get('SF_PANEL_MIN_FILL_PCT')} B={B} {dt*1e3:.3f} ms {B*16/dt:.0f} frames/s", flush=True)
