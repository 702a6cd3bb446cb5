"""frames/s vs clips per batch (bf16 mode, SigLIP-base, 16 x 224^2 clips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda")
for B in (1, 2, 4, 8, 16, 32):
    x = torch.randn(B, 16, 3, 224, 224).cuda()
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = max(3, 40 // B)
    for _ in range(n): m(x)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B:2d}: {ms:8.3f} ms/step  {B*16/ms*1e3:9.0f} frames/s")
