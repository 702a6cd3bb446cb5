"""Two half-batches (4 + 4 clips) on two HIP streams vs one batch of 8: does kernel-level interleaving of two
independent forwards beat one lock-step forward?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda")
x = torch.randn(8, 16, 3, 224, 224).cuda()
xa, xb = x[:4].contiguous(), x[4:].contiguous()
sa_, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
def one(n=10):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two(n=10, offset=False):
    def step():
        with torch.cuda.stream(sa_): m(xa)
        with torch.cuda.stream(sb_): m(xb)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print(f"one batch of 8        : {one():.3f} ms")
    print(f"two streams of 4 + 4  : {two():.3f} ms")
