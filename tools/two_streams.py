"""Throughput with 1 vs 2 batches in flight (independent steps pipelined on two HIP streams)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda")
xs = [torch.randn(8, 16, 3, 224, 224).cuda()] * 2
streams = [torch.cuda.Stream() for _ in range(3)]
def run(nstreams, steps=20):
    for i in range(4):
        with torch.cuda.stream(streams[i % nstreams]): m(xs[i % 2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % nstreams]): m(xs[i % 2])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
for r in range(3):
    for n in (1, 2, 3):
        ms = run(n)
        print(f"round {r}: {n} stream(s): {ms:.3f} ms/step -> {128/ms*1e3:.0f} frames/s")
