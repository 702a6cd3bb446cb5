"""Where a streamed frame's wall-clock latency goes besides the kernels: per-frame latency (sync - call - sync, as bench.py measures it),
the GPU span between HIP events around the same call, and the host time of the call itself (returns before the GPU is done)."""
import time
import torch
import streamformer_amd as sa

dev = torch.device("cuda:0")
cfg = sa.siglip_base(num_frames=64)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, seed=0))
m.to(dev).eval()
x = torch.randn(1, 64, 3, 224, 224, generator=torch.Generator().manual_seed(64)).to(dev)
frames = [x[:, t:t + 1].contiguous() for t in range(64)]
m(x[:, :2])
cache = m.new_cache(1, 64)
lat, gpu, host = [], [], []
for rep in range(4):
    cache.reset()
    for t in range(64):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        m(frames[t], use_cache=True, past_key_values=cache)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep:
            lat.append(t2 - t0); host.append(t1 - t0); gpu.append(e0.elapsed_time(e1) * 1e-3)
med = lambda v: sorted(v)[len(v) // 2] * 1e6
print(f"per-frame latency p50 {med(lat):.1f} us; GPU span between events p50 {med(gpu):.1f} us; host time of the call p50 {med(host):.1f} us")
# pieces of the host path
import timeit
n = 2000
print("torch.empty x2 us:", 1e6 * timeit.timeit(lambda: (torch.empty(1, 1, 196, 768, device=dev), torch.empty(1, 1, 768, device=dev)), number=n) / n)
print("_sync us:", 1e6 * timeit.timeit(lambda: m._sync(trust_versions=True), number=n) / n)
print("_pos_table us:", 1e6 * timeit.timeit(lambda: m._pos_table(224, 224), number=n) / n)
print("slice us:", 1e6 * timeit.timeit(lambda: x[:, 3:4], number=n) / n)
