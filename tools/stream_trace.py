"""Timeline of the bf16 streaming path for rocprofv3 --kernel-trace: 64 single-frame calls x REPS (config #5).
   rocprofv3 --kernel-trace -d gpurun_out/strace -o t -- python tools/stream_trace.py ; python tools/stream_timeline.py gpurun_out/strace/.../t_results.db"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa

mode = os.environ.get("SF_MODE", "bf16")
reps = int(os.environ.get("SF_REPS", "3"))
cfg = sa.siglip_base(num_frames=64)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
m.load_state_dict(sa.make_state_dict(cfg, seed=0))
m.to("cuda").eval()
x = torch.randn(1, 64, 3, 224, 224, generator=torch.Generator().manual_seed(64)).cuda()
cache = m.new_cache(1, 64)
lat = []
for rep in range(reps):
    cache.reset()
    for t in range(64):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
        torch.cuda.synchronize()
        if rep:
            lat.append(time.perf_counter() - t0)
lat.sort()
print(f"[{mode}] p50 {1e3*lat[len(lat)//2]:.3f} ms p99 {1e3*lat[int(len(lat)*0.99)]:.3f} mean {1e3*sum(lat)/len(lat):.3f} ms")
