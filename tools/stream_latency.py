"""Config #5 probe: 64-frame online clip, one frame per call, B=1 (num_frames=64 SigLIP-base)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa

cfg = sa.siglip_base(num_frames=64)
sd = sa.make_state_dict(cfg, seed=0)
x = torch.randn(1, 64, 3, 224, 224).cuda()
for mode in ("bf16", "fp32"):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda")
    cache = m.new_cache(1, 64)
    lat = []
    for rep in range(3):
        cache.reset()
        for t in range(64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
            torch.cuda.synchronize()
            if rep:
                lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"[{mode}] per-frame latency p50 {1e3*lat[len(lat)//2]:.3f} ms  p99 {1e3*lat[int(len(lat)*0.99)]:.3f} ms  "
          f"mean {1e3*sum(lat)/len(lat):.3f} ms -> {len(lat)/sum(lat):.1f} frames/s; cache {cache.nbytes/1e6:.0f} MB")
