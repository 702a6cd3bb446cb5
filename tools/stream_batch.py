"""Streaming step with several streams per call (one new 224^2 frame for each of S streams, 64-frame caches):
latency per call and frames/s across the streams.  SF_MODE=bf16|fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base(num_frames=64)
mode = os.environ.get("SF_MODE", "bf16")
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
m.load_state_dict(sa.make_state_dict(cfg, seed=0)); m.to("cuda").eval()
for S in (1, 2, 4, 8, 16):
    x = torch.randn(S, 64, 3, 224, 224).cuda()
    cache = m.new_cache(S, 64)
    lat = []
    with torch.no_grad():
        for rep in range(3):
            cache.reset(); torch.cuda.synchronize()
            for t in range(64):
                t0 = time.perf_counter()
                out = m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
                torch.cuda.synchronize()
                if rep: lat.append(time.perf_counter() - t0)
    lat.sort()
    p50 = lat[len(lat) // 2]
    print(f"{mode} streams={S}: p50 {p50*1e3:.3f} ms per call = {S/p50:.0f} frames/s")
    del cache, x
