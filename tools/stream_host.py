"""Host-side cost of one streaming call (time until the call returns, GPU work still queued)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base(num_frames=64)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, seed=0)); m.to("cuda").eval()
x = torch.randn(1, 64, 3, 224, 224).cuda()
cache = m.new_cache(1, 64)
for rep in range(3):
    cache.reset(); torch.cuda.synchronize()
    ts = []
    for t in range(64):
        t0 = time.perf_counter(); m(x[:, t:t + 1], use_cache=True, past_key_values=cache); ts.append(time.perf_counter() - t0)
        if t % 8 == 7: torch.cuda.synchronize()
    ts.sort()
print(f"host time per call: median {1e6*ts[32]:.1f} us, min {1e6*ts[0]:.1f} us")
import cProfile, pstats
cache.reset(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for t in range(64): m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
