"""First pass vs steady state of the streaming path (config #5): per-frame latency of a FRESH cache's first 64 frames
(lazy set-up, graph capture) and of the second / third pass over the same cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
mode = os.environ.get("SF_MODE", "bf16")
cfg = sa.siglip_base(num_frames=64)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
m.load_state_dict(sa.make_state_dict(cfg, seed=0))
m.to("cuda").eval()
x = torch.randn(1, 64, 3, 224, 224, generator=torch.Generator().manual_seed(64)).cuda()
m(x[:, :2])                      # library / allocator warm-up outside the stream under test
cache = m.new_cache(1, 64)
for rep in range(3):
    cache.reset()
    lat = []
    for t in range(64):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m(x[:, t:t + 1], use_cache=True, past_key_values=cache)
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t0))
    s = sorted(lat)
    print(f"[{mode}] pass {rep}: first frames {' '.join(f'{v:.2f}' for v in lat[:4])} ms | p50 {s[32]:.3f} mean {sum(lat)/64:.3f} max {s[-1]:.3f} ms", flush=True)
