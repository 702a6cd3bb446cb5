import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda")
x = torch.randn(8, 16, 3, 224, 224).cuda()
def run(n=10):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for label, env in (("none", {"SF_G256_STAGGER_NS": "0"}), ("both", {}), ("up only", {"SF_G256_STAGGER_ONLY": "2"}), ("qkv only", {"SF_G256_STAGGER_ONLY": "1"}),
                       ("up only 4us", {"SF_G256_STAGGER_ONLY": "2", "SF_G256_STAGGER_NS": "4000"}), ("qkv only 3us", {"SF_G256_STAGGER_ONLY": "1", "SF_G256_STAGGER_NS": "3000"}),
                       ("qkv only 2us", {"SF_G256_STAGGER_ONLY": "1", "SF_G256_STAGGER_NS": "2000"})):
        for k in ("SF_G256_STAGGER_NS", "SF_G256_STAGGER_ONLY"): os.environ.pop(k, None)
        os.environ.update(env)
        print(f"{label:>14}: {run():.3f} ms", flush=True)
