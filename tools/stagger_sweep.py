"""Sweep SF_G256_STAGGER_NS (phase stagger of the 256^2 GEMM) in one process: ms/step of the B=8 forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa

cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0))
m.to("cuda")
x = torch.randn(8, 16, 3, 224, 224).cuda()
def run(n=10):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
vals = [None] + [int(v) for v in sys.argv[1:]] if len(sys.argv) > 1 else [None, 0, 2000, 4000, 6000, 8000, 10000, 13000, 16000, 20000, 26000]
groups = os.environ.get("SGROUPS", "3").split(",")
for gq in groups:
  os.environ["SF_G256_STAGGER_GROUPS"] = gq
  for rep in range(1):
    for v in vals:
        key = os.environ.get("SWEEP_KEY", "SF_G256_STAGGER_NS")
        if v is None: os.environ.pop(key, None)
        else: os.environ[key] = str(v)
        sa._native.lib.sf_reload_switches()          # the library reads its switch table once; re-read after every change
        print(f"groups {gq} stagger {'auto' if v is None else v:>6}: {run():.3f} ms", flush=True)
