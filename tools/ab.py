"""Within-process A/B of runtime switches (env vars read per forward), interleaved rounds.
    python tools/ab.py SF_DISABLE_LN_FOLD [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa

var = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
res = {0: [], 1: []}
for r in range(rounds):
    for flag in (0, 1):
        if flag: os.environ[var] = "1"
        else: os.environ.pop(var, None)
        res[flag].append(run())
for flag in (0, 1):
    v = sorted(res[flag])
    print(f"{var}={'set' if flag else 'unset'}: median {v[len(v)//2]:.3f} ms  min {v[0]:.3f}  all {[round(a,3) for a in res[flag]]}")
