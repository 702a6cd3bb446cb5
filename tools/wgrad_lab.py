"""Weight-gradient GEMM lab: what does the 256^2 kernel's K loop run at when the M range is NOT split?

  rocprofv3 --kernel-trace --stats -d out -- python tools/wgrad_lab.py
runs sf_op_wgrad on (a) the step's own shapes with today's split plan and (b) a synthetic [M, 4608] x [M, 3328] problem
(18 x 13 = 234 tiles of 256^2: one workgroup per CU, one round) with SF_WGRAD_NSPLIT=1, i.e. every workgroup walks all
M = 25 088 token rows and writes its tile once.  (b)'s time / 234 tiles is the cost of a tile when weight gradients of
two layers are batched into one launch; compare with (a)'s kernel + reduce times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from streamformer_amd import _native

M = 25088
lib = _native.lib


def run(n1, n2, reps):
    dy = torch.randn(M, n1, device="cuda").bfloat16()
    x = torch.randn(M, n2, device="cuda").bfloat16()
    out = torch.zeros(n1, n2, device="cuda")
    db = torch.zeros(n1, device="cuda")
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = lib.sf_op_wgrad(dy.data_ptr(), n1, x.data_ptr(), n2, M, n1, n2, 1.0, 0, out.data_ptr(), n2, db.data_ptr(), None)
        assert rc == 0
        ts.append((time.perf_counter() - t0) * 1e6)
    ref = (dy[:, :256].float().T @ x[:, :256].float())
    err = float((out[:256, :256] - ref).norm() / ref.norm())
    return min(ts), err


which = os.environ.get("WGRAD_LAB", "all")
if which in ("all", "step"):
    for n1, n2 in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
        print("step shape", n1, n2, "host us (malloc+sync incl.) %.0f rel err %.2e" % run(n1, n2, 5), flush=True)
if which == "all":          # the split override is read once per process: the no-split case runs in its own
    import subprocess
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, WGRAD_LAB="grouped", SF_WGRAD_NSPLIT="1"), check=True)
if which == "grouped":
    assert os.environ.get("SF_WGRAD_NSPLIT") == "1", "run with SF_WGRAD_NSPLIT=1"
    print("234 tiles, no split", "host us %.0f rel err %.2e" % run(4608, 3328, 5), flush=True)
