"""Kernel sequence of ONE forward at the BASELINE batch:  rocprofv3 --kernel-trace ... -- python tools/forward_trace.py ; then
python tools/forward_trace.py <results.db> prints the launches of the last forward in order with durations."""
import os, sys
if len(sys.argv) > 1:
    import sqlite3
    rows = list(sqlite3.connect(sys.argv[1]).execute("select name, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "sf_gather_rows" in r[0]]
    last = rows[idx[-1]:]
    print(f"{len(last)} launches, span {(last[-1][2]-last[0][1])/1e3:.1f} us, busy {sum(e-s for _,s,e in last)/1e3:.1f} us")
    for name, s, e in last[:int(os.environ.get('N', '60'))]:
        print(f"{(e-s)/1e3:8.2f} us  {name.split('(')[0][:90]}")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224).cuda()
for _ in range(4):
    m(x)
torch.cuda.synchronize()
