"""Temporal attention backward at the training step's shape (8 clips x 196 patches x 12 heads, 16 frames): 12-wave workgroups that own a CU
(SF_TBWD_OWN_CU=1) against the product's 4-wave workgroups with their exact LDS, alternating, microseconds per launch; outputs compared bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd._native as nat
dev = torch.device("cuda:0"); st = nat.current_stream_handle(dev)
g = torch.Generator().manual_seed(3)
rb = lambda *s: (torch.randn(*s, generator=g) * 0.5).bfloat16().to(dev)
qkv, o, do = rb(8, 16, 196, 2304), rb(8, 16, 196, 768), rb(8, 16, 196, 768)
outs = {}
def run(share, n):
    if share: os.environ.pop("SF_TBWD_OWN_CU", None)
    else: os.environ["SF_TBWD_OWN_CU"] = "1"
    nat.lib.sf_reload_switches()
    dq = torch.zeros_like(qkv)
    call = lambda: nat.check(nat.lib.sf_op_attention_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), dq.data_ptr(), 1, 8 * 196, 16, 196, 12, 1, st))
    for _ in range(20): call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): call()
    b.record(); torch.cuda.synchronize()
    outs[share] = dq
    return a.elapsed_time(b) * 1e3 / n
for rnd in range(3):
    print(f"round {rnd}: owns a CU {run(False, 400):.2f} us   shares {run(True, 400):.2f} us", flush=True)
print("bit-identical:", torch.equal(outs[False], outs[True]), " finite:", bool(torch.isfinite(outs[False].float()).all()))
