"""LayerNorm backward at the training step's shape (25 088 x 768, fp32 dy through the op entry): us per launch, TB/s.
SF_LN_BWD_BLOCKS caps the grid (one process per setting)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from streamformer_amd import _native as nat
rows, D = 25088, 768
dev = torch.device("cuda", 0)
x, dy, gin = (torch.randn(rows, D, device=dev) for _ in range(3))
gamma = torch.randn(D, device=dev)
dx = torch.empty_like(x); dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
def run():
    nat.check(nat.lib.sf_op_layernorm_bwd(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), gin.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                          rows, D, 1e-6, nat.current_stream_handle(dev)))
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
by = rows * D * 4 * 4
print(f"SF_LN_BWD_BLOCKS={os.environ.get('SF_LN_BWD_BLOCKS', 'default')}: {us:.1f} us per call (kernel + finish), {by / us / 1e6:.2f} TB/s", flush=True)
