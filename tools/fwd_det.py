"""Bit-reproducibility of the inference forward under GPU contention (two concurrent processes on one device): N forwards of the same
8-clip batch, every output compared bit for bit with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
mode = os.environ.get("SF_MODE", "bf16")
B = int(os.environ.get("SF_DET_B", "8")); N = int(os.environ.get("SF_DET_N", "40"))
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=mode)
m.load_state_dict(sa.make_state_dict(cfg, seed=0)); m.to("cuda").eval()
x = torch.randn(B, 16, 3, 224, 224, generator=torch.Generator().manual_seed(3)).cuda()
ref = None; bad = 0
for it in range(N):
    o = m(x); torch.cuda.synchronize()
    cur = (o.last_hidden_state.clone(), o.pooler_output.clone())
    if ref is None: ref = cur; continue
    if not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])):
        bad += 1
        d0 = (cur[0] - ref[0]).abs(); d1 = (cur[1] - ref[1]).abs()
        print(f"[pid {os.getpid()} {mode}] forward {it}: lhs max diff {float(d0.max()):.3e} in {int((d0 > 0).sum())} elements (clips {sorted(set(d0.flatten(1).amax(1).nonzero().flatten().tolist()))}); "
              f"pooler {float(d1.max()):.3e} in {int((d1 > 0).sum())}", flush=True)
print(f"[pid {os.getpid()} {mode}] {N - 1} repeats, {bad} differing", flush=True)
