"""Bit-reproducibility of the TRAINING forward under GPU contention: N forwards of the same batch (two concurrent processes on one
device); last_hidden_state and pooler_output compared bit for bit with the first, differing frames named."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd.training import StreamformerTrainer
B = int(os.environ.get("SF_DET_B", "8")); N = int(os.environ.get("SF_DET_N", "150"))
lora = os.environ.get("SF_DET_LORA", "1") == "1"
cfg = sa.siglip_base(add_lora_spatial=lora)
tr = StreamformerTrainer(cfg, sa.make_state_dict(cfg, seed=0, lora=lora), ["retrieval", "localization"], freeze_spatial=lora, device="cuda:0")
x = torch.randn(B, 16, 3, 224, 224, generator=torch.Generator().manual_seed(5)).cuda()
ref = None; bad = 0
for it in range(N):
    lhs, pooler = tr.forward(x)
    torch.cuda.synchronize()
    cur = (None if lhs is None else lhs.clone(), pooler.clone())
    if ref is None: ref = cur; continue
    msg = []
    if cur[0] is not None and not torch.equal(cur[0], ref[0]):
        d = (cur[0] - ref[0]).abs().flatten(2).amax(2)            # [B, T]
        msg.append(f"lhs max diff {float(d.max()):.3e}, frames {d.nonzero().tolist()[:6]} ({int((d > 0).sum())} frames)")
    if not torch.equal(cur[1], ref[1]):
        d = (cur[1] - ref[1]).abs().amax(-1)
        cols = ((cur[1] - ref[1]).abs() > 0).nonzero()
        msg.append(f"pooler max diff {float(d.max()):.3e}, frames {d.nonzero().tolist()[:6]} ({int((d > 0).sum())} frames), columns {sorted(set(cols[:, -1].tolist()))[:24]} ({len(set(cols[:, -1].tolist()))})")
    if msg:
        bad += 1
        print(f"[pid {os.getpid()}] forward {it}: " + "; ".join(msg), flush=True)
print(f"[pid {os.getpid()}] {N - 1} repeats, {bad} differing", flush=True)
