"""Bit-reproducibility of the training forward / backward under GPU contention: N passes of forward + localization loss + backward on the
same inputs, run as two concurrent processes on one device (python tools/train_det.py & python tools/train_det.py); every pass is compared
bit for bit with the first one (pooler_output, the loss gradient, every parameter gradient) and the first differing tensors are named."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd.training import StreamformerTrainer

B = int(os.environ.get("SF_DET_B", "8"))
N = int(os.environ.get("SF_DET_N", "12"))
lora = os.environ.get("SF_DET_LORA", "1") == "1"
cfg = sa.siglip_base(add_lora_spatial=lora)
sd = sa.make_state_dict(cfg, seed=0, lora=lora)
tr = StreamformerTrainer(cfg, sd, ["retrieval", "localization"], freeze_spatial=lora, device="cuda:0")
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 16, 3, 224, 224, generator=g).cuda()
lab = torch.randn(20, cfg.hidden_size, generator=g); lab = (lab / lab.norm(dim=-1, keepdim=True)).cuda()
ti = {"kind": "localization", "label_emb": lab, "labels": torch.randint(-1, 20, (B, 16), generator=g).cuda()}
names = tr.parameter_names(trainable_only=True)
ref = None
bad = 0
for it in range(N):
    tr.zero_grad()
    lhs, pooler = tr.forward(x)
    torch.cuda.synchronize()
    lhs0, pool0 = lhs.clone(), pooler.clone()          # as the forward left them
    _, gp, _ = tr.loss_and_grad("localization", pooler, ti)
    tr.backward(gp)
    torch.cuda.synchronize()
    cur = {"pooler": pooler.clone(), "gp": gp.clone(), "grads": tr.grads.clone(), "lhs": lhs.clone(), "pool0": pool0, "lhs0": lhs0}
    if ref is None:
        ref = cur
        continue
    msg = []
    if not torch.equal(cur["pooler"], ref["pooler"]): msg.append(f"pooler max diff {float((cur['pooler'] - ref['pooler']).abs().max()):.3e}")
    if not torch.equal(cur["gp"], ref["gp"]): msg.append("loss gradient differs")
    for k_ in ("pool0", "lhs0", "lhs"):
        if not torch.equal(cur[k_], ref[k_]):
            d_ = (cur[k_] - ref[k_]).abs()
            fr = d_.flatten(2).amax(2) if d_.dim() == 4 else d_.amax(-1)
            msg.append(f"{k_} differs: max {float(d_.max()):.3e}, {int((d_ > 0).sum())} elements, frames {fr.nonzero().tolist()[:8]}")
    if not torch.equal(cur["pooler"], cur["pool0"]): msg.append("pooler_output CHANGED between the end of the forward and the end of the backward")
    if not torch.equal(cur["grads"], ref["grads"]):
        d = (cur["grads"] - ref["grads"]).abs()
        differing = []
        for n in names:
            e = tr.layout[tr.extra_slot.get(n, n)]
            dd = d[e["offset"]: e["offset"] + e["numel"]]
            if float(dd.max()) > 0:
                differing.append((n, float(dd.max())))
                if n.endswith("in_proj_weight"):            # q / k / v row thirds
                    t3 = dd.view(3, -1)
                    differing.append(("q/k/v thirds", [float(t3[i].max()) for i in range(3)], [int((t3[i] > 0).sum()) for i in range(3)]))
                    D_ = cfg.hidden_size
                    nz = (dd.view(3 * D_, D_) > 0).nonzero()
                    rows = sorted(set(int(r) for r in nz[:, 0].tolist()))
                    cols = sorted(set(int(c_) for c_ in nz[:, 1].tolist()))
                    import collections
                    vr = nz[nz[:, 0] >= 2 * D_]
                    hist_i = collections.Counter(int((r - 2 * D_) % 8) for r in vr[:, 0].tolist())
                    hist_c4 = collections.Counter(int(c_ // 4) for c_ in vr[:, 1].tolist())
                    hist_cm = collections.Counter(int(c_ % 4) for c_ in vr[:, 1].tolist())
                    hist_h = collections.Counter(int((r - 2 * D_) // 64) for r in vr[:, 0].tolist())
                    hist_j0 = collections.Counter(int(((r - 2 * D_) % 64) // 8) for r in vr[:, 0].tolist())
                    e0 = tr.layout[tr.extra_slot.get(n, n)]
                    cg = cur["grads"][e0["offset"]: e0["offset"] + e0["numel"]].view(3 * D_, D_)
                    rg = ref["grads"][e0["offset"]: e0["offset"] + e0["numel"]].view(3 * D_, D_)
                    samp = [(int(r), int(c_), float(rg[r, c_]), float(cg[r, c_])) for r, c_ in vr[:6].tolist()]
                    differing.append(("samples (row, col, ref, cur)", samp, "|dWv| max", float(rg[2 * D_:].abs().max())))
                    differing.append(("row%8", dict(hist_i), "col%4", dict(hist_cm), "threads c4", sorted(hist_c4.items())[:40], "heads", dict(hist_h), "j0", dict(hist_j0)))
        msg.append(f"{len(differing)} of {len(names)} gradients differ; first in parameter order: {differing[:4]}; last: {differing[-3:]}")
    if msg:
        bad += 1
        print(f"[pid {os.getpid()}] pass {it}: " + "; ".join(msg), flush=True)
print(f"[pid {os.getpid()}] {N - 1} repeats, {bad} differing", flush=True)
