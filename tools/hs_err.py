import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import test_hip_parity as t
import streamformer_amd as sa
g = t.load_npz(os.path.join(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"), "f1_small.npz"))
cfg = t.small_cfg(); sd = t.make_state_dict(cfg, seed=1)
for fuse in (True, False):
    m = t.build(sa, cfg, sd, "bf16", fuse)
    for T in (1, 5, 16):
        x = t.frames(100 + T, (2, T, 3, 48, 48))
        out = m(x.cuda(), output_hidden_states=True)
        hs = torch.stack([h.cpu() for h in out.hidden_states])
        ref = torch.as_tensor(g[f"T{T}_hidden_states"])
        d = (hs - ref).abs()
        print(fuse, T, "maxabs", float(d.max()), "per-layer", [round(float(d[i].max()), 4) for i in range(d.shape[0])], "max|ref|", [round(float(ref[i].abs().max()), 2) for i in range(ref.shape[0])])
