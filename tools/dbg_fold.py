import os, sys, torch
sys.path.insert(0, "/root/repo")
import streamformer_amd as sa
from oracle import streamformer_oracle as O
from tests.helpers import small_cfg, frames
for (B, T) in ((1, 1), (2, 2), (2, 5), (1, 16)):
    cfg = small_cfg()
    sd = sa.make_state_dict(cfg, seed=3)
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16"); m.load_state_dict(sd); m.to("cuda")
    x = frames(1, (B, T, 3, 48, 48))
    want = O.forward(sd, cfg, x, output_hidden_states=True)
    out = m(x.cuda(), output_hidden_states=True)
    errs = [float((out.hidden_states[i].cpu() - want["hidden_states"][i]).abs().max()) for i in range(3)]
    print(B, T, "M=", B * T * 9, "hidden-state errs", ["%.3g" % e for e in errs], "lhs %.3g" % float((out.last_hidden_state.cpu() - want["last_hidden_state"]).abs().max()))
