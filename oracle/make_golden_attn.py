"""Fixture F10: output_attentions=True of the reference (spatial probabilities per layer, modeling:703-716,
1052-1057) on the small config — pins the oracle's `collect["attentions"]` (build container only).

    python oracle/make_golden_attn.py     # writes tests/golden/f10_attentions.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as G  # noqa: E402
from oracle import streamformer_oracle as O  # noqa: E402
from streamformer_amd.init_weights import make_state_dict, state_dict_sha256  # noqa: E402


def main():
    ref = G.import_reference()
    cfg = G.small_cfg()
    sd = make_state_dict(cfg, seed=10)
    m = G.build_ref(ref, cfg, sd)
    x = G.frames(10, (2, 5, 3, 48, 48))
    with torch.no_grad():
        out = m(x, output_attentions=True)
    att = out.attentions
    assert len(att) == cfg.num_hidden_layers and tuple(att[0].shape) == (2 * 5, cfg.num_attention_heads, 9, 9), att[0].shape
    collect = {}
    mine = O.forward(sd, cfg, x, collect=collect)
    G.check("last_hidden_state", mine["last_hidden_state"], out.last_hidden_state)
    for i, a in enumerate(att):
        G.check(f"attentions[{i}]", collect["attentions"][i], a, tol=2e-6)
    np.savez_compressed(os.path.join(G.OUT, "f10_attentions.npz"), sha256=np.array(state_dict_sha256(sd)),
                        attentions=torch.stack(list(att)).numpy(), last_hidden_state=out.last_hidden_state.numpy())
    print("wrote f10_attentions.npz", os.path.getsize(os.path.join(G.OUT, "f10_attentions.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
