"""Fixture F12: the order in which the REFERENCE encoder yields named_parameters() (build container only).

    python oracle/make_golden_param_order.py      # writes tests/golden/f12_param_order.json

torch.optim.AdamW numbers its state by the position of a parameter in the param groups, and the reference's groups are
filled by walking ``model.named_parameters()`` (optim_factory.py:59-104: first occurrence of "decay" / "no_decay" opens
the group).  A ``checkpoint-*.pth`` (utils.py:608-636) therefore only reloads if the writer enumerates parameters in the
reference's order.  Stored: (name, shape, requires_grad under the LoRA recipe of modeling:1471-1484) for the small config
with and without ``add_lora_spatial``, and the key order of ``state_dict()`` (parameters + mask buffers).  Data only: no reference source text.
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_golden as G  # noqa: E402
from oracle import train_oracle as TO  # noqa: E402
from streamformer_amd.init_weights import make_state_dict  # noqa: E402


def main():
    ref = G.import_reference()
    out = {}
    for lora in (False, True):
        cfg = G.small_cfg(add_lora_spatial=lora)
        m = G.build_ref(ref, cfg, make_state_dict(cfg, seed=8, lora=lora))
        rows = []
        for name, p in m.named_parameters():
            rows.append([name, list(p.shape), not TO.is_frozen(name, lora)])
        out["lora" if lora else "plain"] = rows
        out[("lora" if lora else "plain") + "_state_dict_keys"] = list(m.state_dict().keys())
    path = os.path.join(G.OUT, "f12_param_order.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
