"""Fixture F12: the order in which the REFERENCE encoder yields named_parameters() (build container only).

    python oracle/make_golden_param_order.py      # writes tests/golden/f12_param_order.json

torch.optim.AdamW numbers its state by the position of a parameter in the param groups, and the reference's groups are
filled by walking ``model.named_parameters()`` (optim_factory.py:59-104: first occurrence of "decay" / "no_decay" opens
the group).  A ``checkpoint-*.pth`` (utils.py:608-636) therefore only reloads if the writer enumerates parameters in the
reference's order.  Stored: (name, shape, requires_grad under the LoRA recipe of modeling:1471-1484) for the small config
with and without ``add_lora_spatial``, and the key order of ``state_dict()`` (parameters + mask buffers).

``wrapper_lora`` (round 3): the same for the reference's WRAPPER ``StreamformerForMultiTaskingSigLIP`` (modeling:1356-1447)
with a retrieval and a localization head after ``prepare_for_multi_tasks()`` + ``frozen_spatial()`` — the object the reference's
optimizer is really built over: its own ``logit_scale`` / ``logit_bias`` come FIRST, then ``timesformer.*``, the (frozen) text tower,
then ``task_heads.*``.  The wrapper's constructor fetches the SigLIP text tower + tokenizer from the hub (modeling:1365-1370); offline
they are replaced by a tiny randomly initialised ``SiglipTextModel`` and a stub tokenizer, which changes neither names nor order of
anything outside ``text_encoder.*`` (those rows are dropped: frozen, never in the optimizer).  Data only: no reference source text.
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


def wrapper_rows(ref):
    import torch
    import torch.distributed as dist
    import models.modeling_timesformer_siglip as M
    from transformers import SiglipTextConfig, SiglipTextModel

    class Tok:
        def __call__(self, texts, return_tensors="pt", padding=None, max_length=64, truncation=False):
            class R(dict):
                def to(self, d):
                    return self
            return R(input_ids=torch.randint(0, 100, (len(texts), 8), generator=torch.Generator().manual_seed(len(texts))))
    tcfg = SiglipTextConfig(vocab_size=100, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                            max_position_embeddings=64, projection_size=128, bos_token_id=1, eos_token_id=2, pad_token_id=0)
    M.AutoTokenizer = type("AT", (), {"from_pretrained": staticmethod(lambda *a, **k: Tok())})
    M.SiglipTextModel = type("ST", (), {"from_pretrained": staticmethod(lambda *a, **k: SiglipTextModel(tcfg))})
    if not dist.is_initialized():            # the retrieval head's constructor asks for the rank (modeling:2293-2295)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
        dist.init_process_group("gloo", rank=0, world_size=1)
    cfg = G.small_cfg(add_lora_spatial=True)
    rc = ref.StreamformerConfig(**{k: v for k, v in cfg.to_dict().items() if k != "model_type"})
    m = M.StreamformerForMultiTaskingSigLIP(rc, {"TaskRetrieval": {}, "TaskLocalization": {"label2id": {"synthetic": {"a": 0, "b": 1}}}})
    m.prepare_for_multi_tasks()
    m.frozen_spatial()
    return [[n, list(p.shape), bool(p.requires_grad)] for n, p in m.named_parameters() if not n.startswith("text_encoder.")]


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
    out["wrapper_lora"] = wrapper_rows(ref)
    path = os.path.join(G.OUT, "f12_param_order.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
