"""CPU restatement of ONE training micro-step / optimizer step of the multitask pre-training
(BASELINE configs #3/#4).  TEST INFRASTRUCTURE ONLY — imported by tests/, oracle/make_golden_train.py
and bench.py's cpu_baseline leg; the product path (streamformer_amd.training) never touches it.

What it follows:
  * step semantics  tools/finetune_tools.py:395-573 (train_one_epoch_multi_task): one task per
    micro-batch, ``loss /= update_freq``, backward every micro-step, optimizer step + zero_grad every
    ``update_freq``-th micro-step (:560-570); lr/wd written into the param groups before the step (:406-410)
  * what is trained  StreamformerForMultiTaskingSigLIP.frozen_spatial (modeling:1471-1484): spatial
    ``attention.attention.qkv`` and ``attention.output.dense`` (weight + bias) frozen, LoRA factors and
    everything else trainable
  * optimizer  torch.optim.AdamW over get_parameter_groups (optim_factory.py:59-104): no weight decay
    for 1-D parameters and names ending in ".bias"; 0-dim (gate, logit_scale, logit_bias) and the
    embedding tables ARE decayed
  * losses  oracle.streamformer_oracle.retrieval_loss / localization_loss (modeling:221-237,
    2238-2282, 2324-2351), each head owning ``logit_scale = log 10`` and ``logit_bias = -2``
    (modeling:1363-1364)
Pinned against the reference's own modules by oracle/make_golden_train.py (fixture F8).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import streamformer_oracle as O

Tensor = torch.Tensor


def is_frozen(name: str, freeze_spatial: bool) -> bool:
    if not freeze_spatial:
        return False
    return (".attention.attention.qkv." in name or ".attention.output.dense." in name) and "temporal_attention" not in name


def no_decay(name: str, p: Tensor) -> bool:
    return p.dim() == 1 or name.endswith(".bias")


def schedule(cfg, B=2, T=4):
    """[(task, pixels, task_input, update_freq)] — deterministic synthetic micro-batches."""
    g = torch.Generator().manual_seed(88)
    D = cfg.hidden_size
    out = []
    lab_emb = torch.randn(5, D, generator=g)
    lab_emb = lab_emb / lab_emb.norm(dim=-1, keepdim=True)
    for i, (task, uf) in enumerate([("retrieval", 1), ("localization", 2), ("localization", 2), ("retrieval", 1)]):
        x = torch.randn(B, T, 3, cfg.image_size, cfg.image_size, generator=g)
        if task == "retrieval":
            ti = {"kind": "retrieval", "text": torch.randn(B, D, generator=g)}
        else:
            ti = {"kind": "localization", "label_emb": lab_emb, "labels": torch.randint(-1, 5, (B, T), generator=g)}
        out.append((task, x, ti, uf))
    return out


class OracleTrainer:
    def __init__(self, sd: Dict[str, Tensor], cfg, head_names: List[str], freeze_spatial: bool = True,
                 lr: float = 1e-3, weight_decay: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8,
                 dtype=torch.float32):
        self.cfg = cfg
        self.sd = {k: v.detach().clone().to(dtype) for k, v in sd.items() if not k.endswith(".mask")}
        self.heads = {n: {"logit_scale": torch.tensor(math.log(10.0), dtype=dtype),
                          "logit_bias": torch.tensor(-2.0, dtype=dtype)} for n in head_names}
        self.named: Dict[str, Tensor] = {}
        for k, v in self.sd.items():
            if not is_frozen(k, freeze_spatial):
                v.requires_grad_(True)
                self.named[k] = v
        for n, h in self.heads.items():
            for k, v in h.items():
                v.requires_grad_(True)
                self.named[f"task_heads.{n}.{k}"] = v
        decay = [p for k, p in self.named.items() if not no_decay(k, p)]
        nodecay = [p for k, p in self.named.items() if no_decay(k, p)]
        self.opt = torch.optim.AdamW([{"params": decay, "weight_decay": weight_decay},
                                      {"params": nodecay, "weight_decay": 0.0}], lr=lr, betas=betas, eps=eps)
        self.micro = 0

    def loss(self, task: str, pixels: Tensor, task_input: dict, drop_path: Optional[Tensor] = None, dropout: Optional[tuple] = None) -> Tensor:
        """``drop_path``: [L, B*N + B*T + B] keep / drop factors of this forward (see O.layer_forward);
        ``dropout``: (seed, hidden_p, attention_p) of this forward's counter-based masks (O.dropout_mask)."""
        out = O.forward_graph(self.sd, self.cfg, pixels, drop_path=drop_path, dropout=dropout)
        h = self.heads[task]
        if task_input["kind"] == "retrieval":
            # other_rank_text: the other ranks' caption features, negatives only (distributed SigLipLoss, modeling:239-297)
            return O.retrieval_loss(out["pooler_output"], task_input["text"], h["logit_scale"], h["logit_bias"],
                                    other_rank_text=task_input.get("other_rank_text"))
        return O.localization_loss(out["pooler_output"], task_input["label_emb"], task_input["labels"],
                                   h["logit_scale"], h["logit_bias"])

    def micro_step(self, task: str, pixels: Tensor, task_input: dict, update_freq: int = 1,
                   lr: Optional[float] = None, weight_decay: Optional[float] = None) -> float:
        loss = self.loss(task, pixels, task_input)
        (loss / update_freq).backward()
        self.micro += 1
        if self.micro % update_freq == 0:
            for g in self.opt.param_groups:
                if lr is not None:
                    g["lr"] = lr
                if weight_decay is not None and g["weight_decay"] > 0:
                    g["weight_decay"] = weight_decay
            self.opt.step()
            self.opt.zero_grad(set_to_none=True)
        return float(loss.detach())

    def grads(self) -> Dict[str, Tensor]:
        return {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in self.named.items()}
