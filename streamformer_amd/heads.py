"""Loss heads of the multitask pre-training step (BASELINE config #3), on the HIP library.

Mirrors ``TimesformerVideoRetrievalHead.forward`` + ``SigLipLoss._loss`` (reference
``models/modeling_timesformer_siglip.py:2324-2351, 221-237``) and the training branch of
``TimesformerUniversalLocalizationHead.forward`` (``:2238-2282``).  Each head owns its
``logit_scale = log 10`` / ``logit_bias = -2`` pair (``:1363-1364, 2204-2205, 2300-2301``).
Text features come from a frozen SigLIP text tower in the reference; here they are inputs.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _native as nat


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


def _scalar_on(dev: torch.device, v) -> torch.Tensor:
    """A head parameter as a 1-element fp32 DEVICE tensor (the kernels read it from HBM: no ``.item()``)."""
    if torch.is_tensor(v):
        return v.detach().to(dev, torch.float32).reshape(1)
    return torch.tensor([float(v)], dtype=torch.float32, device=dev)


def _workspace(dev: torch.device, B: int, T: int) -> torch.Tensor:
    return torch.empty(nat.lib.sf_loss_workspace_bytes(B, T), dtype=torch.uint8, device=dev)


class RetrievalHead:
    """``logit_scale`` / ``logit_bias`` may be Python floats or (views of) device tensors — the trainer passes views
    into its flat parameter buffer, so the loss always sees the current values without a host copy."""

    def __init__(self, logit_scale=math.log(10.0), logit_bias=-2.0):
        self.logit_scale = logit_scale
        self.logit_bias = logit_bias

    def loss(self, pooler_output: torch.Tensor, text_features: torch.Tensor, rank: int = 0,
             need_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """pooler_output [B,T,D] (cuda), text_features [W*B, D]: all ranks' caption features, this rank's
        block at rows rank*B.. (the ring exchange of modeling:244-295 delivers exactly these negatives).
        Returns (loss [1], d loss/d pooler [B,T,D], d loss/d (logit_scale, logit_bias) [2])."""
        p, t = _f32(pooler_output), _f32(text_features.to(pooler_output.device))
        B, T, D = p.shape
        Bt = t.shape[0]
        if t.dim() != 2 or t.shape[1] != D:
            raise ValueError(f"text_features must be [rows, {D}], got {tuple(t.shape)}")
        if (rank + 1) * B > Bt:
            raise ValueError(f"rank {rank} with {B} clips needs text rows {rank * B}..{(rank + 1) * B - 1}, the table has {Bt}")
        dev = p.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        gp = torch.empty_like(p) if need_grad else None
        gs = torch.empty(2, dtype=torch.float32, device=dev) if need_grad else None
        ls, lb, ws = _scalar_on(dev, self.logit_scale), _scalar_on(dev, self.logit_bias), _workspace(dev, B, T)
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_retrieval_loss(p.data_ptr(), t.data_ptr(), B, T, D, Bt, rank * B, ls.data_ptr(), lb.data_ptr(),
                                                loss.data_ptr(), nat.ptr(gp), nat.ptr(gs), ws.data_ptr(), ws.numel(),
                                                nat.current_stream_handle(dev)))
        return loss, gp, gs


class LocalizationHead:
    def __init__(self, label_embeddings: torch.Tensor, logit_scale=math.log(10.0), logit_bias=-2.0):
        self.label_embeddings = label_embeddings     # [L, D], unit-norm means of prompt embeddings (:2211-2223)
        self.logit_scale = logit_scale
        self.logit_bias = logit_bias

    def loss(self, pooler_output: torch.Tensor, labels: torch.Tensor, need_grad: bool = True):
        p = _f32(pooler_output)
        dev = p.device
        e = _f32(self.label_embeddings.to(dev))
        lab = labels.to(dev, torch.int32).contiguous()
        B, T, D = p.shape
        if tuple(lab.shape) != (B, T):
            raise ValueError(f"labels must be [{B}, {T}], got {tuple(lab.shape)}")
        if e.shape[0] > 4096:
            raise ValueError(f"{e.shape[0]} label classes: the localization kernel holds at most 4096 per frame row")
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        gp = torch.empty_like(p) if need_grad else None
        gs = torch.empty(2, dtype=torch.float32, device=dev) if need_grad else None
        ls, lb, ws = _scalar_on(dev, self.logit_scale), _scalar_on(dev, self.logit_bias), _workspace(dev, B, T)
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_localization_loss(p.data_ptr(), e.data_ptr(), lab.data_ptr(), B, T, D, e.shape[0],
                                                   ls.data_ptr(), lb.data_ptr(), loss.data_ptr(), nat.ptr(gp),
                                                   nat.ptr(gs), ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev)))
        return loss, gp, gs
