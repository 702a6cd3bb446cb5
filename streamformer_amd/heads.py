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


class RetrievalHead:
    def __init__(self, logit_scale: float = math.log(10.0), logit_bias: float = -2.0):
        self.logit_scale = float(logit_scale)
        self.logit_bias = float(logit_bias)

    def loss(self, pooler_output: torch.Tensor, text_features: torch.Tensor, rank: int = 0,
             need_grad: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """pooler_output [B,T,D] (cuda), text_features [W*B, D]: all ranks' caption features, this rank's
        block at rows rank*B.. (the ring exchange of modeling:244-295 delivers exactly these negatives).
        Returns (loss [1], d loss/d pooler [B,T,D], d loss/d (logit_scale, logit_bias) [2])."""
        p, t = _f32(pooler_output), _f32(text_features)
        B, T, D = p.shape
        Bt = t.shape[0]
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        gp = torch.empty_like(p) if need_grad else None
        gs = torch.empty(2, dtype=torch.float32, device=p.device) if need_grad else None
        with torch.cuda.device(p.device):
            nat.check(nat.lib.sf_retrieval_loss(p.data_ptr(), t.data_ptr(), B, T, D, Bt, rank * B, self.logit_scale,
                                                self.logit_bias, loss.data_ptr(), nat.ptr(gp), nat.ptr(gs),
                                                nat.current_stream_handle(p.device)))
        return loss, gp, gs


class LocalizationHead:
    def __init__(self, label_embeddings: torch.Tensor, logit_scale: float = math.log(10.0), logit_bias: float = -2.0):
        self.label_embeddings = label_embeddings     # [L, D], unit-norm means of prompt embeddings (:2211-2223)
        self.logit_scale = float(logit_scale)
        self.logit_bias = float(logit_bias)

    def loss(self, pooler_output: torch.Tensor, labels: torch.Tensor, need_grad: bool = True):
        p = _f32(pooler_output)
        e = _f32(self.label_embeddings.to(p.device))
        lab = labels.to(p.device, torch.int32).contiguous()
        B, T, D = p.shape
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        gp = torch.empty_like(p) if need_grad else None
        gs = torch.empty(2, dtype=torch.float32, device=p.device) if need_grad else None
        with torch.cuda.device(p.device):
            nat.check(nat.lib.sf_localization_loss(p.data_ptr(), e.data_ptr(), lab.data_ptr(), B, T, D, e.shape[0],
                                                   self.logit_scale, self.logit_bias, loss.data_ptr(), nat.ptr(gp),
                                                   nat.ptr(gs), nat.current_stream_handle(p.device)))
        return loss, gp, gs
