"""Seeded, "de-trivialised" random weights in the reference's state_dict naming.

There are no pretrained weights offline, so benchmarks, tests and golden fixtures all use weights
made here.  The key names and shapes are the ones the reference's
``TimesformerMultiTaskingModelSigLIP.state_dict()`` produces (classes at
``models/modeling_timesformer_siglip.py:300-457, 502-517, 720-763, 808-899, 1128-1139, 1241-1258``),
so the same dict loads into the reference (``load_state_dict``) and into this package
(``from_pretrained`` / ``load_state_dict``).

The reference's own random init (``modeling:1077-1109``, ``:896``, ``:377``) leaves the temporal
gate at 0, time embeddings at 0, every bias at 0 and every LayerNorm at identity — a parity test on
such weights passes with a broken temporal branch.  This generator therefore draws *everything*:
gate != 0, random time/position embeddings, biases, LN affine, and LoRA-B != 0 when LoRA is on.
It is this repo's own routine (plain normal draws from one ``torch.Generator``), not a restatement
of the reference's ``_init_weights``.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict

import torch

from .configuration import LORA_RANK, StreamformerConfig


def make_state_dict(cfg: StreamformerConfig, seed: int = 0, lora: bool | None = None,
                    dtype: torch.dtype = torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """Return an ordered state_dict for ``cfg``.  ``lora`` defaults to ``cfg.add_lora_spatial``."""
    if lora is None:
        lora = bool(cfg.add_lora_spatial)
    g = torch.Generator(device="cpu")
    g.manual_seed(0x5F3759DF ^ (seed * 2654435761 % (1 << 31)))
    D, I, P, C = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels
    N, T = cfg.num_patches, cfg.num_frames
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def normal(*shape, std=1.0, mean=0.0):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * std + mean).to(dtype)

    def linear(prefix, out_f, in_f, std=None, bias=True):
        # fan-in scaled so activations stay O(1) through 12 layers (values reach a few units
        # after the post-LN, like the reference reports for its own init: SURVEY §7).
        sd[prefix + ".weight"] = normal(out_f, in_f, std=std if std is not None else in_f ** -0.5)
        if bias:
            sd[prefix + ".bias"] = normal(out_f, std=0.05)

    def layernorm(prefix):
        sd[prefix + ".weight"] = normal(D, std=0.1, mean=1.0)
        sd[prefix + ".bias"] = normal(D, std=0.05)

    sd["embeddings.position_embeddings"] = normal(1, N, D, std=0.3)
    if cfg.attention_type != "space_only":
        sd["embeddings.time_embeddings"] = normal(1, T, D, std=0.3)
    sd["embeddings.patch_embeddings.projection.weight"] = normal(D, C, P, P, std=(C * P * P) ** -0.5)
    sd["embeddings.patch_embeddings.projection.bias"] = normal(D, std=0.05)

    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{i}."
        # order follows module registration in the reference (modeling:857-896)
        linear(p + "attention.attention.qkv", 3 * D, D, bias=cfg.qkv_bias)
        if lora:
            sd[p + "attention.attention.qkv_lora_a.weight"] = normal(LORA_RANK, D, std=D ** -0.5)
            sd[p + "attention.attention.qkv_lora_b.weight"] = normal(3 * D, LORA_RANK, std=0.05)
        linear(p + "attention.output.dense", D, D)
        if lora:
            sd[p + "attention.output.dense_lora_a.weight"] = normal(LORA_RANK, D, std=D ** -0.5)
            sd[p + "attention.output.dense_lora_b.weight"] = normal(D, LORA_RANK, std=0.05)
        linear(p + "intermediate.dense", I, D)
        linear(p + "output.dense", D, I)
        layernorm(p + "layernorm_before")
        layernorm(p + "layernorm_after")
        if cfg.attention_type == "divided_space_time":
            sd[p + "temporal_attention_gating"] = normal(1, std=0.25, mean=0.6).reshape(())
            layernorm(p + "temporal_layernorm")
            linear(p + "temporal_attention.attention.qkv", 3 * D, D, bias=cfg.qkv_bias)
            if cfg.enable_causal_temporal:  # persistent buffer, unused by forward (modeling:515-517)
                sd[p + "temporal_attention.attention.mask"] = torch.tril(torch.ones(T, T)).to(dtype)
            linear(p + "temporal_attention.output.dense", D, D)
            linear(p + "temporal_dense", D, D)

    layernorm("post_layernorm")
    sd["head.probe"] = normal(1, 1, D, std=1.0)
    sd["head.attention.in_proj_weight"] = normal(3 * D, D, std=D ** -0.5)
    sd["head.attention.in_proj_bias"] = normal(3 * D, std=0.05)
    linear("head.attention.out_proj", D, D)
    layernorm("head.layernorm")
    linear("head.mlp.fc1", I, D)
    linear("head.mlp.fc2", D, I)
    return sd


def state_dict_sha256(sd: Dict[str, torch.Tensor]) -> str:
    """Digest over names, shapes and fp32 bytes (catches RNG drift between boxes)."""
    h = hashlib.sha256()
    for k in sorted(sd):
        t = sd[k].detach().to(torch.float32).contiguous()
        h.update(k.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(t.numpy().tobytes())
    return h.hexdigest()
