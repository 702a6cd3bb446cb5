"""Seeded random weights in the HF ``SiglipVisionModel`` naming (input of the SigLIP -> StreamFormer weight surgery,
reference tools/initialize_SigLIP_weights.py:25-264).  TEST INFRASTRUCTURE ONLY (fixture F11): there are no released
SigLIP weights offline, so the surgery is pinned on a random-init vision tower of the same architecture."""
from __future__ import annotations

from collections import OrderedDict

import torch


def siglip_cfg_kwargs(cfg):
    return dict(hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                num_hidden_layers=cfg.num_hidden_layers, image_size=cfg.image_size, patch_size=cfg.patch_size,
                hidden_act=cfg.hidden_act, layer_norm_eps=cfg.layer_norm_eps)


def make_siglip_state_dict(cfg, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(0x51611 ^ seed)
    D, I, N, P, C = cfg.hidden_size, cfg.intermediate_size, cfg.num_patches, cfg.patch_size, cfg.num_channels
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def n(*shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=g) * std + mean

    def lin(name, o, i):
        sd[name + ".weight"] = n(o, i, std=i ** -0.5)
        sd[name + ".bias"] = n(o, std=0.05)

    def ln(name):
        sd[name + ".weight"] = n(D, std=0.1, mean=1.0)
        sd[name + ".bias"] = n(D, std=0.05)

    v = "vision_model."
    sd[v + "embeddings.patch_embedding.weight"] = n(D, C, P, P, std=(C * P * P) ** -0.5)
    sd[v + "embeddings.patch_embedding.bias"] = n(D, std=0.05)
    sd[v + "embeddings.position_embedding.weight"] = n(N, D, std=0.3)
    for i in range(cfg.num_hidden_layers):
        p = v + f"encoder.layers.{i}."
        ln(p + "layer_norm1")
        for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "self_attn." + proj, D, D)
        ln(p + "layer_norm2")
        lin(p + "mlp.fc1", I, D)
        lin(p + "mlp.fc2", D, I)
    ln(v + "post_layernorm")
    sd[v + "head.probe"] = n(1, 1, D)
    sd[v + "head.attention.in_proj_weight"] = n(3 * D, D, std=D ** -0.5)
    sd[v + "head.attention.in_proj_bias"] = n(3 * D, std=0.05)
    lin(v + "head.attention.out_proj", D, D)
    ln(v + "head.layernorm")
    lin(v + "head.mlp.fc1", I, D)
    lin(v + "head.mlp.fc2", D, I)
    return sd
