"""Checkpoint formats either side of the encoder.

* ``siglip_vision_to_streamformer``: the key mapping of the reference's SigLIP -> StreamFormer weight
  surgery (``tools/initialize_SigLIP_weights.py:25-264``): q/k/v projections are concatenated into the
  fused ``qkv`` Linear, ``layer_norm1/2`` become ``layernorm_before/after``, ``mlp.fc1/fc2`` become
  ``intermediate/output.dense``, the attention-pooling ``head.*`` is taken verbatim; the temporal
  branch has no SigLIP counterpart and is drawn N(0, 0.02) with zero biases, identity LayerNorm and a
  zero gate (``:229-240``), so the converted model reproduces per-frame SigLIP until it is trained.
* ``load_training_checkpoint``: the ``checkpoint-*.pth`` dict the reference trainer writes
  (``utils.py:625-631``: ``{"model", "optimizer", "epoch", "scaler", "args"}``) -> encoder state_dict.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

from .configuration import StreamformerConfig
from .modeling import expected_keys, normalize_checkpoint_keys


def siglip_vision_to_streamformer(siglip_sd: Dict[str, torch.Tensor], cfg: StreamformerConfig,
                                  seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """``siglip_sd``: state_dict of HF ``SiglipModel`` / ``SiglipVisionModel`` (keys with or without the
    ``vision_model.`` prefix).  Returns a full StreamFormer encoder state_dict for ``cfg``."""
    src = {}
    for k, v in siglip_sd.items():
        if k.startswith("text_model.") or k in ("logit_scale", "logit_bias"):
            continue
        src[k[len("vision_model."):] if k.startswith("vision_model.") else k] = v.detach().float()
    D = cfg.hidden_size
    g = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    out["embeddings.patch_embeddings.projection.weight"] = src["embeddings.patch_embedding.weight"]
    out["embeddings.patch_embeddings.projection.bias"] = src["embeddings.patch_embedding.bias"]
    out["embeddings.position_embeddings"] = src["embeddings.position_embedding.weight"].reshape(1, -1, D)
    out["embeddings.time_embeddings"] = torch.zeros(1, cfg.num_frames, D)           # modeling:377
    for i in range(cfg.num_hidden_layers):
        s, d = f"encoder.layers.{i}.", f"encoder.layer.{i}."
        for wb in ("weight", "bias"):
            out[d + f"attention.attention.qkv.{wb}"] = torch.cat(
                [src[s + f"self_attn.{p}_proj.{wb}"] for p in ("q", "k", "v")], dim=0)
            out[d + f"attention.output.dense.{wb}"] = src[s + f"self_attn.out_proj.{wb}"]
            out[d + f"layernorm_before.{wb}"] = src[s + f"layer_norm1.{wb}"]
            out[d + f"layernorm_after.{wb}"] = src[s + f"layer_norm2.{wb}"]
            out[d + f"intermediate.dense.{wb}"] = src[s + f"mlp.fc1.{wb}"]
            out[d + f"output.dense.{wb}"] = src[s + f"mlp.fc2.{wb}"]
        # temporal branch: fresh parameters
        out[d + "temporal_attention_gating"] = torch.zeros(())
        out[d + "temporal_layernorm.weight"] = torch.ones(D)
        out[d + "temporal_layernorm.bias"] = torch.zeros(D)
        for name, o in (("temporal_attention.attention.qkv", 3 * D), ("temporal_attention.output.dense", D),
                        ("temporal_dense", D)):
            out[d + name + ".weight"] = torch.randn(o, D, generator=g) * 0.02
            out[d + name + ".bias"] = torch.zeros(o)
    for wb in ("weight", "bias"):
        out[f"post_layernorm.{wb}"] = src[f"post_layernorm.{wb}"]
    for k, v in src.items():
        if k.startswith("head."):
            out[k] = v
    exp = expected_keys(cfg, lora=False)
    missing = [k for k in exp if k not in out]
    bad = [k for k in exp if k in out and tuple(out[k].shape) != tuple(exp[k])]
    if missing or bad:
        raise ValueError(f"SigLIP checkpoint does not match the config: missing {missing[:5]}, shape mismatch {bad[:5]}")
    return OrderedDict((k, out[k].contiguous()) for k in exp)


def load_training_checkpoint(path: str) -> "OrderedDict[str, torch.Tensor]":
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    return normalize_checkpoint_keys(sd)


def dump_for_c_host(config, state_dict: Dict[str, torch.Tensor], path: str) -> None:
    """Write the weight file ``examples/host_forward.cpp`` reads (config ints, eps, then name / shape / fp32 data per
    tensor): the hand-off format for a host that binds the C ABI without Python."""
    import struct
    acts = {"gelu": 0, "gelu_new": 1, "gelu_pytorch_tanh": 1, "relu": 2}
    c = config
    with open(path, "wb") as f:
        f.write(struct.pack("<12i", c.image_size, c.patch_size, c.num_channels, c.num_frames, c.hidden_size, c.num_hidden_layers,
                            c.num_attention_heads, c.intermediate_size, acts[c.hidden_act], int(c.qkv_bias),
                            int(c.enable_causal_temporal), int(c.add_lora_spatial)))
        f.write(struct.pack("<f", float(c.layer_norm_eps)))
        items = [(k, v) for k, v in state_dict.items() if not k.endswith(".mask")]
        f.write(struct.pack("<i", len(items)))
        for k, v in items:
            t = v.detach().to(torch.float32).contiguous().cpu()
            name = k.encode()
            f.write(struct.pack("<i", len(name)))
            f.write(name)
            f.write(struct.pack("<i", t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape) if t.dim() else b"")
            f.write(t.numpy().tobytes())
