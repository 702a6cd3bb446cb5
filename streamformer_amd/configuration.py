"""StreamformerConfig — the config contract of the encoder's ``from_pretrained()``.

Mirrors the fields and defaults of the reference's
``models/configuration_streamformer.py:90-135`` (``StreamformerConfig(PretrainedConfig)``,
``model_type = "timesformer"`` at ``:88``) so a ``config.json`` written by the reference's
``save_pretrained`` loads unchanged.  It is deliberately NOT a ``transformers.PretrainedConfig``
subclass: the hot path needs a dozen integers, not the HF machinery, and importing ``transformers``
costs seconds on a fresh box.  Unknown keys of a ``config.json`` (HF bookkeeping such as
``architectures``, ``torch_dtype``, ``transformers_version``) are kept verbatim in ``extra`` and
written back by :meth:`to_dict`.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict

_FIELDS = dict(
    image_size=224,
    patch_size=16,
    num_channels=3,
    num_frames=16,
    hidden_size=768,
    num_hidden_layers=12,
    num_attention_heads=12,
    intermediate_size=3072,
    hidden_act="gelu",
    hidden_dropout_prob=0.0,
    attention_probs_dropout_prob=0.0,
    initializer_range=0.02,
    layer_norm_eps=1e-6,
    qkv_bias=True,
    attention_type="divided_space_time",
    drop_path_rate=0,
    clip_config=None,
    enable_causal_temporal=False,
    add_lora_spatial=False,
)

# HF PretrainedConfig attributes the reference's forward reads (modeling:1306-1318).
_HF_DEFAULTS = dict(output_attentions=False, output_hidden_states=False, use_return_dict=True)

LORA_RANK = 32  # hard-coded in the reference: modeling:1280-1281 (`_add_lora(32)`)


class StreamformerConfig:
    model_type = "timesformer"  # configuration_streamformer.py:88

    def __init__(self, **kwargs: Any) -> None:
        for k, v in _FIELDS.items():
            setattr(self, k, kwargs.pop(k, v))
        for k, v in _HF_DEFAULTS.items():
            setattr(self, k, kwargs.pop(k, v))
        if "return_dict" in kwargs:  # HF spelling in config.json
            self.use_return_dict = bool(kwargs.pop("return_dict"))
        kwargs.pop("model_type", None)
        self.extra: Dict[str, Any] = kwargs
        self.validate()

    # -- validation mirrors the errors the reference raises ---------------------------------
    def validate(self) -> None:
        if self.attention_type not in ("divided_space_time", "space_only", "joint_space_time"):
            # modeling:869-874
            raise ValueError("Unknown attention type: {}".format(self.attention_type))
        if self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")

    # -- derived quantities -------------------------------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid  # modeling:321-323

    # -- (de)serialisation ----------------------------------------------------------------------
    def to_dict(self) -> Dict[str, Any]:
        d = {k: getattr(self, k) for k in _FIELDS}
        d["model_type"] = self.model_type
        d.update(self.extra)
        return d

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory: str) -> None:
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            f.write(self.to_json_string())

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "StreamformerConfig":
        return cls(**dict(d))

    @classmethod
    def from_pretrained(cls, name_or_dir: str, **overrides: Any) -> "StreamformerConfig":
        path = name_or_dir
        if os.path.isdir(path):
            path = os.path.join(path, "config.json")
        if not os.path.isfile(path):
            # a hub id, as HF's PretrainedConfig.from_pretrained takes it (configuration_streamformer.py:90-135 inherits it): cache first, then the network
            try:
                from huggingface_hub import hf_hub_download
                path = hf_hub_download(repo_id=name_or_dir, filename="config.json")
            except Exception as e:
                raise OSError(f"{name_or_dir!r} is not a local directory with a config.json and could not be fetched from the hub "
                              f"({type(e).__name__}: {e})") from e
        with open(path) as f:
            d = json.load(f)
        d.update(overrides)
        return cls.from_dict(d)

    def __repr__(self) -> str:
        return "StreamformerConfig " + self.to_json_string()


def siglip_base(**overrides: Any) -> StreamformerConfig:
    """SigLIP-base/16 224px, 16 frames, causal temporal attention: BASELINE.json's config."""
    kw = dict(enable_causal_temporal=True)
    kw.update(overrides)
    return StreamformerConfig(**kw)
