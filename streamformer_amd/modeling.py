"""Python mirror of the reference encoder module, backed by the HIP library.

Drop-in surface (reference ``models/modeling_timesformer_siglip.py``):

* ``TimesformerMultiTaskingModelSigLIP`` IS a ``torch.nn.Module`` whose parameter tree has the reference's names
  (``embeddings.position_embeddings`` ... ``head.mlp.fc2.bias``, the unused ``...attention.mask`` buffers included), so a
  wrapper can hold it as ``self.timesformer = TimesformerMultiTaskingModelSigLIP(config)`` (``:1362``), ``.to(device)``
  it, ``state_dict()`` / ``load_state_dict()`` it with or without the ``timesformer.`` prefix, and read
  ``next(model.parameters()).device / .dtype`` (``vqa_enc:1574-1581``).  The ``nn.Parameter`` tensors are the fp32
  (or bf16 / fp16 after ``.to(dtype)``) master copy; the library packs its own bf16 operands from them whenever they
  change (tracked through the tensors' version counters).
* ``TimesformerMultiTaskingModelSigLIP.from_pretrained(dir)``           (``:1066-1075``, HF classmethod)
* ``model(pixel_values[B,T,3,H,W], output_attentions=None, output_hidden_states=None,
  return_dict=None)`` -> object with ``last_hidden_state (B,T,N,D)``, ``pooler_output (B,T,D)``,
  ``hidden_states`` (L+1 x ``(B, N*T, D)`` patch-major, ``:1352``), ``attentions``; a tuple when
  ``return_dict=False`` (``:1299-1354``)
* streaming kwargs of the VideoQA copy: ``past_key_values``, ``use_cache``, ``cache_position``, also together with
  ``output_hidden_states=True`` (the tower's call form, ``downstream/VideoQA/.../timesformer_encoder.py:1316-1392, 1536``)
* sub-modules with the reference's call shapes: ``model.embeddings(pixel_values)``, ``model.encoder(h, num_frames=T)``,
  ``model.encoder.layer[i](h, T, output_attentions=False)``, ``model.post_layernorm(x)``, ``model.head(x)``
* ``add_lora_spatial()``, ``frozen_spatial()`` (``:1271-1297``), ``.config``, ``.device``, ``.dtype``.

All arithmetic of the forward runs in ``libstreamformer_hip.so`` through ``_native`` (ctypes);
torch supplies device memory, streams and the tensor container only.  There is no CPU or eager
fallback: without the library or without a GPU the forward raises.
"""
from __future__ import annotations

import os
import weakref
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import _native as nat
from .configuration import LORA_RANK, StreamformerConfig

_ACT_CODES = {"gelu": 0, "gelu_new": 1, "gelu_pytorch_tanh": 1, "relu": 2}
_COMPUTE = {"bf16": nat.SF_COMPUTE_BF16, "bfloat16": nat.SF_COMPUTE_BF16, torch.bfloat16: nat.SF_COMPUTE_BF16,
            "bf16x3": nat.SF_COMPUTE_BF16X3, "fp32": nat.SF_COMPUTE_BF16X3, "float32": nat.SF_COMPUTE_BF16X3,
            torch.float32: nat.SF_COMPUTE_BF16X3}
_TORCH2SF = {torch.uint8: nat.SF_U8, torch.float32: nat.SF_F32, torch.bfloat16: nat.SF_BF16, torch.float16: nat.SF_F16,
             torch.float64: nat.SF_F64}


class ModelOutput(OrderedDict):
    """Attribute + key + index access, like ``transformers.utils.ModelOutput`` (None fields are skipped
    by integer indexing / ``to_tuple``)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return self.to_tuple()[k]
        return super().__getitem__(k)

    def to_tuple(self) -> tuple:
        return tuple(v for v in self.values() if v is not None)


def BaseModelOutputWithPooling(last_hidden_state, pooler_output, hidden_states=None, attentions=None):
    return ModelOutput(last_hidden_state=last_hidden_state, pooler_output=pooler_output,
                       hidden_states=hidden_states, attentions=attentions)


def BaseModelOutputWithPast(last_hidden_state, past_key_values=None, hidden_states=None, attentions=None,
                            pooler_output=None):
    # vqa_enc:1387-1392 drops pooler_output; it is kept here as an extra trailing field.
    return ModelOutput(last_hidden_state=last_hidden_state, past_key_values=past_key_values,
                       hidden_states=hidden_states, attentions=attentions, pooler_output=pooler_output)


class StreamCache:
    """Temporal KV-cache of one stream: the ``past_key_values`` object of the streaming forward.

    Library-owned device memory (``sf_cache``); mirrors the two things the reference uses of HF's
    ``DynamicCache``: ``get_seq_length()`` (vqa_enc:328-331) and being threaded through calls.
    A cache belongs to the packed weights it was created against: when the model re-packs (new weights, another
    compute mode, another device) the cache is invalidated and the next use raises instead of reading freed memory."""

    def __init__(self, model: "TimesformerMultiTaskingModelSigLIP", batch: int, max_frames: int, H: int, W: int,
                 policy: str = "stop"):
        if policy not in ("stop", "slide"):
            raise ValueError("policy must be 'stop' (raise at capacity, like the reference) or 'slide' (sliding window over the last max_frames frames)")
        self._model = weakref.ref(model)
        self._h = nat.C.c_void_p()
        nat.check(nat.lib.sf_cache_create(model._handle, batch, max_frames, H, W, nat.C.byref(self._h)))
        self.batch, self.max_frames, self.H, self.W, self.policy = batch, max_frames, H, W, policy
        if policy == "slide":
            nat.check(nat.lib.sf_cache_set_policy(self._h, 1))
        model._caches.add(self)

    @property
    def valid(self) -> bool:
        return bool(self._h)

    def _require(self):
        if not self._h:
            raise RuntimeError("this past_key_values belongs to weights / a compute mode / a device the model has since "
                               "left (load_state_dict, set_compute_dtype or .to() re-packed it): start a new cache")
        return self._h

    def get_seq_length(self, layer_idx: int = 0) -> int:
        """Frames the cache holds (HF ``DynamicCache`` semantics): never more than ``max_frames`` under the sliding window."""
        return min(self.frames_seen, self.max_frames)

    @property
    def frames_seen(self) -> int:
        """Frames streamed through this cache since its creation / last ``reset()``."""
        return nat.lib.sf_cache_length(self._require())

    def reset(self) -> None:
        nat.check(nat.lib.sf_cache_reset(self._require()))

    @property
    def nbytes(self) -> int:
        return nat.lib.sf_cache_bytes(self._require())

    def _invalidate(self) -> None:
        h, self._h = self._h, None
        if h:
            nat.lib.sf_cache_destroy(h)

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass


def expected_keys(cfg: StreamformerConfig, lora: Optional[bool] = None) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter names and shapes of the reference module."""
    lora = bool(cfg.add_lora_spatial) if lora is None else lora
    D, I, P, C, N, T = (cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels,
                        cfg.num_patches, cfg.num_frames)
    k: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    k["embeddings.position_embeddings"] = (1, N, D)
    k["embeddings.time_embeddings"] = (1, T, D)
    k["embeddings.patch_embeddings.projection.weight"] = (D, C, P, P)
    k["embeddings.patch_embeddings.projection.bias"] = (D,)

    def lin(p, o, i, bias=True):
        k[p + ".weight"] = (o, i)
        if bias:
            k[p + ".bias"] = (o,)

    def ln(p):
        k[p + ".weight"] = (D,)
        k[p + ".bias"] = (D,)

    # order = the reference's named_parameters() (tests/golden/f12_param_order.json, pinned by oracle/make_golden_param_order.py):
    # torch.optim state ids, DDP buckets and checkpoint-*.pth optimizer entries follow it
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{i}."
        k[p + "temporal_attention_gating"] = ()
        lin(p + "attention.attention.qkv", 3 * D, D, cfg.qkv_bias)
        if lora:
            k[p + "attention.attention.qkv_lora_a.weight"] = (LORA_RANK, D)
            k[p + "attention.attention.qkv_lora_b.weight"] = (3 * D, LORA_RANK)
        lin(p + "attention.output.dense", D, D)
        if lora:
            k[p + "attention.output.dense_lora_a.weight"] = (LORA_RANK, D)
            k[p + "attention.output.dense_lora_b.weight"] = (D, LORA_RANK)
        lin(p + "intermediate.dense", I, D)
        lin(p + "output.dense", D, I)
        ln(p + "layernorm_before")
        ln(p + "layernorm_after")
        ln(p + "temporal_layernorm")
        lin(p + "temporal_attention.attention.qkv", 3 * D, D, cfg.qkv_bias)
        lin(p + "temporal_attention.output.dense", D, D)
        lin(p + "temporal_dense", D, D)
    ln("post_layernorm")
    k["head.probe"] = (1, 1, D)
    k["head.attention.in_proj_weight"] = (3 * D, D)
    k["head.attention.in_proj_bias"] = (3 * D,)
    lin("head.attention.out_proj", D, D)
    ln("head.layernorm")
    lin("head.mlp.fc1", I, D)
    lin("head.mlp.fc2", D, I)
    return k


def _lora_keys(cfg: StreamformerConfig, i: int) -> "OrderedDict[str, Tuple[int, ...]]":
    D, p = cfg.hidden_size, f"encoder.layer.{i}."
    return OrderedDict([(p + "attention.attention.qkv_lora_a.weight", (LORA_RANK, D)),
                        (p + "attention.attention.qkv_lora_b.weight", (3 * D, LORA_RANK)),
                        (p + "attention.output.dense_lora_a.weight", (LORA_RANK, D)),
                        (p + "attention.output.dense_lora_b.weight", (D, LORA_RANK))])


def normalize_checkpoint_keys(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """Strip the wrapper prefix and drop non-encoder entries.

    Checkpoints saved from ``StreamformerForMultiTaskingSigLIP`` carry ``timesformer.`` (HF strips it via
    ``base_model_prefix``, modeling:1073; done by hand at downstream/OVIS/mask2former/
    timesformer_maskformer_model.py:123-124), plus ``task_heads.*`` / text-tower entries that
    ``extract_oad_feature.py:79-81`` drops."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("timesformer."):
            k = k[len("timesformer."):]
        elif k.split(".")[0] in ("task_heads", "text_encoder", "text_model", "logit_scale", "logit_bias"):
            continue
        out[k] = v
    return out


# ----------------------------------------------------------------------------------------------------------
# the module tree
# ----------------------------------------------------------------------------------------------------------
class _Node(nn.Module):
    """Parameter container with the reference's attribute path; numeric children index like ``nn.ModuleList``."""

    def _bind(self, root: "TimesformerMultiTaskingModelSigLIP") -> None:
        object.__setattr__(self, "_root_ref", weakref.ref(root))      # not a registered sub-module: no cycle

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop("_root_ref", None)              # weak references do not pickle; the root re-binds its tree in __setstate__
        return d

    @property
    def _root(self) -> "TimesformerMultiTaskingModelSigLIP":
        r = getattr(self, "_root_ref", None)
        r = r() if r is not None else None
        if r is None:
            raise RuntimeError("this sub-module is not attached to a TimesformerMultiTaskingModelSigLIP")
        return r

    def __getitem__(self, i: int) -> nn.Module:
        n = len(self)
        if isinstance(i, slice):
            return [self._modules[str(j)] for j in range(n)][i]
        if i < 0:
            i += n
        return self._modules[str(i)]

    def __len__(self) -> int:
        return sum(1 for k in self._modules if k.isdigit())

    def __bool__(self) -> bool:       # a module is truthy whatever __len__ says
        return True

    def __iter__(self):
        return (self._modules[str(j)] for j in range(len(self)))


def _to_frame_major(x: torch.Tensor, T: int) -> torch.Tensor:
    """reference (B, N*T, D), token = n*T + t  ->  [B, T, N, D] contiguous fp32"""
    B, NT, D = x.shape
    return x.reshape(B, NT // T, T, D).permute(0, 2, 1, 3).contiguous().float()


def _to_patch_major(h: torch.Tensor) -> torch.Tensor:
    B, T, N, D = h.shape
    return h.permute(0, 2, 1, 3).reshape(B, N * T, D)


class TimesformerEmbeddingsSigLIP(_Node):
    """``model.embeddings(pixel_values) -> (B, N*T, D)`` (TimesformerEmbeddingsSigLIP.forward, modeling:413-457)."""

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        m = self._root
        B, T, _, H, W = pixel_values.shape
        ws = m._stage_ws(B, T, H, W)
        dev = m.device
        x = pixel_values.to(dev)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        x = x.contiguous()
        c = m.config
        N = (H // c.patch_size) * (W // c.patch_size)
        h = torch.empty(B, T, N, c.hidden_size, dtype=torch.float32, device=dev)
        pos = m._pos_table(H, W)
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_embed(m._handle, x.data_ptr(), _TORCH2SF[x.dtype], B, T, H, W, h.data_ptr(), nat.ptr(pos),
                                       ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev)))
        return _to_patch_major(h)


class TimesformerLayerSigLIP(_Node):
    """``model.encoder.layer[i](hidden_states, T, output_attentions=False) -> (hidden_states[, attn])``
    (TimesformerLayerSigLIP.forward, modeling:934-1004), patch-major in and out."""

    index: int = 0

    def forward(self, hidden_states: torch.Tensor, T: int, output_attentions: bool = False):
        out = self._root.encoder._run(hidden_states, T, self.index, self.index + 1, output_attentions)
        return (out[0],) + ((out[1][0],) if output_attentions else ())


class TimesformerEncoder(_Node):
    """``model.encoder(hidden_states, num_frames=T, ...)`` (TimesformerEncoder.forward, modeling:1019-1063)."""

    def _run(self, hidden_states: torch.Tensor, T: int, la: int, lb: int, want_attn: bool):
        m = self._root
        B, NT, D = hidden_states.shape
        H, W = m._grid(NT, T)
        ws = m._stage_ws(B, T, H, W)
        dev = m.device
        h = _to_frame_major(hidden_states.to(dev), T)
        N = NT // T
        att = torch.empty(lb - la, B * T, m.config.num_attention_heads, N, N, dtype=torch.float32, device=dev) if want_attn else None
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_layers(m._handle, h.data_ptr(), B, T, H, W, la, lb, nat.ptr(att), ws.data_ptr(), ws.numel(),
                                        nat.current_stream_handle(dev)))
        return _to_patch_major(h), (tuple(att[i] for i in range(lb - la)) if want_attn else None)

    def forward(self, hidden_states: torch.Tensor, num_frames: int, output_attentions: bool = False,
                output_hidden_states: bool = False, return_dict: bool = True):
        L = self._root.config.num_hidden_layers
        hs, atts = ((hidden_states,) if output_hidden_states else None), (() if output_attentions else None)
        x = hidden_states
        if output_hidden_states:                      # layer by layer: every intermediate is an output
            for i in range(L):
                x, a = self._run(x, num_frames, i, i + 1, output_attentions)
                hs = hs + (x,)
                if output_attentions:
                    atts = atts + a
        else:
            x, a = self._run(x, num_frames, 0, L, output_attentions)
            atts = a
        if not return_dict:
            return tuple(v for v in (x, hs, atts) if v is not None)
        return ModelOutput(last_hidden_state=x, hidden_states=hs, attentions=atts)


class _PostLayerNorm(_Node):
    """nn.LayerNorm(D, eps) with the post_layernorm weights (modeling:1251, 1330), any leading shape."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self._root
        m._sync()
        dev = m.device
        xf = x.to(dev, torch.float32).contiguous()
        y = torch.empty_like(xf)
        g = self.weight.detach().to(dev, torch.float32).contiguous()
        b = self.bias.detach().to(dev, torch.float32).contiguous()
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_op_layernorm(xf.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), xf.numel() // xf.shape[-1],
                                              xf.shape[-1], float(m.config.layer_norm_eps), nat.current_stream_handle(dev)))
        return y


class TimesformerSiglipMultiheadAttentionPoolingHead(_Node):
    """TimesformerSiglipMultiheadAttentionPoolingHead.forward (modeling:1141-1154): x (F, N, D) -> (F, D)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self._root
        Fr, N, D = x.shape
        H, W = m._grid(N, 1)
        ws = m._stage_ws(Fr, 1, H, W)
        dev = m.device
        xf = x.to(dev, torch.float32).contiguous()
        pool = torch.empty(Fr, D, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_post_head(m._handle, xf.data_ptr(), Fr, 1, H, W, None, pool.data_ptr(), ws.data_ptr(), ws.numel(),
                                           nat.current_stream_handle(dev)))
        return pool


_NODE_CLASSES = {"embeddings": TimesformerEmbeddingsSigLIP, "encoder": TimesformerEncoder, "post_layernorm": _PostLayerNorm,
                 "head": TimesformerSiglipMultiheadAttentionPoolingHead}


class TimesformerMultiTaskingModelSigLIP(nn.Module):
    """MI355X-native stand-in for the reference class of the same name (modeling:1241-1354)."""

    config_class = StreamformerConfig
    base_model_prefix = "timesformer"      # modeling:1073
    main_input_name = "pixel_values"       # modeling:1074

    def __init__(self, config: StreamformerConfig, compute_dtype: Any = "fp32", device: Any = None,
                 fuse_temporal_proj: bool = True):
        super().__init__()
        if config.attention_type != "divided_space_time":
            # the reference asserts the same wherever StreamFormer touches the encoder (modeling:1272-1274)
            raise NotImplementedError(
                f"attention_type={config.attention_type!r}: only 'divided_space_time' is on the StreamFormer path")
        if config.hidden_act not in _ACT_CODES:
            raise ValueError(f"unsupported hidden_act {config.hidden_act!r}")
        if compute_dtype not in _COMPUTE:
            raise ValueError(f"compute_dtype must be one of 'bf16' (throughput) or 'fp32'/'bf16x3' (accurate), got {compute_dtype!r}")
        self.config = config
        self._compute = _COMPUTE[compute_dtype]
        self._fuse = bool(fuse_temporal_proj)
        self._lora = bool(config.add_lora_spatial)
        self._handle = None
        self._handle_device: Optional[torch.device] = None
        self._packed_token = None           # (device, version sum, ...) the native weights were packed from
        self._force_repack = True
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        self._caches: "weakref.WeakSet[StreamCache]" = weakref.WeakSet()
        self._plist: List[torch.Tensor] = []
        from .processing import TimesformerImageProcessor
        self.image_processor = TimesformerImageProcessor(size=(config.image_size, config.image_size),
                                                         crop_size={"height": config.image_size, "width": config.image_size})
        self._build_tree()
        self._engine = None                 # autograd bridge state (autograd.TrainEngine), built on the first training forward
        # Constructed in eval mode, like a from_pretrained() model: the autograd path (12 GB of saved activations per 8
        # clips) is entered only after an explicit .train(), which every training loop issues (tools/finetune_tools.py:403).
        self.eval()
        if device is not None:
            self.to(device)

    # --------------------------------------------------------------------------------- parameter tree
    def _node_for(self, path: List[str]) -> _Node:
        """Get / create the container module at ``path`` (attribute names of the reference)."""
        node: nn.Module = self
        for depth, name in enumerate(path):
            child = node._modules.get(name)
            if child is None:
                if depth == 0:
                    cls = _NODE_CLASSES.get(name, _Node)
                elif depth == 2 and path[0] == "encoder" and path[1] == "layer":
                    cls = TimesformerLayerSigLIP
                else:
                    cls = _Node
                child = cls()
                child._bind(self)
                if cls is TimesformerLayerSigLIP:
                    child.index = int(name)
                node.add_module(name, child)
            node = child
        return node

    def _default_value(self, key: str, shape: Tuple[int, ...], g: torch.Generator) -> torch.Tensor:
        """Reference-like defaults for a freshly constructed model (modeling:1077-1109, :896, :377, :533-534):
        trunc-normal(0.02) matrices, zero biases, identity LayerNorm, zero gate / time embeddings, LoRA A ~ N(0, 0.02), B = 0."""
        std = float(self.config.initializer_range)
        if key.endswith("layernorm.weight") or key.endswith("layernorm_before.weight") or key.endswith("layernorm_after.weight"):
            return torch.ones(shape)
        if key.endswith(".bias") or key.endswith("gating") or key.endswith("time_embeddings") or key.endswith("_lora_b.weight"):
            return torch.zeros(shape)
        if key == "head.probe":
            return torch.randn(shape, generator=g)
        if key.endswith("_lora_a.weight"):
            return torch.randn(shape, generator=g) * 0.02
        return torch.nn.init.trunc_normal_(torch.empty(shape), std=std, a=-2 * std, b=2 * std, generator=g)

    def _register(self, key: str, value: torch.Tensor, like: Optional[torch.Tensor] = None) -> None:
        *path, leaf = key.split(".")
        node = self._node_for(path)
        if like is not None:
            value = value.to(device=like.device, dtype=like.dtype)
        node.register_parameter(leaf, nn.Parameter(value))

    def _build_tree(self) -> None:
        g = torch.Generator().manual_seed(0)
        c = self.config
        for k, shape in expected_keys(c, self._lora).items():
            self._register(k, self._default_value(k, shape, g))
        for i in range(c.num_hidden_layers):      # persistent buffer of the reference, unused by its forward (modeling:515-517)
            self._node_for(["encoder", "layer", str(i), "temporal_attention", "attention"]).register_buffer(
                "mask", torch.tril(torch.ones(c.num_frames, c.num_frames)))
        self._refresh_plist()

    def _refresh_plist(self) -> None:
        self._named = OrderedDict(self.named_parameters())
        self._plist = list(self._named.values())
        self._force_repack = True

    def _param(self, key: str) -> torch.Tensor:
        return self._named[key]

    # ------------------------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """``nn.Module.load_state_dict`` after key normalisation: wrapper (``timesformer.``) / DDP (``module.``) prefixes
        stripped, task-head and text-tower entries dropped, LoRA factors adopted when the checkpoint has them."""
        sd = normalize_checkpoint_keys(state_dict)
        if any("_lora_" in k for k in sd) and not self._lora:
            self._enable_lora_keys()
        exp = expected_keys(self.config, self._lora)
        for k, shape in exp.items():
            if k in sd and tuple(sd[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shape)}")
        mask_shape = (self.config.num_frames,) * 2
        sd = OrderedDict((k, v) for k, v in sd.items()
                         if not (k.endswith("temporal_attention.attention.mask") and tuple(v.shape) != mask_shape))
        res = super().load_state_dict(sd, strict=False, assign=assign)
        missing = [k for k in res.missing_keys if not k.endswith("temporal_attention.attention.mask")]   # constant buffers
        unexpected = list(res.unexpected_keys)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                               f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        if assign:
            self._refresh_plist()
        self._force_repack = True
        from torch.nn.modules.module import _IncompatibleKeys      # what nn.Module.load_state_dict returns (also unpacks as a pair)
        return _IncompatibleKeys(missing, unexpected)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, config: Optional[StreamformerConfig] = None,
                        compute_dtype: Any = "fp32", device: Any = None, device_map: Any = None,
                        torch_dtype: Any = None, **kwargs):
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            # a hub id ("org/name"), as the reference's HF classmethod takes it (modeling:1066-1075): resolve it to a local snapshot through
            # huggingface_hub (cache first, then the network); every failure — no network, no such repo, library absent — names the fix
            try:
                from huggingface_hub import snapshot_download
                path = snapshot_download(repo_id=path, revision=kwargs.pop("revision", None), cache_dir=kwargs.pop("cache_dir", None),
                                         local_files_only=bool(kwargs.pop("local_files_only", False)),
                                         allow_patterns=["config.json", "*.safetensors", "pytorch_model.bin", "model.bin"])
            except Exception as e:
                raise OSError(f"{pretrained_model_name_or_path!r} is not a local directory and could not be fetched from the hub "
                              f"({type(e).__name__}: {e}).  Download the checkpoint (config.json + model.safetensors or pytorch_model.bin) "
                              "and pass its directory") from e
        cfg_over = {k: kwargs.pop(k) for k in list(kwargs) if k in StreamformerConfig().to_dict()}
        cfg = config or StreamformerConfig.from_pretrained(path, **cfg_over)
        sd = None
        st = os.path.join(path, "model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            for name in ("pytorch_model.bin", "model.bin"):
                p = os.path.join(path, name)
                if os.path.isfile(p):
                    sd = torch.load(p, map_location="cpu", weights_only=True)
                    break
        if sd is None:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {path!r}")
        if isinstance(sd, dict) and "model" in sd and "state_dict" not in sd and not any(k.startswith(("embeddings", "timesformer")) for k in sd):
            sd = sd["model"]     # checkpoint-*.pth layout (utils.py:625-631)
        if any("_lora_" in k for k in sd):
            cfg.add_lora_spatial = True
        model = cls(cfg, compute_dtype=compute_dtype)
        model.load_state_dict(sd, strict=True)
        if isinstance(torch_dtype, torch.dtype):
            model.to(torch_dtype)
        if device is None and isinstance(device_map, (str, int, torch.device)):
            device = device_map
        if device is None and torch.cuda.is_available():
            device = "cuda"
        if device is not None:
            model.to(device)
        model.eval()
        return model

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True) -> None:
        os.makedirs(save_directory, exist_ok=True)
        self.config.add_lora_spatial = self._lora
        self.config.save_pretrained(save_directory)
        sd = {k: v.detach().to("cpu").contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    # ------------------------------------------------------------------------------ LoRA surface
    def _enable_lora_keys(self) -> None:
        if self._lora:
            return
        self._lora = True
        self.config.add_lora_spatial = True
        g = torch.Generator().manual_seed(1)
        like = self._plist[0]
        for i in range(self.config.num_hidden_layers):
            for k, shape in _lora_keys(self.config, i).items():
                self._register(k, self._default_value(k, shape, g), like=like)
        self._refresh_plist()

    def add_lora_spatial(self) -> None:
        """modeling:1271-1282: rank-32 LoRA on every spatial qkv / output.dense; base weights frozen."""
        self._enable_lora_keys()
        for i in range(self.config.num_hidden_layers):
            for n in ("attention.attention.qkv", "attention.output.dense"):
                for leaf in ("weight", "bias"):
                    p = self._named.get(f"encoder.layer.{i}.{n}.{leaf}")
                    if p is not None:
                        p.requires_grad = False
        print("Added LoRA to the following layers: ",
              [f"timesformer.encoder.layer.{i}.attention" for i in range(self.config.num_hidden_layers)])

    def frozen_spatial(self) -> None:
        """modeling:1284-1297 freezes the spatial qkv (its ``attention.dense`` line would raise in the
        reference; only the qkv freeze is observable)."""
        for i in range(self.config.num_hidden_layers):
            for leaf in ("weight", "bias"):
                p = self._named.get(f"encoder.layer.{i}.attention.attention.qkv.{leaf}")
                if p is not None:
                    p.requires_grad = False

    def trainable_parameter_names(self) -> List[str]:
        return [k for k, p in self._named.items() if p.requires_grad]

    # ------------------------------------------------------------------------- module-like surface
    @property
    def device(self) -> torch.device:
        return self._plist[0].device

    @property
    def dtype(self) -> torch.dtype:
        return self._plist[0].dtype

    @property
    def compute_dtype(self) -> str:
        return "bf16" if self._compute == nat.SF_COMPUTE_BF16 else "bf16x3"

    def num_parameters(self) -> int:
        return sum(v.numel() for v in self._plist)

    def _train_engine(self):
        """The flat-buffer training state behind the autograd bridge; rebuilt when device / LoRA / freeze pattern change."""
        from .autograd import TrainEngine
        if self._engine is None or self._engine.signature != TrainEngine._signature(self):
            self._engine = None
            self._engine = TrainEngine(self)
        return self._engine

    def _apply(self, fn, recurse: bool = True):
        out = super()._apply(fn, recurse)
        self._refresh_plist()            # .to() / .cuda() / .half(): new storage -> re-pack, drop device-bound scratch
        self._engine = None
        self._ws.clear()
        self._pos_cache.clear()
        return out

    def set_compute_dtype(self, compute_dtype: Any):
        c = _COMPUTE[compute_dtype]
        if c != self._compute:
            self._compute = c
            self._force_repack = True
            self._ws.clear()
        return self

    def refresh_weights(self) -> None:
        """Force a re-pack of the library's operands on the next forward (needed only after writes the version counters
        cannot see, e.g. through ``param.data``)."""
        self._force_repack = True

    # -------------------------------------------------------------------- copy / pickle (torch.save(model), copy.deepcopy)
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_handle"] = None                   # native state is rebuilt from the parameters on first use
        d["_handle_device"] = None
        d["_packed_token"] = None
        d["_force_repack"] = True
        d["_ws"] = {}
        d["_pos_cache"] = {}
        d["_caches"] = None
        d["_engine"] = None
        d.pop("_named", None)
        d.pop("_plist", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._caches = weakref.WeakSet()
        for m in self.modules():
            if isinstance(m, _Node):
                m._bind(self)
        self._refresh_plist()

    def _release_native(self) -> None:
        for c in list(self._caches):
            c._invalidate()
        h, self._handle = self._handle, None
        if h:
            nat.lib.sf_destroy(h)

    def __del__(self):
        try:
            self._release_native()
        except Exception:
            pass

    # ------------------------------------------------------------------------------ native sync
    def _sf_config(self) -> nat.SfConfig:
        c = self.config
        return nat.SfConfig(c.image_size, c.patch_size, c.num_channels, c.num_frames, c.hidden_size,
                            c.num_hidden_layers, c.num_attention_heads, c.intermediate_size,
                            _ACT_CODES[c.hidden_act], int(bool(c.qkv_bias)), int(bool(c.enable_causal_temporal)),
                            int(self._lora), float(c.layer_norm_eps))

    def _token(self):
        v = 0
        for p in self._plist:
            v += p._version
        ip = self.image_processor
        return (self._plist[0].device, self._compute, v, tuple(ip.image_mean), tuple(ip.image_std), ip.rescale_factor)

    def _sync(self, trust_versions: bool = False) -> None:
        """Create the handle on the current device and (re)upload packed weights when stale.
        ``trust_versions``: skip the scan of the parameters' version counters (~45 us for SigLIP-base) — used for the
        frames of a running stream, whose valid cache already proves the packing it was started on is still the live one."""
        if trust_versions and self._handle and not self._force_repack:
            return
        tok = self._token()
        if not self._force_repack and self._handle and tok == self._packed_token:
            return
        dev = tok[0]
        if dev.type != "cuda":
            raise RuntimeError("the StreamFormer HIP encoder runs on an AMD GPU only: call .to('cuda') first "
                               "(there is no CPU fallback)")
        self._release_native()           # live StreamCaches of the old packing are invalidated with it
        cfg = self._sf_config()
        h = nat.C.c_void_p()
        nat.check(nat.lib.sf_create(nat.C.byref(cfg), dev.index or 0, nat.C.byref(h)))
        self._handle = h
        for k, p in self._named.items():
            t = p.detach().to("cpu").contiguous()
            if t.dtype not in _TORCH2SF or t.dtype == torch.uint8:
                t = t.float()
            shape = (nat.C.c_int64 * max(t.dim(), 1))(*t.shape)
            nat.check(nat.lib.sf_load_tensor(h, k.encode(), t.data_ptr(), _TORCH2SF[t.dtype], shape, t.dim()))
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_finalize_weights(h, self._compute, 1, int(self._fuse)))
        ip = self.image_processor           # uint8 frames: rescale + normalize fused into the patch kernel
        nch = len(ip.image_mean)
        mean = (nat.C.c_float * nch)(*ip.image_mean)
        std = (nat.C.c_float * nch)(*ip.image_std)
        nat.check(nat.lib.sf_set_pixel_normalization(h, mean, std, nch, ip.rescale_factor))
        self._pos_cache.clear()
        self._packed_token = tok
        self._force_repack = False

    def _workspace(self, key: tuple, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _pos_table(self, H: int, W: int) -> Optional[torch.Tensor]:
        """Resized position table for non-native inputs (modeling:380-411): a handful of KFLOP of
        bicubic-antialias resampling done once per resolution with torch on the host."""
        c = self.config
        N = c.num_patches
        if (H // c.patch_size) * (W // c.patch_size) == N and H == W:
            return None
        key = (H, W)
        if key not in self._pos_cache:
            M = int(round(N ** 0.5))
            assert N == M * M
            w0, h0 = W // c.patch_size, H // c.patch_size
            pe = self._param("embeddings.position_embeddings").detach().to("cpu", torch.float32)
            pe = pe.reshape(1, M, M, c.hidden_size).permute(0, 3, 1, 2)
            pe = F.interpolate(pe, size=(w0, h0), mode="bicubic", antialias=True)
            assert (w0, h0) == tuple(pe.shape[-2:])
            self._pos_cache[key] = pe.permute(0, 2, 3, 1).reshape(-1, c.hidden_size).contiguous().to(self.device)
        return self._pos_cache[key]

    # ------------------------------------------------------------------------------------ forward
    def new_cache(self, batch_size: int = 1, max_frames: Optional[int] = None, height: Optional[int] = None,
                  width: Optional[int] = None, policy: str = "stop") -> StreamCache:
        """``policy="slide"``: bounded memory for streams that outlive the cache — past ``max_frames`` every new frame replaces the
        oldest one and attends to the last ``max_frames`` frames (an extension: the reference raises at ``config.num_frames``,
        timesformer_encoder.py:343-348)."""
        self._sync()
        c = self.config
        with torch.cuda.device(self.device):
            return StreamCache(self, batch_size, max_frames or c.num_frames, height or c.image_size, width or c.image_size, policy)

    def forward(self, pixel_values: torch.Tensor, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                past_key_values: Optional[StreamCache] = None, use_cache: bool = False,
                cache_position: Optional[torch.Tensor] = None):
        c = self.config
        output_attentions = c.output_attentions if output_attentions is None else output_attentions
        output_hidden_states = c.output_hidden_states if output_hidden_states is None else output_hidden_states
        return_dict = c.use_return_dict if return_dict is None else return_dict
        if pixel_values.dim() != 5:
            raise ValueError(f"pixel_values must be (B, T, C, H, W), got {tuple(pixel_values.shape)}")
        B, T, C_, H, W = pixel_values.shape
        if C_ != c.num_channels:
            raise ValueError(f"expected {c.num_channels} channels, got {C_}")
        streaming = bool(use_cache) or past_key_values is not None
        if self.training and torch.is_grad_enabled() and not streaming and any(p.requires_grad for p in self._plist):
            # the reference wrapper's training forward (modeling:1506-1511): outputs carry a grad_fn (autograd.py)
            if output_attentions or output_hidden_states:
                raise NotImplementedError("output_attentions / output_hidden_states are not available on the autograd path: "
                                          "call under torch.no_grad() or in eval mode for them")
            if self.device.type != "cuda":
                raise RuntimeError("the StreamFormer HIP encoder runs on an AMD GPU only: call .to('cuda') first")
            from .autograd import encoder_forward_with_grad
            x = pixel_values.to(self.device)
            lhs, pool = encoder_forward_with_grad(self, x)
            if not return_dict:
                return (lhs,)
            return BaseModelOutputWithPooling(lhs, pool)
        self._sync(trust_versions=isinstance(past_key_values, StreamCache) and past_key_values.valid
                   and past_key_values.get_seq_length() > 0)
        dev = self.device
        x = pixel_values.to(dev)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        if x.dtype == torch.uint8 and W % 8:
            x = self.image_processor.normalize(x)       # byte path needs 8-pixel rows; fall back to fp32 frames
        x = x.contiguous()
        N = (H // c.patch_size) * (W // c.patch_size)
        D, L = c.hidden_size, c.num_hidden_layers
        pos = self._pos_table(H, W)
        out_dtype = self.dtype if self.dtype in (torch.bfloat16, torch.float16) else None    # a half-precision module answers in kind

        def cast(t):
            return t if (t is None or out_dtype is None) else t.to(out_dtype)

        with torch.cuda.device(dev):
            stream = nat.current_stream_handle(dev)
            skey = int(stream or 0)          # one workspace per HIP stream: concurrent forwards never share scratch
            lhs = torch.empty(B, T, N, D, dtype=torch.float32, device=dev)
            pool = torch.empty(B, T, D, dtype=torch.float32, device=dev)
            hs = torch.empty(L + 1, B, T, N, D, dtype=torch.float32, device=dev) if output_hidden_states else None
            nbytes = nat.C.c_size_t()
            if streaming:
                cache = past_key_values
                if cache is None:
                    cache = StreamCache(self, B, max(c.num_frames, T), H, W)
                elif not isinstance(cache, StreamCache):
                    raise TypeError("past_key_values must be the StreamCache a previous call returned (or model.new_cache())")
                ch = cache._require()
                if cache._model() is not self:
                    raise ValueError("past_key_values belongs to another model")
                # a sliding-window cache saturates get_seq_length() at its capacity while frames_seen keeps counting: an HF-style
                # caller that derives cache_position from get_seq_length() is as right as one that counts frames
                if cache_position is not None and int(cache_position[0]) not in (cache.frames_seen, cache.get_seq_length()):
                    raise ValueError("cache_position must continue the cache (vqa_enc:1340-1349)")
                if (cache.batch, cache.H, cache.W) != (B, H, W):
                    raise ValueError("past_key_values was created for a different batch size / resolution")
                nat.check(nat.lib.sf_stream_workspace_bytes(self._handle, ch, T, nat.C.byref(nbytes)))
                ws = self._workspace(("s", B, T, H, W, skey), nbytes.value)
                att = None
                if output_attentions:            # spatial probabilities of the new frames (timesformer_encoder.py:494, 557, 720-754)
                    att = torch.empty(L, B * T, c.num_attention_heads, N, N, dtype=torch.float32, device=dev)
                    nat.check(nat.lib.sf_forward_stream_attentions(self._handle, ch, x.data_ptr(), _TORCH2SF[x.dtype], T,
                                                                   lhs.data_ptr(), pool.data_ptr(), nat.ptr(hs), att.data_ptr(),
                                                                   nat.ptr(pos), ws.data_ptr(), ws.numel(), stream))
                else:
                    nat.check(nat.lib.sf_forward_stream(self._handle, ch, x.data_ptr(), _TORCH2SF[x.dtype], T,
                                                        lhs.data_ptr(), pool.data_ptr(), nat.ptr(hs), nat.ptr(pos), ws.data_ptr(),
                                                        ws.numel(), stream))
            else:
                nat.check(nat.lib.sf_workspace_bytes(self._handle, B, T, H, W, nat.C.byref(nbytes)))
                ws = self._workspace(("f", B, T, H, W, skey), nbytes.value)
                att = None
                if output_attentions:
                    # the reference materialises these anyway (modeling:703-705); here only on request
                    att = torch.empty(L, B * T, c.num_attention_heads, N, N, dtype=torch.float32, device=dev)
                    nat.check(nat.lib.sf_forward_attentions(self._handle, x.data_ptr(), _TORCH2SF[x.dtype], B, T, H, W,
                                                            lhs.data_ptr(), pool.data_ptr(), nat.ptr(hs), att.data_ptr(),
                                                            nat.ptr(pos), ws.data_ptr(), ws.numel(), stream))
                else:
                    nat.check(nat.lib.sf_forward(self._handle, x.data_ptr(), _TORCH2SF[x.dtype], B, T, H, W, lhs.data_ptr(),
                                                 pool.data_ptr(), nat.ptr(hs), nat.ptr(pos), ws.data_ptr(), ws.numel(), stream))
        hidden = None
        if hs is not None:
            # the reference hands back patch-major (B, N*T, D) tensors (modeling:1352): permuted views
            hidden = tuple(cast(hs[i]).permute(0, 2, 1, 3).reshape(B, N * T, D) for i in range(L + 1))
        lhs, pool = cast(lhs), cast(pool)
        if streaming:
            attentions = tuple(att[i] for i in range(L)) if att is not None else None
            if not return_dict:
                return (lhs,) + ((hidden,) if hidden is not None else ()) + ((attentions,) if attentions is not None else ()) + (cache,)
            return BaseModelOutputWithPast(lhs, past_key_values=cache if use_cache or past_key_values is not None else None,
                                           hidden_states=hidden, attentions=attentions, pooler_output=pool)
        attentions = tuple(att[i] for i in range(L)) if att is not None else None    # spatial probabilities per layer
        if not return_dict:
            return (lhs,) + ((hidden,) if hidden is not None else ()) + ((attentions,) if attentions is not None else ())
        return BaseModelOutputWithPooling(lhs, pool, hidden_states=hidden, attentions=attentions)

    # -------------------------------------------------------------------------------- sub-modules
    # The reference's users that drive the encoder piecewise (ViT-Adapter interaction blocks:
    # `blk(x, T, output_attentions=False)[0]`, modeling_timesformer_siglip_adapter.py:424-425; the video
    # classifier: embeddings -> encoder -> post_layernorm -> head, downstream/AR/...:121-134) keep their
    # call shapes: tensors cross this surface in the reference's PATCH-major (B, N*T, D) order and are
    # permuted to the library's frame-major layout around each native call.
    def _stage_ws(self, B: int, T: int, H: int, W: int):
        self._sync()
        n = nat.C.c_size_t()
        nat.check(nat.lib.sf_workspace_bytes(self._handle, B, T, H, W, nat.C.byref(n)))
        return self._workspace(("f", B, T, H, W, int(nat.current_stream_handle(self.device) or 0)), n.value)

    def _grid(self, n_tokens: int, T: int) -> Tuple[int, int]:
        """(H, W) of a frame whose patch grid has n_tokens / T cells (square, or the config's aspect)."""
        N = n_tokens // T
        P = self.config.patch_size
        side = int(round(N ** 0.5))
        if side * side != N or N * T != n_tokens:
            raise ValueError(f"{n_tokens} tokens do not form {T} frames of a square patch grid")
        return side * P, side * P

    @torch.no_grad()
    def forward_features(self, pixel_values: torch.Tensor, pooling_method: str = "mean") -> torch.Tensor:
        """StreamformerForMultiTaskingSigLIP.forward_features (modeling:1525-1536): "mean" over frames (the default),
        "no_pooling" = every frame's pooled feature, anything else = the last frame."""
        pooled = self(pixel_values).pooler_output
        if pooling_method == "mean":
            return torch.mean(pooled, 1, False)
        if pooling_method == "no_pooling":
            return pooled
        return pooled[:, -1]


class TimesformerVisionTower(nn.Module):
    """The VideoQA vision tower (vqa_enc:1462-1598): ctor ``(vision_tower, vision_tower_cfg, delay_load)``, ``load_model``,
    the per-stream state machine (``past_key_values`` threaded through calls, outputs concatenated along time, the last
    ``context_length`` frames returned, ``clear_cache()``) and the read-only properties LLaVA reads.

    ``vision_tower`` is a checkpoint directory (``from_pretrained``) or an already constructed encoder;
    ``vision_tower_cfg`` any object with the reference's optional attributes (``streaming_mode``, ``context_length``,
    ``unfreeze_mm_vision_tower``, ``mm_tunable_parts``) — keyword arguments override it.  Memory is bounded: only the
    window that can still be returned is kept (the reference keeps every frame it has ever seen)."""

    def __init__(self, vision_tower, vision_tower_cfg: Any = None, delay_load: bool = False, *, context_length: Optional[int] = None,
                 streaming_mode: Optional[bool] = None, max_frames: Optional[int] = None, compute_dtype: Any = "fp32",
                 cache_policy: str = "stop"):
        super().__init__()
        self.is_loaded = False
        self._compute_dtype = compute_dtype
        self._max_frames = max_frames
        self._cache_policy = cache_policy        # "slide": streams may outlive max_frames / config.num_frames (StreamCache policy)
        if isinstance(vision_tower, nn.Module):
            self.vision_tower_name = getattr(vision_tower, "name_or_path", type(vision_tower).__name__)
            self._adopt(vision_tower)
        else:
            self.vision_tower_name = str(vision_tower)
            # vqa_enc:1472-1492: every branch ends in load_model() ("Force loading checkpoint!!!"), delay_load included
            self.load_model()
        cfg = vision_tower_cfg
        self.streaming_mode = bool(getattr(cfg, "streaming_mode", False) if streaming_mode is None else streaming_mode)
        if self.streaming_mode:
            self.context_length = int(getattr(cfg, "context_length", 16) if context_length is None else context_length)
            self.past_key_values: Optional[StreamCache] = None
            self.hidden_states: Optional[torch.Tensor] = None
        self._spare: Optional[StreamCache] = None

    def _adopt(self, model: TimesformerMultiTaskingModelSigLIP) -> None:
        self.vision_tower = model
        self.config = model.config
        # vqa_enc:1515-1521 builds a processor of the model's size; the encoder already owns exactly that one, and its
        # mean / std are what the fused uint8 path applies
        self.image_processor = model.image_processor
        self.vision_tower.requires_grad_(False)
        self.is_loaded = True

    def load_model(self, device_map=None) -> None:
        if self.is_loaded:
            print("{} is already loaded, `load_model` called again, skipping.".format(self.vision_tower_name))
            return
        self._adopt(TimesformerMultiTaskingModelSigLIP.from_pretrained(self.vision_tower_name, device_map=device_map,
                                                                         compute_dtype=self._compute_dtype))

    def clear_cache(self) -> None:
        """vqa_enc:1528-1530.  The K/V buffers are kept aside and recycled by the next stream instead of re-allocated."""
        if not self.streaming_mode:
            return
        self.hidden_states = None
        if self.past_key_values is not None and self.past_key_values.valid:
            self._spare = self.past_key_values
        self.past_key_values = None

    def _cache_for(self, B: int, H: int, W: int) -> StreamCache:
        pkv = self.past_key_values
        if pkv is not None and not pkv.valid:       # the model re-packed its weights: the old stream is gone
            pkv, self.past_key_values, self.hidden_states = None, None, None
        if pkv is None:
            sp, self._spare = self._spare, None
            if sp is not None and sp.valid and (sp.batch, sp.H, sp.W, sp.policy) == (B, H, W, self._cache_policy) and sp._model() is self.vision_tower:
                sp.reset()
                pkv = sp
            else:
                pkv = self.vision_tower.new_cache(B, self._max_frames or self.config.num_frames, H, W, policy=self._cache_policy)
        return pkv

    def forward(self, images):
        if self.streaming_mode:
            # images: the new frames of one stream, (1, T, C, H, W)   (vqa_enc:1532-1544)
            B, T, _, H, W = images.shape
            cache = self._cache_for(B, H, W)
            x = images if images.dtype == torch.uint8 else images.to(device=self.device, dtype=self.dtype)
            outputs = self.vision_tower(x, use_cache=True, past_key_values=cache, cache_position=None)
            self.past_key_values = outputs.past_key_values
            lhs = outputs.last_hidden_state
            self.hidden_states = lhs if self.hidden_states is None else torch.cat([self.hidden_states, lhs], dim=1)
            self.hidden_states = self.hidden_states[:, -self.context_length:]       # bounded: nothing older is ever returned
            return self.hidden_states.to(images.dtype if images.dtype.is_floating_point else lhs.dtype)
        if type(images) is list:
            # vqa_enc:1545-1555 (whose `image_features.shape` assert on a list cannot run): per item the last encoder
            # hidden state, patch-major (1, N*T, D), for a clip [T, C, H, W]
            feats = []
            for image in images:
                out = self.vision_tower(image.to(device=self.device, dtype=self.dtype).unsqueeze(0), output_hidden_states=True)
                feats.append(out.hidden_states[-1].to(image.dtype))
            return feats
        out = self.vision_tower(images.to(device=self.device, dtype=self.dtype), output_hidden_states=False)
        return out.last_hidden_state.to(images.dtype)        # (B, T, N, D)

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        for p in self.vision_tower.parameters():
            return p.dtype

    @property
    def device(self):
        for p in self.vision_tower.parameters():
            return p.device

    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    @property
    def num_patches_per_side(self):
        return self.config.image_size // self.config.patch_size

    @property
    def image_size(self):
        return self.config.image_size
