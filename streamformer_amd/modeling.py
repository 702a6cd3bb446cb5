"""Python mirror of the reference encoder module, backed by the HIP library.

Drop-in surface (reference ``models/modeling_timesformer_siglip.py``):

* ``TimesformerMultiTaskingModelSigLIP.from_pretrained(dir)``           (``:1066-1075``, HF classmethod)
* ``model(pixel_values[B,T,3,H,W], output_attentions=None, output_hidden_states=None,
  return_dict=None)`` -> object with ``last_hidden_state (B,T,N,D)``, ``pooler_output (B,T,D)``,
  ``hidden_states`` (L+1 x ``(B, N*T, D)`` patch-major, ``:1352``), ``attentions``; a tuple when
  ``return_dict=False`` (``:1299-1354``)
* streaming kwargs of the VideoQA copy: ``past_key_values``, ``use_cache``, ``cache_position``
  (``downstream/VideoQA/.../timesformer_encoder.py:1316-1392``)
* ``add_lora_spatial()``, ``frozen_spatial()`` (``:1271-1297``), ``.config``, ``.device``,
  ``.eval()``, ``.to()``, ``state_dict()/load_state_dict()/save_pretrained()``.

All arithmetic of the forward runs in ``libstreamformer_hip.so`` through ``_native`` (ctypes);
torch supplies device memory, streams and the tensor container only.  There is no CPU or eager
fallback: without the library or without a GPU the forward raises.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

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
    ``DynamicCache``: ``get_seq_length()`` (vqa_enc:328-331) and being threaded through calls."""

    def __init__(self, model: "TimesformerMultiTaskingModelSigLIP", batch: int, max_frames: int, H: int, W: int):
        self._model = model
        self._h = nat.C.c_void_p()
        nat.check(nat.lib.sf_cache_create(model._handle, batch, max_frames, H, W, nat.C.byref(self._h)))
        self.batch, self.max_frames, self.H, self.W = batch, max_frames, H, W

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return nat.lib.sf_cache_length(self._h)

    def reset(self) -> None:
        nat.check(nat.lib.sf_cache_reset(self._h))

    @property
    def nbytes(self) -> int:
        return nat.lib.sf_cache_bytes(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                nat.lib.sf_cache_destroy(h)
            except Exception:
                pass


def expected_keys(cfg: StreamformerConfig, lora: Optional[bool] = None) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys and shapes of the reference module."""
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

    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{i}."
        k[p + "temporal_attention_gating"] = ()
        ln(p + "temporal_layernorm")
        lin(p + "temporal_attention.attention.qkv", 3 * D, D, cfg.qkv_bias)
        lin(p + "temporal_attention.output.dense", D, D)
        lin(p + "temporal_dense", D, D)
        ln(p + "layernorm_before")
        lin(p + "attention.attention.qkv", 3 * D, D, cfg.qkv_bias)
        lin(p + "attention.output.dense", D, D)
        if lora:
            k[p + "attention.attention.qkv_lora_a.weight"] = (LORA_RANK, D)
            k[p + "attention.attention.qkv_lora_b.weight"] = (3 * D, LORA_RANK)
            k[p + "attention.output.dense_lora_a.weight"] = (LORA_RANK, D)
            k[p + "attention.output.dense_lora_b.weight"] = (D, LORA_RANK)
        ln(p + "layernorm_after")
        lin(p + "intermediate.dense", I, D)
        lin(p + "output.dense", D, I)
    ln("post_layernorm")
    k["head.probe"] = (1, 1, D)
    k["head.attention.in_proj_weight"] = (3 * D, D)
    k["head.attention.in_proj_bias"] = (3 * D,)
    lin("head.attention.out_proj", D, D)
    ln("head.layernorm")
    lin("head.mlp.fc1", I, D)
    lin("head.mlp.fc2", D, I)
    return k


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


class TimesformerMultiTaskingModelSigLIP:
    """MI355X-native stand-in for the reference class of the same name (modeling:1241-1354)."""

    config_class = StreamformerConfig
    base_model_prefix = "timesformer"      # modeling:1073
    main_input_name = "pixel_values"       # modeling:1074

    def __init__(self, config: StreamformerConfig, compute_dtype: Any = "bf16", device: Any = None,
                 fuse_temporal_proj: bool = True):
        if config.attention_type != "divided_space_time":
            # the reference asserts the same wherever StreamFormer touches the encoder (modeling:1272-1274)
            raise NotImplementedError(
                f"attention_type={config.attention_type!r}: only 'divided_space_time' is on the StreamFormer path")
        if config.hidden_act not in _ACT_CODES:
            raise ValueError(f"unsupported hidden_act {config.hidden_act!r}")
        if compute_dtype not in _COMPUTE:
            raise ValueError(f"compute_dtype must be one of 'bf16' (throughput) or 'fp32'/'bf16x3' (accurate), got {compute_dtype!r}")
        self.config = config
        self.training = False
        self._compute = _COMPUTE[compute_dtype]
        self._fuse = bool(fuse_temporal_proj)
        self._lora = bool(config.add_lora_spatial)
        self._frozen: set = set()
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()   # host fp32 master copy
        self._handle = None
        self._device = torch.device("cpu")
        self._dirty = True
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        from .processing import TimesformerImageProcessor
        self.image_processor = TimesformerImageProcessor(size=(config.image_size, config.image_size))
        self._init_default_weights()
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------------------------ weights
    def _init_default_weights(self) -> None:
        """Reference-like defaults for a freshly constructed model (modeling:1077-1109, :896, :377):
        trunc-normal(0.02) matrices, zero biases, identity LayerNorm, zero gate / time embeddings."""
        g = torch.Generator().manual_seed(0)
        std = float(self.config.initializer_range)
        for k, shape in expected_keys(self.config, self._lora).items():
            if k.endswith("layernorm.weight") or k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight"):
                t = torch.ones(shape)
            elif k.endswith(".bias") or k.endswith("gating") or k.endswith("time_embeddings") or k.endswith("_lora_b.weight"):
                t = torch.zeros(shape)
            elif k == "head.probe":
                t = torch.randn(shape, generator=g)
            else:
                t = torch.nn.init.trunc_normal_(torch.empty(shape), std=std, a=-2 * std, b=2 * std, generator=g)
            self._sd[k] = t

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        sd = normalize_checkpoint_keys(state_dict)
        has_lora = any("_lora_" in k for k in sd)
        if has_lora and not self._lora:
            self._enable_lora_keys()
        exp = expected_keys(self.config, self._lora)
        unexpected = [k for k in sd if k not in exp and not k.endswith("temporal_attention.attention.mask")]
        missing = [k for k in exp if k not in sd]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                               f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        for k, shape in exp.items():
            if k in sd:
                t = sd[k].detach().to("cpu", torch.float32)
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)} vs model {tuple(shape)}")
                self._sd[k] = t.contiguous().clone()
        self._dirty = True
        return missing, unexpected

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, config: Optional[StreamformerConfig] = None,
                        compute_dtype: Any = "bf16", device: Any = None, device_map: Any = None,
                        torch_dtype: Any = None, **kwargs):
        path = str(pretrained_model_name_or_path)
        if not os.path.isdir(path):
            raise OSError(f"{path!r} is not a local directory (no hub access in this build); expected "
                          "config.json + model.safetensors or pytorch_model.bin")
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
        if device is None and isinstance(device_map, (str, int, torch.device)):
            device = device_map
        if device is None and torch.cuda.is_available():
            device = "cuda"
        if device is not None:
            model.to(device)
        return model

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True) -> None:
        os.makedirs(save_directory, exist_ok=True)
        self.config.add_lora_spatial = self._lora
        self.config.save_pretrained(save_directory)
        if safe_serialization:
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in self._sd.items()}, os.path.join(save_directory, "model.safetensors"),
                      metadata={"format": "pt"})
        else:
            torch.save(dict(self._sd), os.path.join(save_directory, "pytorch_model.bin"))

    # ------------------------------------------------------------------------------ LoRA surface
    def _enable_lora_keys(self) -> None:
        self._lora = True
        self.config.add_lora_spatial = True
        g = torch.Generator().manual_seed(1)
        for k, shape in expected_keys(self.config, True).items():
            if k not in self._sd:
                # modeling:533-534: A ~ N(0, 0.02), B = 0  => no change to the forward until trained
                self._sd[k] = torch.zeros(shape) if "_lora_b" in k else torch.randn(shape, generator=g) * 0.02
        self._dirty = True

    def add_lora_spatial(self) -> None:
        """modeling:1271-1282: rank-32 LoRA on every spatial qkv / output.dense; base weights frozen."""
        self._enable_lora_keys()
        for i in range(self.config.num_hidden_layers):
            for n in ("attention.attention.qkv", "attention.output.dense"):
                self._frozen.update({f"encoder.layer.{i}.{n}.weight", f"encoder.layer.{i}.{n}.bias"})
        print("Added LoRA to the following layers: ",
              [f"timesformer.encoder.layer.{i}.attention" for i in range(self.config.num_hidden_layers)])

    def frozen_spatial(self) -> None:
        """modeling:1284-1297 freezes the spatial qkv (its ``attention.dense`` line would raise in the
        reference; only the qkv freeze is observable)."""
        for i in range(self.config.num_hidden_layers):
            self._frozen.update({f"encoder.layer.{i}.attention.attention.qkv.weight",
                                 f"encoder.layer.{i}.attention.attention.qkv.bias"})

    def trainable_parameter_names(self) -> List[str]:
        return [k for k in self._sd if k not in self._frozen]

    # ------------------------------------------------------------------------- module-like surface
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32

    @property
    def compute_dtype(self) -> str:
        return "bf16" if self._compute == nat.SF_COMPUTE_BF16 else "bf16x3"

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = bool(mode)
        return self

    def requires_grad_(self, flag: bool = True):
        return self

    def parameters(self):
        return iter(self._sd.values())

    def num_parameters(self) -> int:
        return sum(v.numel() for v in self._sd.values())

    def cuda(self, device: Any = None):
        return self.to("cuda" if device is None else device)

    def to(self, *args, **kwargs):
        dev = kwargs.get("device")
        for a in args:
            if isinstance(a, (str, int, torch.device)):
                dev = a
        if dev is None:
            return self            # dtype-only .to(): the residual stream is fp32 by design
        dev = torch.device(dev if not isinstance(dev, int) else f"cuda:{dev}")
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev != self._device:
            self._device = dev
            self._dirty = True
            self._ws.clear()
            self._pos_cache.clear()
        return self

    def set_compute_dtype(self, compute_dtype: Any):
        c = _COMPUTE[compute_dtype]
        if c != self._compute:
            self._compute = c
            self._dirty = True
            self._ws.clear()
        return self

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                nat.lib.sf_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------------------ native sync
    def _sf_config(self) -> nat.SfConfig:
        c = self.config
        return nat.SfConfig(c.image_size, c.patch_size, c.num_channels, c.num_frames, c.hidden_size,
                            c.num_hidden_layers, c.num_attention_heads, c.intermediate_size,
                            _ACT_CODES[c.hidden_act], int(bool(c.qkv_bias)), int(bool(c.enable_causal_temporal)),
                            int(self._lora), float(c.layer_norm_eps))

    def _sync(self) -> None:
        """Create the handle on the current device and (re)upload packed weights when stale."""
        if self._device.type != "cuda":
            raise RuntimeError("the StreamFormer HIP encoder runs on an AMD GPU only: call .to('cuda') first "
                               "(there is no CPU fallback)")
        if not self._dirty and self._handle:
            return
        if self._handle:
            nat.lib.sf_destroy(self._handle)
            self._handle = None
        cfg = self._sf_config()
        h = nat.C.c_void_p()
        nat.check(nat.lib.sf_create(nat.C.byref(cfg), self._device.index or 0, nat.C.byref(h)))
        self._handle = h
        for k, t in self._sd.items():
            t = t.contiguous()
            shape = (nat.C.c_int64 * max(t.dim(), 1))(*t.shape)
            nat.check(nat.lib.sf_load_tensor(h, k.encode(), t.data_ptr(), _TORCH2SF[t.dtype], shape, t.dim()))
        with torch.cuda.device(self._device):
            nat.check(nat.lib.sf_finalize_weights(h, self._compute, 1, int(self._fuse)))
        ip = self.image_processor           # uint8 frames: rescale + normalize fused into the patch kernel
        nch = len(ip.image_mean)
        mean = (nat.C.c_float * nch)(*ip.image_mean)
        std = (nat.C.c_float * nch)(*ip.image_std)
        nat.check(nat.lib.sf_set_pixel_normalization(h, mean, std, nch, ip.rescale_factor))
        self._dirty = False

    def _workspace(self, key: tuple, nbytes: int) -> torch.Tensor:
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws.pop(key, None)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self._device)
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
            pe = self._sd["embeddings.position_embeddings"].float().reshape(1, M, M, c.hidden_size).permute(0, 3, 1, 2)
            pe = F.interpolate(pe, size=(w0, h0), mode="bicubic", antialias=True)
            assert (w0, h0) == tuple(pe.shape[-2:])
            self._pos_cache[key] = pe.permute(0, 2, 3, 1).reshape(-1, c.hidden_size).contiguous().to(self._device)
        return self._pos_cache[key]

    # ------------------------------------------------------------------------------------ forward
    def new_cache(self, batch_size: int = 1, max_frames: Optional[int] = None, height: Optional[int] = None,
                  width: Optional[int] = None) -> StreamCache:
        self._sync()
        c = self.config
        with torch.cuda.device(self._device):
            return StreamCache(self, batch_size, max_frames or c.num_frames, height or c.image_size, width or c.image_size)

    def forward(self, pixel_values: torch.Tensor, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                past_key_values: Optional[StreamCache] = None, use_cache: bool = False,
                cache_position: Optional[torch.Tensor] = None):
        c = self.config
        output_attentions = c.output_attentions if output_attentions is None else output_attentions
        output_hidden_states = c.output_hidden_states if output_hidden_states is None else output_hidden_states
        return_dict = c.use_return_dict if return_dict is None else return_dict
        if output_attentions and (use_cache or past_key_values is not None):
            raise NotImplementedError("output_attentions with use_cache")
        if pixel_values.dim() != 5:
            raise ValueError(f"pixel_values must be (B, T, C, H, W), got {tuple(pixel_values.shape)}")
        B, T, C_, H, W = pixel_values.shape
        if C_ != c.num_channels:
            raise ValueError(f"expected {c.num_channels} channels, got {C_}")
        self._sync()
        dev = self._device
        x = pixel_values.to(dev)
        if x.dtype not in (torch.float32, torch.bfloat16, torch.uint8):
            x = x.float()
        if x.dtype == torch.uint8 and W % 8:
            x = self.image_processor.normalize(x)       # byte path needs 8-pixel rows; fall back to fp32 frames
        x = x.contiguous()
        N = (H // c.patch_size) * (W // c.patch_size)
        D, L = c.hidden_size, c.num_hidden_layers
        pos = self._pos_table(H, W)
        streaming = bool(use_cache) or past_key_values is not None
        with torch.cuda.device(dev):
            stream = nat.current_stream_handle(dev)
            skey = int(stream or 0)          # one workspace per HIP stream: concurrent forwards never share scratch
            lhs = torch.empty(B, T, N, D, dtype=torch.float32, device=dev)
            pool = torch.empty(B, T, D, dtype=torch.float32, device=dev)
            nbytes = nat.C.c_size_t()
            if streaming:
                if output_hidden_states:
                    raise NotImplementedError("output_hidden_states with use_cache")
                cache = past_key_values
                if cache is None:
                    cache = StreamCache(self, B, max(c.num_frames, T), H, W)
                if cache_position is not None and int(cache_position[0]) != cache.get_seq_length():
                    raise ValueError("cache_position must continue the cache (vqa_enc:1340-1349)")
                if (cache.batch, cache.H, cache.W) != (B, H, W):
                    raise ValueError("past_key_values was created for a different batch size / resolution")
                nat.check(nat.lib.sf_stream_workspace_bytes(self._handle, cache._h, T, nat.C.byref(nbytes)))
                ws = self._workspace(("s", B, T, H, W, skey), nbytes.value)
                nat.check(nat.lib.sf_forward_stream(self._handle, cache._h, x.data_ptr(), _TORCH2SF[x.dtype], T,
                                                    lhs.data_ptr(), pool.data_ptr(), nat.ptr(pos), ws.data_ptr(),
                                                    ws.numel(), stream))
                if not return_dict:
                    return (lhs, cache)
                return BaseModelOutputWithPast(lhs, past_key_values=cache, pooler_output=pool)
            hs = torch.empty(L + 1, B, T, N, D, dtype=torch.float32, device=dev) if output_hidden_states else None
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
            hidden = tuple(hs[i].permute(0, 2, 1, 3).reshape(B, N * T, D) for i in range(L + 1))
        attentions = tuple(att[i] for i in range(L)) if att is not None else None    # spatial probabilities per layer
        if not return_dict:
            return (lhs,) + ((hidden,) if hidden is not None else ()) + ((attentions,) if attentions is not None else ())
        return BaseModelOutputWithPooling(lhs, pool, hidden_states=hidden, attentions=attentions)

    __call__ = forward


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
        return self._workspace(("f", B, T, H, W, int(nat.current_stream_handle(self._device) or 0)), n.value)

    def _grid(self, n_tokens: int, T: int) -> Tuple[int, int]:
        """(H, W) of a frame whose patch grid has n_tokens / T cells (square, or the config's aspect)."""
        N = n_tokens // T
        P = self.config.patch_size
        side = int(round(N ** 0.5))
        if side * side != N or N * T != n_tokens:
            raise ValueError(f"{n_tokens} tokens do not form {T} frames of a square patch grid")
        return side * P, side * P

    @property
    def embeddings(self):
        return _Embeddings(self)

    @property
    def encoder(self):
        return _Encoder(self)

    def post_layernorm(self, x: torch.Tensor) -> torch.Tensor:
        """nn.LayerNorm(D, eps) with the post_layernorm weights (modeling:1251, 1330), any leading shape."""
        self._sync()
        dev = self._device
        xf = x.to(dev, torch.float32).contiguous()
        y = torch.empty_like(xf)
        g = self._sd["post_layernorm.weight"].to(dev, torch.float32).contiguous()
        b = self._sd["post_layernorm.bias"].to(dev, torch.float32).contiguous()
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_op_layernorm(xf.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), xf.numel() // xf.shape[-1],
                                              xf.shape[-1], float(self.config.layer_norm_eps), nat.current_stream_handle(dev)))
        return y

    def head(self, x: torch.Tensor) -> torch.Tensor:
        """TimesformerSiglipMultiheadAttentionPoolingHead.forward (modeling:1141-1154): x (F, N, D) -> (F, D)."""
        Fr, N, D = x.shape
        H, W = self._grid(N, 1)
        ws = self._stage_ws(Fr, 1, H, W)
        dev = self._device
        xf = x.to(dev, torch.float32).contiguous()
        pool = torch.empty(Fr, D, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_post_head(self._handle, xf.data_ptr(), Fr, 1, H, W, None, pool.data_ptr(), ws.data_ptr(), ws.numel(),
                                           nat.current_stream_handle(dev)))
        return pool

    def forward_features(self, pixel_values: torch.Tensor, pooling_method: str = "last") -> torch.Tensor:
        """StreamformerForMultiTaskingSigLIP.forward_features (modeling:1525-1536)."""
        out = self.forward(pixel_values)
        if pooling_method == "last":
            return out.pooler_output[:, -1]
        if pooling_method in ("mean", "avg"):
            return out.pooler_output.mean(dim=1)
        raise ValueError(pooling_method)



def _to_frame_major(x: torch.Tensor, T: int) -> torch.Tensor:
    """reference (B, N*T, D), token = n*T + t  ->  [B, T, N, D] contiguous fp32"""
    B, NT, D = x.shape
    return x.reshape(B, NT // T, T, D).permute(0, 2, 1, 3).contiguous().float()


def _to_patch_major(h: torch.Tensor) -> torch.Tensor:
    B, T, N, D = h.shape
    return h.permute(0, 2, 1, 3).reshape(B, N * T, D)


class _Embeddings:
    """``model.embeddings(pixel_values) -> (B, N*T, D)`` (TimesformerEmbeddingsSigLIP.forward, modeling:413-457)."""

    def __init__(self, model: "TimesformerMultiTaskingModelSigLIP"):
        self.model = model

    def __call__(self, pixel_values: torch.Tensor) -> torch.Tensor:
        m = self.model
        B, T, _, H, W = pixel_values.shape
        ws = m._stage_ws(B, T, H, W)
        dev = m._device
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


class _Layer:
    """``model.encoder.layer[i](hidden_states, T, output_attentions=False) -> (hidden_states[, attn])``
    (TimesformerLayerSigLIP.forward, modeling:934-1004), patch-major in and out."""

    def __init__(self, model: "TimesformerMultiTaskingModelSigLIP", index: int):
        self.model, self.index = model, index

    def __call__(self, hidden_states: torch.Tensor, T: int, output_attentions: bool = False):
        out = self.model.encoder._run(hidden_states, T, self.index, self.index + 1, output_attentions)
        return (out[0],) + ((out[1][0],) if output_attentions else ())


class _Encoder:
    """``model.encoder(hidden_states, num_frames=T, ...)`` (TimesformerEncoder.forward, modeling:1019-1063)."""

    def __init__(self, model: "TimesformerMultiTaskingModelSigLIP"):
        self.model = model
        self.layer = [_Layer(model, i) for i in range(model.config.num_hidden_layers)]

    def _run(self, hidden_states: torch.Tensor, T: int, la: int, lb: int, want_attn: bool):
        m = self.model
        B, NT, D = hidden_states.shape
        H, W = m._grid(NT, T)
        ws = m._stage_ws(B, T, H, W)
        dev = m._device
        h = _to_frame_major(hidden_states.to(dev), T)
        N = NT // T
        att = torch.empty(lb - la, B * T, m.config.num_attention_heads, N, N, dtype=torch.float32, device=dev) if want_attn else None
        with torch.cuda.device(dev):
            nat.check(nat.lib.sf_layers(m._handle, h.data_ptr(), B, T, H, W, la, lb, nat.ptr(att), ws.data_ptr(), ws.numel(),
                                        nat.current_stream_handle(dev)))
        return _to_patch_major(h), (tuple(att[i] for i in range(lb - la)) if want_attn else None)

    def __call__(self, hidden_states: torch.Tensor, num_frames: int, output_attentions: bool = False,
                 output_hidden_states: bool = False, return_dict: bool = True):
        L = self.model.config.num_hidden_layers
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


class TimesformerVisionTower:
    """Per-stream state machine of the VideoQA vision tower (vqa_enc:1494-1500, 1528-1544): threads the
    KV-cache, concatenates outputs along time and returns the last ``context_length`` frames."""

    def __init__(self, model: TimesformerMultiTaskingModelSigLIP, context_length: int = 16,
                 max_frames: Optional[int] = None, streaming_mode: bool = True):
        self.vision_tower = model
        self.config = model.config
        self.context_length = context_length
        self.streaming_mode = streaming_mode
        self.max_frames = max_frames or model.config.num_frames
        self.past_key_values: Optional[StreamCache] = None
        self.hidden_states: Optional[torch.Tensor] = None
        self.image_processor = model.image_processor      # vqa_enc:1503-1505 keeps the processor on the tower

    def clear_cache(self) -> None:
        self.hidden_states = None
        if self.past_key_values is not None:
            self.past_key_values.reset()

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        if not self.streaming_mode:
            return self.vision_tower(images).last_hidden_state
        B, T, _, H, W = images.shape
        if self.past_key_values is None:
            self.past_key_values = self.vision_tower.new_cache(B, self.max_frames, H, W)
        out = self.vision_tower(images, use_cache=True, past_key_values=self.past_key_values)
        lhs = out.last_hidden_state
        self.hidden_states = lhs if self.hidden_states is None else torch.cat([self.hidden_states, lhs], dim=1)
        # bounded memory: nothing older than the window is ever returned (the reference keeps all)
        self.hidden_states = self.hidden_states[:, -self.context_length:]
        return self.hidden_states.to(images.dtype)

    __call__ = forward
