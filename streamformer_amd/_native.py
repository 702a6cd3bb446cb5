"""ctypes binding of ``libstreamformer_hip.so`` (the C ABI in ``include/streamformer_hip.h``).

There is no fallback: if the shared library is missing or does not load, importing this module
raises, and so does every product path that needs it.  ``torch`` is imported first on purpose — the
library is linked against ``libamdhip64.so.7`` by SONAME only, so it binds to the HIP runtime torch
has already loaded and shares its device context, allocator pointers and streams.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below: shared HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstreamformer_hip.so")
if os.environ.get("SF_LIB"):      # tools/ only: "lab" = the -DSF_LAB measurement build (`build.py --lab`), other names = A/B builds (`build.py --variant=name -D...`)
    LIB_PATH = os.path.join(_HERE, "libstreamformer_hip_%s.so" % os.environ["SF_LIB"])

SF_OK = 0
SF_ERR_INVALID, SF_ERR_STATE, SF_ERR_HIP, SF_ERR_WORKSPACE, SF_ERR_UNKNOWN_KEY, SF_ERR_CAPACITY = -1, -2, -3, -4, -5, -6
SF_F32, SF_BF16, SF_F16, SF_F64, SF_U8 = 0, 1, 2, 3, 4
SF_COMPUTE_BF16, SF_COMPUTE_BF16X3 = 0, 1


class SfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "image_size", "patch_size", "num_channels", "num_frames", "hidden_size", "num_hidden_layers",
        "num_attention_heads", "intermediate_size", "hidden_act", "qkv_bias", "enable_causal_temporal",
        "add_lora_spatial")] + [("layer_norm_eps", C.c_float)]


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"streamformer_hip error {code}: {msg}")
        self.code = code


# name -> (restype, argtypes): every symbol include/streamformer_hip.h declares
_P, _I, _F, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "sf_create": (_I, [C.POINTER(SfConfig), _I, C.POINTER(_P)]),
    "sf_destroy": (None, [_P]),
    "sf_last_error": (C.c_char_p, []),
    "sf_abi_version": (_I, []),
    "sf_load_tensor": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64), _I]),
    "sf_finalize_weights": (_I, [_P, _I, _I, _I]),
    "sf_missing_weights": (_I, [_P]),
    "sf_set_pixel_normalization": (_I, [_P, _P, _P, _I, _F]),
    "sf_workspace_bytes": (_I, [_P, _I, _I, _I, _I, C.POINTER(_SZ)]),
    "sf_forward": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_forward_profile": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _SZ, _P, C.POINTER(_F)]),
    "sf_forward_attentions": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_embed": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "sf_layers": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "sf_post_head": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "sf_cache_create": (_I, [_P, _I, _I, _I, _I, C.POINTER(_P)]),
    "sf_cache_reset": (_I, [_P]),
    "sf_cache_length": (_I, [_P]),
    "sf_cache_bytes": (_SZ, [_P]),
    "sf_cache_set_policy": (_I, [_P, _I]),
    "sf_cache_destroy": (None, [_P]),
    "sf_stream_workspace_bytes": (_I, [_P, _P, _I, C.POINTER(_SZ)]),
    "sf_forward_stream": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_forward_stream_attentions": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "sf_op_linear": (_I, [_P, _P, _P, _P, _F, _I, _P, _I, _I, _I, _I, _P, _SZ, _P]),
    "sf_op_linear_workspace_bytes": (_SZ, [_I, _I, _I]),
    "sf_op_attention": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "sf_op_attention_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "sf_loss_workspace_bytes": (_SZ, [_I, _I]),
    "sf_retrieval_loss": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_localization_loss": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "sf_trainer_create": (_I, [C.POINTER(SfConfig), _I, _I, _I, C.POINTER(_P)]),
    "sf_trainer_destroy": (None, [_P]),
    "sf_trainer_num_params": (_I, [_P]),
    "sf_trainer_param_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "sf_trainer_total_floats": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sf_trainer_num_stages": (_I, [_P]),
    "sf_trainer_stage_range": (_I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "sf_trainer_sync_weights": (_I, [_P, _P, _P]),
    "sf_trainer_workspace_bytes": (_I, [_P, _I, _I, C.POINTER(_SZ)]),
    "sf_trainer_forward": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "sf_trainer_backward": (_I, [_P, _P, _P, _P, _I, _I, _P, _SZ, _P]),
    "sf_trainer_adamw_step": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _F, _P, _F, _I, _P]),
    "sf_trainer_set_drop_path": (_I, [_P, _P, _I, _I]),
    "sf_trainer_set_dropout": (_I, [_P, _F, _F, C.c_uint32]),
    "sf_trainer_set_nonfinite_guard": (_I, [_P, _P, _P]),
    "sf_trainer_set_extra_steps": (_I, [_P, C.POINTER(C.c_int32), _I]),
    "sf_trainer_grad_sumsq": (_I, [_P, _P, _P, _P]),
    "sf_op_wgrad": (_I, [_P, _I, _P, _I, _I, _I, _I, _F, _I, _P, _I, _P, _P]),
    "sf_op_attention_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sf_op_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "sf_reload_switches": (None, []),
    "sf_switch_info": (C.c_char_p, [_I, _I]),
    "sf_bench_launch_floor": (_I, [_I, _I, _I, _P, C.POINTER(_F)]),
    "sf_bench_gemm": (_I, [_P, _I, _I, _I, _P, _SZ, _P, C.POINTER(_F), C.POINTER(C.c_double)]),
    "sf_bench_attention": (_I, [_P, _I, _I, _I, _I, _P, _SZ, _P, C.POINTER(_F), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python streamformer_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  There is no non-HIP fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name} (stale build? run "
                              "`python streamformer_amd/build.py`)") from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(code: int) -> None:
    if code != SF_OK:
        raise NativeError(code, (lib.sf_last_error() or b"").decode(errors="replace"))


def ptr(t) -> int:
    """Device/host address of a tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream_handle(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
