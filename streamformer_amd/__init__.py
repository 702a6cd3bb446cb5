"""streamformer_amd — MI355X-native (gfx950) StreamFormer encoder hot path.

One hot path of Go2Heart/StreamFormer, rebuilt from scratch: the
``TimesformerMultiTaskingModelSigLIP`` forward (+ its KV-cached streaming variant and the
retrieval / localization loss heads), as hand-written HIP kernels behind a C ABI
(``include/streamformer_hip.h``), with this package as the Python mirror of the reference module.
Importing the package loads ``libstreamformer_hip.so``; there is no fallback if it is missing.
"""
from .configuration import StreamformerConfig, siglip_base  # noqa: F401
from .init_weights import make_state_dict, state_dict_sha256  # noqa: F401
from . import _native  # noqa: F401  (raises ImportError when the HIP library is not built)
from .modeling import (  # noqa: F401
    BaseModelOutputWithPast,
    BaseModelOutputWithPooling,
    StreamCache,
    TimesformerMultiTaskingModelSigLIP,
    TimesformerVisionTower,
)
from . import heads  # noqa: F401
from .multitask import (  # noqa: F401
    StreamformerForMultiTaskingSigLIP,
    TimesformerUniversalLocalizationHead,
    TimesformerVideoRetrievalHead,
)
from .processing import TimesformerImageProcessor  # noqa: F401

__version__ = "0.1.0"
