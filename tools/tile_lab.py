"""Times the four per-layer GEMM shapes at small batches through sf_bench_gemm:  python tools/tile_lab.py M [M ...]
Environment picks the kernel: SF_DISABLE_GEMM_TILE=1 (panel / 256^2 path), SF_TILE_SHAPE=<id> (force a tile candidate),
SF_TILE_MAX_M=<rows>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
m(torch.randn(1, 2, 3, 224, 224).cuda())
ws = torch.randn(1 << 30, dtype=torch.bfloat16, device="cuda").view(torch.uint8)
ms, fl = nat.C.c_float(), nat.C.c_double()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SF_"))
for M in [int(a) for a in sys.argv[1:]]:
    out = []
    for which, name in ((2, "qkv"), (0, "up"), (3, "out"), (1, "down")):
        nat.check(nat.lib.sf_bench_gemm(m._handle, M, which, 30, ws.data_ptr(), ws.numel(), nat.current_stream_handle(torch.device("cuda")),
                                        nat.C.byref(ms), nat.C.byref(fl)))
        out.append(f"{name} {ms.value*1e3:7.1f} us {fl.value/ms.value/1e9:6.0f} TF")
    print(f"M={M:6d} [{tag}] " + " | ".join(out), flush=True)
