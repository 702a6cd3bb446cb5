"""Per-shape sweep of the 256^2 GEMM phase stagger through sf_bench_gemm (which: 0 mlp_up, 2 qkv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda"); m._sync()
dev = torch.device("cuda:0")
M = 8 * 16 * 196
ws = torch.randn(1 << 29, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, fl = nat.C.c_float(), nat.C.c_double()
for which, name in ((0, "mlp_up"), (2, "qkv")):
    for key, vals in (("SF_G256_STAGGER_NS", [0, 3000, 4000, 5000, 6000, 7000, 8000, 9000]),):
        for v in vals:
            os.environ[key] = str(v)
            best = 1e9
            for _ in range(3):
                nat.check(nat.lib.sf_bench_gemm(m._handle, M, which, 20, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(fl)))
                best = min(best, ms.value)
            print(f"{name} stagger {v:>5} ns: {best*1e3:.1f} us  {fl.value/best/1e9:.0f} TF", flush=True)
        os.environ.pop(key, None)
