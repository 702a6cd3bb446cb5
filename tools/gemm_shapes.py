"""Per-launch time of the four Linear shapes at the BASELINE batch (M = 25088), HIP events around back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda"); m._sync()
dev = torch.device("cuda", 0)
ws = torch.randn(1 << 29, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, fl = nat.C.c_float(), nat.C.c_double()
M = int(os.environ.get("SF_M", "25088"))
out = []
for which, name in ((0, "up"), (1, "down"), (2, "qkv"), (3, "out")):
    nat.check(nat.lib.sf_bench_gemm(m._handle, M, which, 30, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(fl)))
    out.append(f"{name} {ms.value*1e3:.1f} us ({fl.value/ms.value/1e9:.0f} TF)")
print("M=%d: " % M + "; ".join(out))
