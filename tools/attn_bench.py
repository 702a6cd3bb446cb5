"""Spatial / temporal attention launch time at the BASELINE batch through the bench hook."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda"); m._sync()
dev = torch.device("cuda", 0)
ws = torch.randn(1 << 28, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, by, fl = nat.C.c_float(), nat.C.c_double(), nat.C.c_double()
B = int(os.environ.get("SF_B", "8"))
for which, name in ((0, "spatial"), (1, "temporal")):
    nat.check(nat.lib.sf_bench_attention(m._handle, B, 16, which, 50, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(by), nat.C.byref(fl)))
    print(f"{name}: {ms.value*1e3:.1f} us, {by.value/ms.value/1e6:.0f} GB/s", end="; ")
print()
