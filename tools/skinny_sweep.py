"""Per-launch time of the four Linear shapes of a layer at small M (the streaming regime), back-to-back launches between HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda"); m._sync()
dev = torch.device("cuda", 0)
ws = torch.randn(1 << 27, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, fl = nat.C.c_float(), nat.C.c_double()
names = {0: "up   N=3072 K=768 gelu", 1: "down N=768 K=3072 resid", 2: "qkv  N=2304 K=768", 3: "out  N=768 K=768 resid"}
print("M      " + "  ".join(f"{names[w]:>24}" for w in range(4)))
for M in (16, 32, 64, 96, 128, 196, 256, 392, 512):
    row = []
    for which in range(4):
        nat.check(nat.lib.sf_bench_gemm(m._handle, M, which, 200, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(fl)))
        row.append(ms.value * 1e3)
    print(f"{M:<6d} " + "  ".join(f"{v:21.2f} us" for v in row))
