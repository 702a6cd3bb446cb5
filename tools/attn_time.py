"""Attention kernels at the BASELINE batch, HIP-event timed back to back: spatial / temporal forward (sf_bench_attention) and the two backward kernels
(op entry, training shapes).  SF_LIB selects an A/B build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
dev = torch.device("cuda", 0)
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype=os.environ.get("SF_MODE", "bf16"))
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda"); m._sync()
ws = torch.randn(1 << 28, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, by, fl = nat.C.c_float(), nat.C.c_double(), nat.C.c_double()
out = []
for which, name in ((0, "spatial fwd"), (1, "temporal fwd")):
    nat.check(nat.lib.sf_bench_attention(m._handle, 8, 16, which, 50, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(by), nat.C.byref(fl)))
    out.append(f"{name} {ms.value*1e3:.1f} us")
D = 768
g = torch.Generator().manual_seed(1)
rb = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)
st = nat.current_stream_handle(dev)
for name, args in (("spatial bwd", (rb(128 * 196, 3 * D), rb(128 * 196, D), rb(128 * 196, D), 0, 128, 196, 1, 12, 0)),
                   ("temporal bwd", (rb(8 * 16 * 196, 3 * D), rb(8 * 16 * 196, D), rb(8 * 16 * 196, D), 1, 8 * 196, 16, 196, 12, 1))):
    qkv, o, do = args[:3]
    dq = torch.empty_like(qkv)
    call = lambda: nat.check(nat.lib.sf_op_attention_bwd(qkv.data_ptr(), o.data_ptr(), do.data_ptr(), dq.data_ptr(), *args[3:], st))
    for _ in range(5): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): call()
    e1.record(); torch.cuda.synchronize()
    out.append(f"{name} {e0.elapsed_time(e1) / 40 * 1e3:.1f} us")
print("; ".join(out))
