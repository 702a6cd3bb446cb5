"""Round 4 lab: 8-clip bf16 forward time + checksum + max-abs of clip 0 against the CPU oracle, and the isolated GEMM shapes,
under whatever SF_PANEL_PP* switches the environment carries (one process per setting: the switches are read once)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from streamformer_amd import _native as nat
cfg = sa.siglip_base()
sd = sa.make_state_dict(cfg, 0)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sd); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1))
xc = x.cuda()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("SF_"))
with torch.no_grad():
    for _ in range(5): out = m(xc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): out = m(xc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    out2 = m(xc)
rep = bool((out.last_hidden_state == out2.last_hidden_state).all()) and bool((out.pooler_output == out2.pooler_output).all())
line = f"[{tag}] fwd {dt*1e3:.3f} ms {128/dt:.0f} f/s reproducible={rep} checksum {out.last_hidden_state.double().abs().mean().item():.9f} {out.pooler_output.double().abs().mean().item():.9f}"
if os.environ.get("PP_LAB_ORACLE"):
    from oracle import streamformer_oracle as O
    want = O.forward(sd, cfg, x[:1])
    lhs = want["last_hidden_state"] if isinstance(want, dict) else want[0]
    pool = want["pooler_output"] if isinstance(want, dict) else want[1]
    line += f" max-abs lhs {float((out.last_hidden_state[:1].cpu() - lhs).abs().max()):.3e} pooler {float((out.pooler_output[:1].cpu() - pool).abs().max()):.3e}"
dev = torch.device("cuda", 0)
ws = torch.randn(1 << 29, dtype=torch.bfloat16, device=dev).view(torch.uint8)
ms, fl = nat.C.c_float(), nat.C.c_double()
parts = []
for which, name in ((1, "down"), (3, "out")):
    nat.check(nat.lib.sf_bench_gemm(m._handle, 25088, which, 30, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(fl)))
    parts.append(f"{name} {ms.value*1e3:.1f} us")
by = nat.C.c_double()
for which, name in ((0, "spatial"), (1, "temporal")):
    nat.check(nat.lib.sf_bench_attention(m._handle, 8, 16, which, 20, ws.data_ptr(), ws.numel(), nat.current_stream_handle(dev), nat.C.byref(ms), nat.C.byref(by), nat.C.byref(fl)))
    parts.append(f"{name} attn {ms.value*1e3:.1f} us ({by.value/ms.value/1e9:.2f} TB/s)")
print(line + " | " + "; ".join(parts), flush=True)
