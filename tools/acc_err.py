"""fp32-accurate mode: forward time at the BASELINE batch and max-abs error of one clip against the CPU oracle (fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
from oracle import streamformer_oracle as O
cfg = sa.siglip_base()
sd = sa.make_state_dict(cfg, 0)
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="fp32")
m.load_state_dict(sd); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1))
xc = x.cuda()
with torch.no_grad():
    for _ in range(3): out = m(xc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m(xc)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
want = O.forward(sd, cfg, x[:1])
lhs = want["last_hidden_state"] if isinstance(want, dict) else want[0]
pool = want["pooler_output"] if isinstance(want, dict) else want[1]
e1 = float((out.last_hidden_state[:1].cpu() - lhs).abs().max()); e2 = float((out.pooler_output[:1].cpu() - pool).abs().max())
print(f"{os.environ.get('SF_ACC_TWO_PLANES')} {os.environ.get('SF_DISABLE_ACC_FOLD')}: {dt*1e3:.2f} ms  {128/dt:.0f} frames/s  max-abs lhs {e1:.3e} pooler {e2:.3e}")
