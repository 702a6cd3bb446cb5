"""8-clip bf16 forward: time + checksum; run with and without SF_DISABLE_RESID_PLANES=1 (profiles/r03_resid_planes_ab.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    for _ in range(5): out = m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): out = m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"planes_off={os.environ.get('SF_DISABLE_RESID_PLANES')} {dt*1e3:.3f} ms  {128/dt:.0f} frames/s  checksum {out.last_hidden_state.double().abs().mean().item():.9f} {out.pooler_output.double().abs().mean().item():.9f}")
