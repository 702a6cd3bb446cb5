"""Two half-batches on two CU-masked HIP streams (hipExtStreamCreateWithCUMask): each forward owns half of the CUs, so
the MFMA-bound main loops of one overlap the HBM-bound epilogues of the other instead of taking turns on the whole chip."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import streamformer_amd as sa
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224).cuda()
xa, xb = x[:4].contiguous(), x[4:].contiguous()
def one(n=10):
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def two(s0, s1, n=10):
    def step():
        with torch.cuda.stream(s0): m(xa)
        with torch.cuda.stream(s1): m(xb)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"one batch of 8, whole chip: {one():.3f} ms", flush=True)
alt = int("55" * 32, 16)
patterns = {"even/odd CUs": (alt, alt << 1), "low/high 128": ((1 << 128) - 1, ((1 << 128) - 1) << 128),
            "blocks of 16": (int("0000ffff" * 8, 16), int("ffff0000" * 8, 16)), "blocks of 32": (int("00000000ffffffff" * 4, 16), int("ffffffff00000000" * 4, 16))}
for name, (a, b) in patterns.items():
    try:
        s0, s1 = masked_stream(a), masked_stream(b)
        print(f"two half-batches, {name:14s}: {two(s0, s1):.3f} ms   (one half-batch alone on its half: {two(s0, s0) :.3f} ms for 2)")
    except Exception as e:
        print(name, "failed:", repr(e)[:200])
print(f"one batch of 8, whole chip: {one():.3f} ms", flush=True)
