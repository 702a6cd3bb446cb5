"""Device-sharing experiment, round 6: is an RCCL-like reduction a victim of the temporal attention backward running beside it?
RCCL at world size 1 launches no reduction kernel, so the stand-in is what a ring step does: out = a + b over a gradient-sized fp32 buffer
(torch.add, and a bf16 variant), on a SECOND stream of this process, while the first stream loops sf_temporal_attn_bwd_kernel at the training
step's shape (the op entry point of the C ABI).  Every result is compared bit for bit with the one computed on an idle device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import streamformer_amd as sa
import streamformer_amd._native as nat

dev = torch.device("cuda:0")
sec = float(os.environ.get("SF_VICTIM_SECONDS", "20"))
g = torch.Generator().manual_seed(1)
D = 768
rb = lambda *s: torch.randn(*s, generator=g).bfloat16().to(dev)
qkv_t, o_t, do_t = rb(8, 16, 196, 3 * D), rb(8, 16, 196, D), rb(8, 16, 196, D)
dq_t = torch.empty_like(qkv_t)
n = 101_770_000 // 4                           # a quarter of the 407 MB gradient buffer per add
a32, b32 = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
a16, b16 = a32.bfloat16(), b32.bfloat16()
want32, want16 = a32 + b32, a16 + b16          # idle device
torch.cuda.synchronize()
s_noise, s_vict = torch.cuda.Stream(), torch.cuda.Stream()
st = s_noise.cuda_stream
rounds = bad = 0
t0 = time.time()
while time.time() - t0 < sec:
    with torch.cuda.stream(s_noise):
        for _ in range(16):
            nat.check(nat.lib.sf_op_attention_bwd(qkv_t.data_ptr(), o_t.data_ptr(), do_t.data_ptr(), dq_t.data_ptr(), 1, 8 * 196, 16, 196, 12, 1, st))
    with torch.cuda.stream(s_vict):
        outs = []
        for _ in range(4):
            outs.append((a32 + b32, a16 + b16))
    torch.cuda.synchronize()
    for o32, o16 in outs:
        rounds += 1
        if not (torch.equal(o32, want32) and torch.equal(o16, want16)):
            bad += 1
print(f"add victim (fp32 + bf16 sums of {n * 4 / 1e6:.0f} MB, second stream) beside sf_temporal_attn_bwd_kernel: {rounds} rounds, {bad} with a differing element", flush=True)
