"""Lab (round 4): 8 clips as ONE call against two half-batches of 4 clips on two HIP streams at once (no CU masks: the dispatcher interleaves
the two kernel sequences, so one half's HBM-bound phases and round tails can sit under the other's MFMA-bound main loops).  Two module
instances (own workspaces), same weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, streamformer_amd as sa
cfg = sa.siglip_base()
sd = sa.make_state_dict(cfg, 0)
ms = []
for _ in range(2):
    m = sa.TimesformerMultiTaskingModelSigLIP(cfg, compute_dtype="bf16")
    m.load_state_dict(sd); m.to("cuda").eval()
    ms.append(m)
x = torch.randn(8, 16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
xa, xb = x[:4].contiguous(), x[4:].contiguous()
sa_, sb_ = torch.cuda.Stream(), torch.cuda.Stream()
def one():
    return ms[0](x)
def two():
    cur = torch.cuda.current_stream()
    sa_.wait_stream(cur); sb_.wait_stream(cur)
    with torch.cuda.stream(sa_): oa = ms[0](xa)
    with torch.cuda.stream(sb_): ob = ms[1](xb)
    cur.wait_stream(sa_); cur.wait_stream(sb_)
    return oa, ob
def timeit(f, n=30):
    with torch.no_grad():
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    ref = one(); oa, ob = two()
same = bool((torch.cat([oa.last_hidden_state, ob.last_hidden_state]) == ref.last_hidden_state).all())
t1 = timeit(one); t2 = timeit(two); t1b = timeit(one); t2b = timeit(two)
th = timeit(lambda: ms[0](xa))
print(f"8 clips, one call: {t1:.3f} / {t1b:.3f} ms;  two 4-clip calls on two streams: {t2:.3f} / {t2b:.3f} ms;  one 4-clip call alone: {th:.3f} ms;  results identical: {same}", flush=True)
