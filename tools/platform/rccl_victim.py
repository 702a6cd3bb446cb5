"""Device-sharing experiment, round 6: an RCCL all-reduce at world size 1 as the VICTIM — the reduction is the identity, so every
element of the result is known bit for bit — looped for SF_VICTIM_SECONDS while another process (tools/noise_ops.py attn_bwd_temporal) runs the
temporal attention backward on the same device.  Prints launches / launches with a differing element."""
import os, sys, time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
ref = torch.randn(407 * 1000 * 1000 // 16, generator=g).to(dev)          # a sixteenth of the 407 MB gradient buffer per call
sec = float(os.environ.get("SF_VICTIM_SECONDS", "20"))
t0 = time.time(); n = bad = 0
while time.time() - t0 < sec:
    t = ref.clone()
    dist.all_reduce(t)
    u = torch.empty_like(ref)
    dist.all_reduce(t, op=dist.ReduceOp.AVG)
    dist.broadcast(t, 0)
    out = [torch.empty_like(t)]
    dist.all_gather(out, t)
    torch.cuda.synchronize()
    n += 1
    if not (torch.equal(t, ref) and torch.equal(out[0], ref)):
        bad += 1
print(f"rccl world-size-1 victim: {n} rounds of all_reduce(SUM) + all_reduce(AVG) + broadcast + all_gather on {ref.numel() * 4 / 1e6:.0f} MB, {bad} with a differing element")
dist.destroy_process_group()
