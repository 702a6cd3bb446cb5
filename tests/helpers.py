"""Shared test helpers: seeded inputs identical to oracle/make_golden.py, small configs."""
import numpy as np
import torch

from streamformer_amd.configuration import StreamformerConfig


def small_cfg(**kw):
    base = dict(image_size=48, patch_size=16, num_frames=16, hidden_size=128, num_hidden_layers=2,
                num_attention_heads=2, intermediate_size=256, enable_causal_temporal=True)
    base.update(kw)
    return StreamformerConfig(**base)


def frames(seed, shape, clamp=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    return x.clamp_(-1, 1) if clamp else x


def maxabs(a, b):
    a = torch.as_tensor(np.asarray(a)) if not torch.is_tensor(a) else a
    b = torch.as_tensor(np.asarray(b)) if not torch.is_tensor(b) else b
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def _rank_entry(fn, rank, world, port, args, q, backend="gloo"):
    """Child process of run_ranks: rendezvous on 127.0.0.1, run fn(rank, world, *args), report result or traceback."""
    import os
    import traceback
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        if backend == "nccl":               # RCCL: one rank per device; bind the communicator to the device up front
            import torch
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        q.put((rank, "ok", fn(rank, world, *args)))
    except BaseException:
        q.put((rank, "error", traceback.format_exc()))
    finally:
        q.close()
        q.join_thread()      # flush the result before leaving
        os._exit(0)          # no destroy_process_group: a peer that died must not leave this rank waiting in a collective


def run_ranks(fn, world, args=(), timeout=240, backend="gloo"):
    """Run fn(rank, world, *args) in `world` spawned processes over `backend` (gloo; "nccl" = RCCL, one rank per GPU); returns [result of rank 0, 1, ...].
    Never hangs: results are awaited with a deadline and every child is killed afterwards; a rank's exception is
    re-raised here with its traceback."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_entry, args=(fn, r, world, port, args, q, backend), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    out, err = {}, None
    try:
        for _ in range(world):
            rank, status, payload = q.get(timeout=timeout)
            if status == "error":
                err = f"rank {rank} failed:\n{payload}"
                break
            out[rank] = payload
    except Exception as e:       # queue.Empty: a rank hung
        err = f"ranks did not finish within {timeout}s ({type(e).__name__}); finished: {sorted(out)}"
    finally:
        for p in procs:
            p.join(5)
            if p.is_alive():
                p.kill()
    assert err is None, err
    return [out[r] for r in range(world)]
