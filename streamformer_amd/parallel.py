"""Data-parallel plumbing (one process per GPU, torch.distributed over RCCL; gloo on CPU in tests).

The encoder forward shards by clips with NO data-path collective (clips are independent: reference
inference shards by video list, scripts/downstream_extract_oad_feature.sh:29-50).  Collectives exist
only around it: the retrieval head needs every rank's caption features once per step (the reference
ring-exchanges them with batch_isend_irecv, modeling:244-295; one all_gather delivers the same
negatives), and a training step all-reduces the gradient buffer (DDP in the reference,
run_finetuning_multi_task.py:421-423).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size) of ``group`` (default process group when None); (0, 1) without torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) shard of n_items for this rank (first n % world ranks get one more)."""
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(t: torch.Tensor, group=None, at_world_1: bool = False) -> torch.Tensor:
    """[B, D] per rank -> [world*B, D] in rank order (same B on every rank): one all-gather into a single output
    tensor (48 KB at B = 8, D = 768) instead of the reference's ring of P2P exchanges (modeling:244-295).
    ``at_world_1``: issue the collective even in a 1-rank group (identity; exercises the backend on one GPU)."""
    rank, ws = world(group)
    if ws == 1 and not (at_world_1 and dist.is_available() and dist.is_initialized()):
        return t
    t = t.contiguous()
    if dist.get_backend(group) == "nccl":            # RCCL: gather straight into one output tensor
        out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        return out
    host = t.cpu()                                      # gloo (CPU tests, same-device smoke runs): gather on the host
    parts = [torch.empty_like(host) for _ in range(ws)]
    dist.all_gather(parts, host, group=group)
    return torch.cat(parts, dim=0).to(t.device)


def max_over_ranks(seconds: float, device=None, group=None) -> float:
    rank, ws = world(group)
    if ws == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def all_reduce_mean_(buckets: List[torch.Tensor], group=None, at_world_1: bool = False) -> None:
    """In-place mean all-reduce of a few large flat gradient buckets (one collective each)."""
    rank, ws = world(group)
    if ws == 1 and not (at_world_1 and dist.is_available() and dist.is_initialized()):
        return
    for b in buckets:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
        b.div_(ws)
