"""CPU, world_size 2 over gloo: the N>1 plumbing around the (collective-free) forward.

  * clip sharding + max-over-ranks timing as bench.py does it;
  * retrieval loss with cross-rank negatives: all_gather of caption features + pos_offset = rank*B
    reproduces the reference's ring exchange (local block with positives + every other rank's block
    as negatives only, modeling:244-295), checked with the oracle on each rank;
  * mean all-reduce of flat gradient buckets.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import streamformer_oracle as O
        from streamformer_amd import parallel as P
        # --- sharding + timing reduction
        lo, hi = P.shard_range(13, rank, world)
        t = P.max_over_ranks(1.0 + rank)
        # --- cross-rank negatives
        g = torch.Generator().manual_seed(100 + rank)
        B, T, D = 4, 3, 32
        pooler = torch.randn(B, T, D, generator=g)
        text = torch.randn(B, D, generator=g)
        allt = P.all_gather_rows(text)
        assert allt.shape == (world * B, D) and torch.equal(allt[rank * B:(rank + 1) * B], text)
        ls, lb = torch.log(torch.tensor(10.0)), torch.tensor(-2.0)
        others = [allt[r * B:(r + 1) * B] for r in range(world) if r != rank]
        want = O.retrieval_loss(pooler, text, ls, lb, other_rank_text=others)
        # pos_offset formulation (what sf_retrieval_loss computes from the gathered table)
        img = pooler[:, -1]
        img = img / img.norm(dim=-1, keepdim=True)
        tn = allt / allt.norm(dim=-1, keepdim=True)
        logits = ls.exp() * img @ tn.t() + lb
        labels = -torch.ones(B, world * B)
        labels[torch.arange(B), rank * B + torch.arange(B)] = 1
        got = -torch.nn.functional.logsigmoid(labels * logits).sum() / B
        # --- gradient buckets
        bucket = torch.full((5,), float(rank + 1))
        P.all_reduce_mean_([bucket])
        q.put((rank, lo, hi, t, float(want), float(got), bucket.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 7), (7, 13)]
    assert all(r[3] == 2.0 for r in res)                       # max over ranks
    for r in res:
        assert abs(r[4] - r[5]) < 1e-5                          # ring semantics == gather + pos_offset
        assert r[6] == [1.5] * 5
