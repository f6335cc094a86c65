"""world_size-2 `gloo` test of the data-parallel exchange step (runs on CPU): the flat-arena
bucketed SUM all-reduce + 1/world scale equals the big-batch gradient, and every rank ends with
identical buffers.  On the GPU box the same code path runs over RCCL/xGMI (backend "nccl")."""

from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neosr_amd.models.base import allreduce_flat_
    from neosr_amd.utils.dist_util import get_dist_info, init_dist

    init_dist("pytorch", backend="gloo")
    init_dist("pytorch", backend="gloo")  # idempotent (reference re-enters it: SURVEY App. B-3)
    assert get_dist_info() == (rank, world)
    g = torch.Generator().manual_seed(123)
    full = torch.randn(world, 100_003, generator=g)          # per-rank "gradients"
    flat = full[rank].clone()
    allreduce_flat_(flat, bucket_bytes=64 * 1024)            # several ragged buckets
    flat *= 1.0 / world
    ok = torch.allclose(flat, full.mean(0), atol=1e-6)
    # loss-dict reduce: mean over ranks visible on rank 0
    from neosr_amd.models.base import base

    m = base.__new__(base)
    m.opt = {"dist": True, "rank": rank, "world_size": world}
    m._log_dev = None
    m._log_work = None
    m.log_dict = {}
    m.reduce_loss_dict({"l_g_pix": torch.tensor(float(rank + 1)), "l_g_total": torch.tensor([2.0 * (rank + 1)])})
    log = m.get_current_log()
    if rank == 0:
        ok = ok and abs(log["l_g_pix"] - 1.5) < 1e-6 and abs(log["l_g_total"] - 3.0) < 1e-6
    # GradSync bookkeeping (the overlapped exchange of the model step): two suffix buckets as the RRDB plan would
    # send them during backward, then the head of the arena from start(); finish() leaves the full SUM everywhere
    from neosr_amd.utils.grad_sync import GradSync

    gs = GradSync(n_marks=2, device="cpu")
    assert gs.mark_blocks(23) == [15, 7] and gs.mark_blocks(2) == [1] and gs.mark_blocks(1) == []
    flat2 = full[rank].clone()
    gs.begin(flat2)
    gs.reduce_suffix(70_000, None)
    gs.reduce_suffix(30_001, None)
    gs.start(flat2, bucket_elems=20_000)
    gs.finish()
    ok = ok and torch.allclose(flat2, full.sum(0), atol=1e-5)
    ok = ok and gs.buckets == [(70_000, 100_003), (30_001, 70_000), (10_001, 30_001), (0, 10_001)]
    flat3 = full[rank].clone()  # layer-composed network: nothing sent during backward
    gs.start(flat3)
    gs.finish()
    ok = ok and torch.allclose(flat3, full.sum(0), atol=1e-5) and gs.buckets == [(0, 100_003)]
    q.put((rank, bool(ok), float(flat.sum())))
    dist.destroy_process_group()


def test_flat_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == pytest.approx(res[1][2], abs=0)      # bit-identical on both ranks
