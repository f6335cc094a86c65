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
    # loss-dict reduce: mean over ranks
    from neosr_amd.models.base import base

    m = base.__new__(base)
    m.opt = {"dist": True, "rank": rank, "world_size": world}
    m._log_dev = None
    m._log_work = None
    m._log_health, m._iters_seen, m._log_iters, m.chain_slow_grace_iters = False, 0, 0, 20
    m.log_dict = {}
    m.reduce_loss_dict({"l_g_pix": torch.tensor(float(rank + 1)), "l_g_total": torch.tensor([2.0 * (rank + 1)])})
    log = m.get_current_log()   # an all-reduce since round 5 (the chain health words must reach every rank): mean everywhere
    mean = (world + 1) / 2
    ok = ok and abs(log["l_g_pix"] - mean) < 1e-6 and abs(log["l_g_total"] - 2 * mean) < 1e-6
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
    # hook-driven buckets of a layer-composed network (GradSync.attach: what SwinIR / HAT generators use): a small torch
    # model, per-rank half batches; buckets leave from inside backward in reverse arena order, `.grad` ends up as views of
    # ONE arena holding the SUM over ranks, and SUM / world equals the big-batch gradient
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(24, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                              torch.nn.Linear(64, 8))
    X = torch.randn(2 * world, 24, generator=g)
    params = list(net.parameters())
    gs2 = GradSync(device="cpu")
    gs2.attach(params, n_buckets=3)
    sent_inside = []
    for step in range(2):  # the second step re-uses the arena
        for p in params:
            p.grad = None
        gs2.armed = True
        gs2.arm_backward()
        net(X[2 * rank : 2 * rank + 2]).square().sum().backward()
        sent_inside.append(gs2.in_backward_buckets)
        assert gs2.end_backward()
        gs2.finish()
    ref = torch.nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else type(m)()
                                for m in net])
    ref.load_state_dict(net.state_dict())
    ref(X).square().sum().backward()
    from neosr_amd.hip.nets import flat_grad_of

    flat_g = flat_grad_of(params)
    ok = ok and flat_g is not None and sent_inside == [2, 2] and len(gs2.buckets) == 2   # (cut at parameter boundaries)
    ok = ok and gs2.buckets[0][1] == flat_g.numel() and sorted(gs2.buckets)[0][0] == 0   # last layers first
    ok = ok and all(torch.allclose(p.grad, rp.grad, rtol=1e-5, atol=1e-6) for p, rp in zip(params, ref.parameters()))
    gs2.armed = False   # accumulation / SAM steps: the hooks stay silent, the model falls back to start()
    for p in params:
        p.grad = None
    gs2.arm_backward()
    net(X[2 * rank : 2 * rank + 2]).square().sum().backward()
    ok = ok and not gs2.end_backward() and flat_grad_of(params) is None
    q.put((rank, bool(ok), float(flat.sum())))
    dist.destroy_process_group()


def _worker_unused(rank: int, world: int, port: int, q) -> None:
    """Hook-driven buckets when a parameter gets NO gradient on some ranks only (a data-dependent branch): every rank must
    still issue the same collectives in the same order (descending arena offset; a complete bucket waits for its
    predecessors), the unused slice contributes zeros — also when the arena still holds last step's gradient there."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neosr_amd.hip.nets import flat_grad_of
    from neosr_amd.utils.dist_util import init_dist
    from neosr_amd.utils.grad_sync import GradSync

    init_dist("pytorch", backend="gloo")
    torch.manual_seed(11)
    L = [torch.nn.Linear(16, 16) for _ in range(4)]
    params = [p for m in L for p in m.parameters()]
    g = torch.Generator().manual_seed(77)
    X = torch.randn(world, 3, 16, generator=g)

    def fwd(x, skip):
        h = L[0](x)
        if not skip:
            h = h + L[1](h)      # the branch
        return L[3](L[2](h))

    gs = GradSync(device="cpu")
    gs.attach(params, n_buckets=4)
    ok = len(gs._bucket_range) == 4
    for step in range(2):
        skip = (rank + step) % 2 == 1           # odd ranks skip the branch in step 0, even ranks in step 1
        for p in params:
            p.grad = None
        gs.armed = True
        gs.arm_backward()
        fwd(X[rank], skip).square().sum().backward()
        inside = gs.in_backward_buckets
        ok = ok and gs.end_backward()
        gs.finish()
        # same collectives, same order, on every rank (highest offset first)
        ok = ok and gs.buckets == sorted(gs._bucket_range, reverse=True)
        # ranks that skipped the branch could only send the buckets in front of it from inside backward
        ok = ok and (inside == 2 if skip else inside == 4)
        ref = [torch.zeros_like(p) for p in params]
        for r in range(world):
            grads = torch.autograd.grad(fwd(X[r], (r + step) % 2 == 1).square().sum(), params, allow_unused=True)
            for a, gr in zip(ref, grads):
                if gr is not None:
                    a += gr
        flat = flat_grad_of(params)
        ok = ok and flat is not None
        ok = ok and all(torch.allclose(p.grad, a, rtol=1e-5, atol=1e-6) for p, a in zip(params, ref))
    q.put((rank, bool(ok), float(flat.sum())))
    dist.destroy_process_group()


def test_hook_buckets_world4_unused_parameters_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unused, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert len({r[2] for r in res}) == 1      # bit-identical sums on all four ranks


@pytest.mark.parametrize("world", [2, 8])
def test_flat_allreduce_gloo(world):
    """world 2, and world 8 = the node size SCALE_rNN runs at (VERDICT r4 #7c): flat bucketed all-reduce, the hook-driven
    buckets of a layer-composed net leaving in one order on all eight ranks, the loss all-reduce"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == pytest.approx(res[1][2], abs=0)      # bit-identical on both ranks
