"""CPU: host-side input pipeline pieces either side of the hot path:
EnlargedSampler's contract (reference neosr/data/data_sampler.py:8-54) and the threaded prefetcher."""

from __future__ import annotations

import torch


def test_enlarged_sampler_contract():
    from neosr_amd.data.data_sampler import EnlargedSampler

    ds = list(range(10))
    world, ratio = 3, 4
    samplers = [EnlargedSampler(ds, world, r, ratio, device="cpu") for r in range(world)]
    per_rank = [list(s) for s in samplers]
    n = -(-len(ds) * ratio // world)  # ceil(10 * 4 / 3) = 14
    assert all(len(p) == n == len(s) for p, s in zip(per_rank, samplers))
    assert all(0 <= v < len(ds) for p in per_rank for v in p)
    # the ranks stride one permutation of range(total_size): interleaved they are that permutation mod len
    g = torch.Generator().manual_seed(0)
    perm = [v % len(ds) for v in torch.randperm(n * world, generator=g).tolist()]
    inter = [per_rank[i % world][i // world] for i in range(n * world)]
    assert inter == perm
    # deterministic in the epoch, different across epochs
    assert list(samplers[0]) == per_rank[0]
    samplers[0].set_epoch(1)
    assert list(samplers[0]) != per_rank[0]
    # every sample is visited at least floor(total/len) times over all ranks
    counts = [sum(p.count(i) for p in per_rank) for i in range(len(ds))]
    assert min(counts) >= (n * world) // len(ds)


