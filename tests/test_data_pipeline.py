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


def test_prefetch_dataloader_yields_everything_in_order():
    # reference: neosr/data/prefetch_dataloader.py:9-66 (prefetch_mode = "cpu")
    from neosr_amd.data.prefetch_dataloader import PrefetchDataLoader, PrefetchGenerator

    data = [{"lq": torch.full((2,), float(i))} for i in range(7)]
    loader = PrefetchDataLoader(num_prefetch_queue=2, dataset=data, batch_size=None, shuffle=False)
    for _ in range(2):  # re-iterable
        got = [int(b["lq"][0]) for b in loader]
        assert got == list(range(7))

    def boom():
        yield 1
        raise RuntimeError("loader failed")

    it = PrefetchGenerator(boom(), 2)
    assert next(it) == 1
    try:
        next(it)
    except RuntimeError as e:
        assert "loader failed" in str(e)
    else:
        raise AssertionError("the producer's exception must reach the consumer")


def test_slurm_launcher_env_contract():
    # reference: neosr/utils/dist_util.py:37-69
    from neosr_amd.utils.dist_util import _slurm_rendezvous

    env = {"SLURM_PROCID": "11", "SLURM_NTASKS": "16", "SLURM_NODELIST": "n[1-2]"}
    out = _slurm_rendezvous(env, "n1", 8, None)
    assert out == {"RANK": "11", "WORLD_SIZE": "16", "LOCAL_RANK": "3", "MASTER_ADDR": "n1", "MASTER_PORT": "29500"}
    assert _slurm_rendezvous({**env, "MASTER_PORT": "4000"}, "n1", 8, None)["MASTER_PORT"] == "4000"
    assert _slurm_rendezvous({**env, "MASTER_PORT": "4000"}, "n1", 8, 5000)["MASTER_PORT"] == "5000"

