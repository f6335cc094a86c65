"""`EnlargedSampler` (neosr/data/data_sampler.py:8-54): DistributedSampler-style rank striding over a
dataset virtually enlarged `ratio` times, so an iteration-based run does not restart the loader workers
after every real epoch.  The permutation is drawn on `device` from a generator seeded with the epoch
(the reference draws it on "cuda"); indices wrap modulo the dataset size."""

from __future__ import annotations

import math
from collections.abc import Iterator

import torch
from torch.utils.data.sampler import Sampler


class EnlargedSampler(Sampler):
    def __init__(self, dataset, num_replicas: int = 1, rank: int = 1, ratio: int = 1, device: str = "cuda") -> None:
        self.dataset = dataset
        self.num_replicas = num_replicas
        self.rank = rank
        self.epoch = 0
        self.device = device
        self.num_samples = math.ceil(len(self.dataset) * ratio / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self) -> Iterator[int]:
        g = torch.Generator(device=self.device)
        g.manual_seed(self.epoch)  # deterministic in the epoch, identical on every rank
        order = torch.randperm(self.total_size, generator=g, device=self.device).tolist()
        size = len(self.dataset)
        mine = [v % size for v in order][self.rank: self.total_size: self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
