"""Random-draw sources for the on-the-fly degradation pipeline.

`otf.feed_data` consumes three RNG families in a fixed order: python `random`
(`choices/choice/randint`), a numpy `Generator` seeded with `manual_seed` (`neosr/utils/rng.py`),
and the torch generator (`rand/randn/poisson/uniform_/randperm`).  `LiveDraws` is that, with the
tensor draws made on the HIP device; `ReplayDraws` replays a recorded sequence (fixtures captured
from the reference) so that the pipeline becomes a deterministic function that can be compared
bit-for-bit in control flow and to 1e-3 in pixels.
"""

from __future__ import annotations

import random
from typing import Any

import numpy as np
import torch


class LiveDraws:
    def __init__(self, seed: int | None, device: torch.device | str) -> None:
        self.rng = np.random.default_rng(seed=seed) if seed is not None else np.random.default_rng()
        self.device = torch.device(device)

    # python `random`
    def choices(self, population, weights=None):
        return random.choices(population, weights)[0]

    def choice(self, seq):
        return random.choice(seq)

    def randint(self, a: int, b: int) -> int:
        return random.randint(a, b)

    # numpy Generator
    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return float(self.rng.uniform(lo, hi))

    def random(self) -> float:
        return float(self.rng.random())

    def integers(self, lo: int, hi: int | None = None) -> int:
        return int(self.rng.integers(lo, hi))

    # torch generator (device draws: no host round trip)
    def rand(self, n: int) -> torch.Tensor:
        return torch.rand(n, dtype=torch.float32, device=self.device)

    def randn(self, *shape: int) -> torch.Tensor:
        """`torch.randn(shape)` (degradations.py:593-598).  On a HIP device the field comes from our own counter-based
        sampler (Philox4x32-10 + Box-Muller), keyed by the device generator's (seed, offset) — advanced like a generator
        draw would — so `torch.manual_seed` still fixes the run and no ATen RNG kernel is on the product path."""
        if self.device.type != "cuda":
            return torch.randn(*shape, dtype=torch.float32, device=self.device)
        from neosr_amd.hip import degrade as D

        n = 1
        for v in shape:
            n *= int(v)
        seed, offset = self._advance(n)
        return D.normal_sample(shape, seed, offset, self.device)

    def _advance(self, n: int) -> tuple[int, int]:
        gen = torch.cuda.default_generators[self.device.index if self.device.index is not None
                                            else torch.cuda.current_device()]
        seed, offset = gen.initial_seed(), gen.get_offset()
        gen.set_offset(offset + 4 * ((n + 3) // 4))
        return seed, offset

    def poisson(self, rate: torch.Tensor) -> torch.Tensor:
        """`torch.poisson(rate)` (degradations.py:782-785).  On a HIP device the field comes from our own
        counter-based sampler, keyed by the device generator's (seed, offset) — which is advanced like a
        generator draw would — so `torch.manual_seed` still fixes the run."""
        if not rate.is_cuda:
            return torch.poisson(rate)
        from neosr_amd.hip import degrade as D

        gen = torch.cuda.default_generators[rate.device.index if rate.device.index is not None
                                            else torch.cuda.current_device()]
        seed, offset = gen.initial_seed(), gen.get_offset()
        gen.set_offset(offset + 4 * ((rate.numel() + 3) // 4))
        return D.poisson_sample(rate, seed, offset)

    def uniform_tensor(self, n: int, lo: float, hi: float) -> torch.Tensor:
        return torch.empty(n, dtype=torch.float32, device=self.device).uniform_(lo, hi)

    def randperm(self, n: int) -> torch.Tensor:
        return torch.randperm(n, device=self.device)


class ReplayDraws:
    """Replays `[(kind, value), ...]` recorded from a reference run; kinds are checked."""

    def __init__(self, record: list[tuple[str, Any]], device: torch.device | str = "cpu") -> None:
        self.record = list(record)
        self.pos = 0
        self.device = torch.device(device)

    def _next(self, kind: str):
        if self.pos >= len(self.record):
            raise RuntimeError(f"draw sequence exhausted (wanted {kind})")
        k, v = self.record[self.pos]
        self.pos += 1
        if k != kind:
            raise RuntimeError(f"draw #{self.pos - 1}: pipeline asked for {kind}, recording has {k}")
        return v

    def _t(self, kind: str) -> torch.Tensor:
        v = self._next(kind)
        return torch.as_tensor(np.asarray(v)).to(self.device)

    def choices(self, population, weights=None):
        return str(self._next("choices"))

    def choice(self, seq):
        return str(self._next("choice"))

    def randint(self, a: int, b: int) -> int:
        return int(self._next("randint"))

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return float(self._next("uniform"))

    def random(self) -> float:
        return float(self._next("random"))

    def integers(self, lo: int, hi: int | None = None) -> int:
        return int(self._next("integers"))

    def rand(self, n: int) -> torch.Tensor:
        return self._t("rand").float()

    def randn(self, *shape: int) -> torch.Tensor:
        return self._t("randn").float().reshape(shape)

    def poisson(self, rate: torch.Tensor) -> torch.Tensor:
        return self._t("poisson").float().reshape(rate.shape)

    def uniform_tensor(self, n: int, lo: float, hi: float) -> torch.Tensor:
        return self._t("uniform_").float()

    def randperm(self, n: int) -> torch.Tensor:
        return self._t("randperm").long()

    def exhausted(self) -> bool:
        return self.pos == len(self.record)
