"""Input staging in front of `feed_data` (neosr/data/prefetch_dataloader.py:69-125, the `prefetch_mode = "cuda"` path of
train.py:204-212): `DevicePrefetcher` (the reference's `CUDAPrefetcher`, same interface: `next()` / `reset()`) runs the
next batch's host->HBM copies on a side HIP stream while the current iteration computes; `next()` makes the compute
stream wait for that copy only.  The reference's CPU-side thread prefetcher (`prefetch_mode = "cpu"`) is host data-loader
plumbing outside SURVEY §8 and is not restated here: any iterable of batch dicts (a plain `torch.utils.data.DataLoader`
with workers) feeds this class.
"""

from __future__ import annotations

from typing import Any

import torch


class DevicePrefetcher:
    def __init__(self, loader, opt: dict[str, Any], device: str | torch.device = "cuda") -> None:
        self.ori_loader = loader
        self.loader = iter(loader)
        self.opt = opt
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DevicePrefetcher stages batches into HBM: it needs a HIP device")
        self.stream = torch.cuda.Stream(self.device)
        self.preload()

    def preload(self) -> None:
        try:
            self.batch = next(self.loader)  # a dict
        except StopIteration:
            self.batch = None
            return
        with torch.cuda.stream(self.stream):
            for k, v in self.batch.items():
                if torch.is_tensor(v):
                    self.batch[k] = v.to(device=self.device, non_blocking=True)

    def next(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        batch = self.batch
        if batch is not None:  # the consumer's stream now owns these buffers
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(torch.cuda.current_stream(self.device))
        self.preload()
        return batch

    def reset(self) -> None:
        self.loader = iter(self.ori_loader)
        self.preload()


CUDAPrefetcher = DevicePrefetcher  # the reference's name (train.py:204-212 builds it for prefetch_mode = "cuda")
