"""Input prefetching either side of `feed_data` (neosr/data/prefetch_dataloader.py):

* `PrefetchGenerator` / `PrefetchDataLoader` — a daemon thread keeps `num_prefetch_queue` batches ready;
* `DevicePrefetcher` (the reference's `CUDAPrefetcher`, same interface: `next()` / `reset()`): the next
  batch's host->HBM copies run on a side HIP stream while the current iteration computes; `next()` makes
  the compute stream wait for that copy only.
"""

from __future__ import annotations

import queue
from collections.abc import Iterator
from threading import Thread
from typing import Any

import torch
from torch.utils.data import DataLoader


class PrefetchGenerator(Thread):
    def __init__(self, generator, num_prefetch_queue: int) -> None:
        super().__init__(daemon=True)
        self.queue: queue.Queue[Any] = queue.Queue(num_prefetch_queue)
        self.generator = generator
        self.start()

    def run(self) -> None:
        for item in self.generator:
            self.queue.put(item)
        self.queue.put(None)  # end marker

    def __next__(self) -> Any:
        item = self.queue.get()
        if item is None:
            raise StopIteration
        return item

    def __iter__(self) -> Iterator:
        return self


class PrefetchDataLoader(DataLoader):
    def __init__(self, num_prefetch_queue: int, **kwargs) -> None:
        self.num_prefetch_queue = num_prefetch_queue
        super().__init__(**kwargs)

    def __iter__(self):
        return PrefetchGenerator(super().__iter__(), self.num_prefetch_queue)


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
