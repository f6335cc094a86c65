"""Input staging either side of `feed_data` (neosr/data/prefetch_dataloader.py, train.py:204-212):

* `PrefetchGenerator` / `PrefetchDataLoader` (`prefetch_mode = "cpu"`, reference lines 9-66): a daemon thread keeps up to
  `num_prefetch_queue` collated batches ready in a bounded queue;
* `DevicePrefetcher` (`prefetch_mode = "cuda"`: the reference's `CUDAPrefetcher`, lines 69-125, same `next()` / `reset()`
  interface) runs the next batch's host->HBM copies on a side HIP stream while the current iteration computes; `next()`
  makes the compute stream wait for that copy only.
"""

from __future__ import annotations

import queue
import threading
from typing import Any

import torch
from torch.utils.data import DataLoader

_END = object()   # end-of-epoch marker in the queue (a batch may legitimately be None)


class PrefetchGenerator(threading.Thread):
    """Iterator over `source` whose items are produced by a background thread, at most `num_prefetch_queue` ahead."""

    def __init__(self, source, num_prefetch_queue: int) -> None:
        super().__init__(daemon=True)
        self._q: queue.Queue[Any] = queue.Queue(maxsize=max(int(num_prefetch_queue), 1))
        self._source = source
        self._error: BaseException | None = None
        self.start()

    def run(self) -> None:
        try:
            for item in self._source:
                self._q.put(item)
        except BaseException as e:   # surfaced in the consumer, not lost in the thread
            self._error = e
        finally:
            self._q.put(_END)

    def __iter__(self):
        return self

    def __next__(self) -> Any:
        item = self._q.get()
        if item is _END:
            if self._error is not None:
                raise self._error
            raise StopIteration
        return item


class PrefetchDataLoader(DataLoader):
    """`torch.utils.data.DataLoader` whose iterator is wrapped in a `PrefetchGenerator` (same constructor contract as the
    reference: `num_prefetch_queue` first, the DataLoader arguments as keywords)."""

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
