"""Data-parallel gradient exchange for one network: a SUM all-reduce of its flat gradient arena over RCCL,
issued in a few large buckets WHILE backward is still running.

The reference gets this from `DistributedDataParallel` (neosr/models/base.py:140-146: 25 MB buckets reduced
as autograd hooks fire).  Here backward of the RRDB trunk is ONE host-side plan that walks the parameter arena
from its end to its start and records a caller-owned HIP event after chosen RRDBs
(`neosr_rrdbnet_backward_marked`): the suffix of the arena behind a mark is final once its event completes, so
a communication stream waits on the event and reduces that suffix while the earlier RRDBs are still computing.
xGMI is point-to-point (7 links, ring collectives are per-link bound): a handful of 15-25 MB messages, not
DDP's many small ones.  The 1/world_size is folded into the fused optimizer kernel (`set_grad_scale`).

Networks whose gradients arrive tensor by tensor (U-Net-SN, SwinIR, HAT: layer-composed) are reduced after
their backward, but still asynchronously: `start()` only enqueues, `finish()` is called right before that
network's optimizer step, so the exchange of G overlaps the whole discriminator phase and the exchange of D
overlaps the generator's optimizer step.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, n_marks: int = 2, device: torch.device | str | None = None) -> None:
        self.n_marks = n_marks
        self.cuda = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
        self.comm = torch.cuda.Stream() if self.cuda else None
        self.events: list = []
        if self.cuda:
            for _ in range(n_marks):
                e = torch.cuda.Event()
                e.record()  # materialises the hipEvent_t so that its handle can be handed to the plan
                self.events.append(e)
        self.armed = False          # set by the model for the backward whose gradients will be stepped
        self._works: list = []
        self._flat: torch.Tensor | None = None
        self._lo = 0                # [self._lo, numel) of self._flat is already being reduced
        self.buckets: list[tuple[int, int]] = []  # (lo, hi) element ranges of the last exchange, in issue order

    # -- plan side ------------------------------------------------------------------------------------
    def mark_blocks(self, num_block: int) -> list[int]:
        """RRDB indices after which a bucket may go, descending (backward order): equal thirds of the trunk."""
        cuts = sorted({(num_block * (i + 1)) // (self.n_marks + 1) for i in range(self.n_marks)}, reverse=True)
        return [c for c in cuts if 0 < c < num_block]

    def event_handles(self, n: int) -> list[int]:
        return [int(e.cuda_event) for e in self.events[:n]]

    def begin(self, flat: torch.Tensor) -> None:
        self.finish()
        self._flat, self._lo, self.buckets = flat, flat.numel(), []

    def reduce_suffix(self, lo: int, event_index: int | None) -> None:
        """[lo, previous lo) of the arena is final once event `event_index` has completed: reduce it."""
        assert self._flat is not None
        hi, self._lo = self._lo, lo
        if hi <= lo:
            return
        self._issue(lo, hi, self.events[event_index] if event_index is not None and self.cuda else None)

    def _issue(self, lo: int, hi: int, event) -> None:
        chunk = self._flat[lo:hi]
        self.buckets.append((lo, hi))
        if self.comm is not None and chunk.is_cuda:
            if event is None:  # everything enqueued so far on the compute stream
                event = torch.cuda.Event()
                event.record()
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(event)
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

    # -- model side -----------------------------------------------------------------------------------
    def start(self, flat: torch.Tensor, bucket_elems: int = 16 << 20) -> None:
        """Enqueue the reduction of whatever part of `flat` is not in flight yet (all of it for layer-composed
        networks), in buckets of at most `bucket_elems`, after the work already on the compute stream."""
        if self._flat is None or self._flat.data_ptr() != flat.data_ptr() or self._flat.numel() != flat.numel():
            self.finish()
            self._flat, self._lo, self.buckets = flat, flat.numel(), []
        hi = self._lo
        self._lo = 0
        event = None
        if self.comm is not None and flat.is_cuda and hi > 0:
            event = torch.cuda.Event()
            event.record()
        while hi > 0:
            lo = max(0, hi - bucket_elems)
            self._issue(lo, hi, event)
            hi = lo

    def finish(self) -> None:
        """Make the compute stream wait for every bucket (call right before the optimizer step)."""
        for w in self._works:
            w.wait()
        self._works = []
        self._flat = None
        self.armed = False
