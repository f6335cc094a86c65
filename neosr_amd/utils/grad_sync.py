"""Data-parallel gradient exchange for one network: a SUM all-reduce of its flat gradient arena over RCCL,
issued in a few large buckets WHILE backward is still running.

The reference gets this from `DistributedDataParallel` (neosr/models/base.py:140-146: 25 MB buckets reduced
as autograd hooks fire).  Here backward of the RRDB trunk is ONE host-side plan that walks the parameter arena
from its end to its start and records a caller-owned HIP event after chosen RRDBs
(`neosr_rrdbnet_backward_marked`): the suffix of the arena behind a mark is final once its event completes, so
a communication stream waits on the event and reduces that suffix while the earlier RRDBs are still computing.
xGMI is point-to-point (7 links, ring collectives are per-link bound): a handful of 15-25 MB messages, not
DDP's many small ones.  The 1/world_size is folded into the fused optimizer kernel (`set_grad_scale`).

Networks whose gradients arrive tensor by tensor (SwinIR, HAT, compact generators: layer-composed) get DDP's own
mechanism, rebuilt around the flat arena (`attach`): the parameter arena is cut into a few contiguous buckets;
post-accumulate hooks (and the deferred-reduction flush of hip/transformer.py, which writes `.grad` outside autograd)
count the parameters of a bucket that have their gradient.  The producers of the large gradients (Linear / Mlp weight +
bias, LayerNorm affine, 3x3 convolution weight + bias) ask `grad_slot()` for the parameter's slice of the persistent
gradient arena and write their result THERE, so a complete bucket is normally already in place; stragglers (small
tensors produced by other ops, accumulated gradients) are moved by one multi-tensor copy and `.grad` is re-pointed at the
arena.  Buckets leave in ONE order on every rank — highest arena offset first, i.e. backward order; a complete bucket is
held back until its predecessors have been sent — so ranks whose parameters became ready in different orders (or not at
all: a data-dependent branch) still issue identical collectives.  An event is recorded and the communication
stream all-reduces the slice while backward is still working on the earlier layers.  What is not complete when backward
returns (unused parameters: zeros) goes from `end_backward()`, in the same order.  The discriminator (two backward passes per
step) and accumulating / SAM steps are reduced after backward, but still asynchronously: `start()` only enqueues,
`finish()` is called right before that network's optimizer step, so the exchange of G overlaps the whole discriminator
phase and the exchange of D overlaps the generator's optimizer step.
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
        # hook-driven buckets of a layer-composed network (attach)
        self._params: list = []
        self._offs: list[int] = []
        self._total = 0
        self._arena: torch.Tensor | None = None
        self._bucket_of: dict[int, int] = {}          # id(param) -> bucket
        self._bucket_params: list[list[int]] = []     # bucket -> parameter indices
        self._bucket_range: list[tuple[int, int]] = []
        self._pending: list[int] = []
        self._sent: list[bool] = []
        self._next = -1                               # next bucket to leave (descending)
        self._seen: set[int] = set()
        self._taken: set[int] = set()
        self._slot_of: dict[int, torch.Tensor] = {}   # id(param) -> its view of the gradient arena
        self._live = False                            # a hook-driven exchange is in progress for this backward
        self.in_backward_buckets = 0                  # buckets issued from inside the last backward (tests / bench)

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
        self.in_backward_buckets = 0

    def reduce_suffix(self, lo: int, event_index: int | None) -> None:
        """[lo, previous lo) of the arena is final once event `event_index` has completed: reduce it."""
        assert self._flat is not None
        hi, self._lo = self._lo, lo
        if hi <= lo:
            return
        self.in_backward_buckets += 1   # (called by the marked backward plan: this suffix leaves while backward still runs)
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

    # -- hook-driven buckets (layer-composed networks) --------------------------------------------------
    def attach(self, params, n_buckets: int = 4) -> None:
        """Register the hooks on `params` (one network, arena order = `named_parameters()` order)."""
        from neosr_amd.hip.nets import arena_layout

        self._params = [p for p in params]
        offs, total = arena_layout(self._params)
        self._offs, self._total = list(offs), int(total)
        # contiguous buckets of roughly equal size, cut at parameter boundaries
        target = max(1, total // n_buckets)
        self._bucket_params, self._bucket_range, cur, lo = [], [], [], 0
        for i, p in enumerate(self._params):
            cur.append(i)
            end = self._offs[i + 1] if i + 1 < len(self._params) else total
            if end - lo >= target and len(self._bucket_params) < n_buckets - 1:
                self._bucket_params.append(cur)
                self._bucket_range.append((lo, end))
                cur, lo = [], end
        if cur:
            self._bucket_params.append(cur)
            self._bucket_range.append((lo, total))
        self._bucket_of = {id(self._params[i]): b for b, idx in enumerate(self._bucket_params) for i in idx}
        for p in self._params:
            if p.requires_grad:
                p.register_post_accumulate_grad_hook(self._on_grad)

    def _on_grad(self, p) -> None:
        if not self._live:
            return
        # a parameter with a reduction still queued in hip/transformer.py is reported by the flush (GRADS_READY), not by
        # the hook of a contribution that came through autograd: its slice must not leave while the flush can still add
        from neosr_amd.hip import transformer as _tr

        if _tr.has_pending(p):
            return
        self.grads_ready((p,))

    def arm_backward(self) -> None:
        """Call right before the backward whose gradients will be stepped (model: `armed` and hooks attached)."""
        if not self._params or not self.armed:
            self._live = False
            return
        self.finish()
        if self._arena is None or self._arena.device != self._params[0].device:
            self._arena = torch.zeros(self._total, device=self._params[0].device, dtype=torch.float32)
            self._slot_of = {id(p): self._arena[o : o + p.numel()].view_as(p) for p, o in zip(self._params, self._offs)}
        self._flat, self._lo, self.buckets = self._arena, self._total, []
        self._pending = [sum(1 for i in idx if self._params[i].requires_grad) for idx in self._bucket_params]
        self._sent = [False] * len(self._bucket_params)
        self._next = len(self._bucket_params) - 1
        self._seen, self._taken = set(), set()
        self._live, self.in_backward_buckets = True, 0
        self.armed = True

    def grads_ready(self, leaves) -> None:
        """`.grad` of these parameters is final for this backward (hook, or the deferred-reduction flush)."""
        if not self._live:
            return
        for p in leaves:
            b = self._bucket_of.get(id(p))
            # (the engine also runs the post-accumulate hook of a leaf whose Function returned None — the deferred
            # reductions deliver that gradient later, outside autograd —: a notification counts once, and only when
            # the gradient is there)
            if b is None or p.grad is None or id(p) in self._seen:
                continue
            self._seen.add(id(p))
            self._pending[b] -= 1
        self._send_in_order(False)

    def grad_slot(self, p) -> torch.Tensor | None:
        """Where the FIRST gradient contribution of `p` in this backward may be written directly: its view of the gradient
        arena (shape of `p`), or None (no hook-driven exchange in progress, `p` not ours, or `p` already has a gradient —
        a second contribution accumulates through autograd as usual)."""
        if not p.is_leaf:   # (derived weights — spectral norm, the expanded 4x4/s2 kernel — have no slice; reading .grad of a
            return None     # non-leaf would also emit autograd's UserWarning)
        if not self._live or p.grad is not None or id(p) in self._seen or id(p) in self._taken:
            return None
        slot = self._slot_of.get(id(p))
        if slot is None:
            return None
        self._taken.add(id(p))   # handed out once per backward: a second producer of the same parameter gets None
        # a FRESH view per call: AccumulateGrad adopts a gradient only when nobody else references the tensor object — the
        # cached view would be cloned and copied back onto the identical slice by _send_bucket
        return slot.view_as(slot)

    def _send_in_order(self, force: bool) -> None:
        while self._next >= 0 and (force or self._pending[self._next] == 0):
            b = self._next
            self._next -= 1
            if not self._sent[b]:
                self._send_bucket(b)
                if not force:
                    self.in_backward_buckets += 1

    def _send_bucket(self, b: int) -> None:
        lo, hi = self._bucket_range[b]
        if self._arena.is_cuda:
            # the transformer block plans finish their parameter gradients on a library stream (hip/transformer.py:
            # join_tails): this stream waits for them before the bucket's gradients are copied / handed to the exchange
            from neosr_amd.hip import transformer as _tr

            _tr.join_tails()
        dst, src = [], []
        with torch.no_grad():
            for i in self._bucket_params[b]:
                p = self._params[i]
                slot = self._slot_of[id(p)]
                g = p.grad
                if g is None:            # unused in this backward: its slice may hold an earlier step's gradient
                    slot.zero_()
                elif g.data_ptr() != slot.data_ptr() or not g.is_contiguous():
                    dst.append(slot)
                    src.append(g)
            if dst:
                torch._foreach_copy_(dst, src)  # noqa: SLF001  (one multi-tensor launch for the stragglers)
            for i in self._bucket_params[b]:  # the optimizer and the all-reduce see ONE buffer: no second copy
                p = self._params[i]
                p.grad = self._slot_of[id(p)]
        self._sent[b] = True
        event = None
        if self.comm is not None and self._arena.is_cuda:
            event = torch.cuda.Event()
            event.record()
        self._issue(lo, hi, event)

    def end_backward(self) -> bool:
        """After the armed backward: send what the hooks left (parameters without a gradient).  True if this network's
        exchange was hook-driven (then `start()` has nothing left to do)."""
        if not self._live:
            return False
        self._send_in_order(True)
        self._live, self._lo = False, 0
        return True

    # -- model side -----------------------------------------------------------------------------------
    def start(self, flat: torch.Tensor, bucket_elems: int = 16 << 20) -> None:
        """Enqueue the reduction of whatever part of `flat` is not in flight yet (all of it for layer-composed
        networks), in buckets of at most `bucket_elems`, after the work already on the compute stream."""
        if self._flat is None or self._flat.data_ptr() != flat.data_ptr() or self._flat.numel() != flat.numel():
            self.finish()
            self._flat, self._lo, self.buckets = flat, flat.numel(), []
            self.in_backward_buckets = 0   # (nothing of this arena left during backward; else the count of the pass stands)
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
        self._live = False
