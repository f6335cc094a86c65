"""Autograd bridges from torch modules to the whole-network HIP plans, and the flat HBM arenas
(parameters / gradients) the fused optimizer and the RCCL all-reduce operate on.
"""

from __future__ import annotations

import contextlib
import ctypes as C
import os

import torch
from torch import nn

from neosr_amd import _C


# --------------------------------------------------------------------------------------------
# flat arenas
# --------------------------------------------------------------------------------------------
ALIGN = 4  # every tensor of an arena starts on a 16-byte boundary (float4 loads in the kernels)


def arena_layout(tensors) -> tuple[list[int], int]:
    """(element offset of each tensor, total elements) of a flat arena holding `tensors` in order,
    each start rounded up to ALIGN elements.  Identical to a packed layout when every numel is a
    multiple of ALIGN (esrgan, compact); pads exist e.g. after SwinIR's 225x6 bias tables."""
    offs, off = [], 0
    for t in tensors:
        offs.append(off)
        off += (t.numel() + ALIGN - 1) // ALIGN * ALIGN
    return offs, off


def parameter_slots(module: nn.Module) -> list:
    """[(owning submodule, name)] of every parameter in `module.parameters()` order.  The walk over the module tree
    (`named_modules` recursion: ~0.1 ms for compact, several ms for hat_l) is done ONCE per module object and kept; the
    tensors are looked up through the slots on every use, so a re-assigned `nn.Parameter` is still seen.  Adding or
    removing submodules / parameters after the first call needs `module._neosr_slots = None`."""
    slots = module.__dict__.get("_neosr_slots")
    if slots is None:
        seen, slots = set(), []
        for m in module.modules():
            for name, prm in m._parameters.items():  # noqa: SLF001
                if prm is not None and id(prm) not in seen:
                    seen.add(id(prm))
                    slots.append((m, name))
        module.__dict__["_neosr_slots"] = slots
    return slots


def parameters_of(module: nn.Module) -> list:
    """`list(module.parameters())` without the tree walk (see `parameter_slots`)."""
    return [m._parameters[name] for m, name in parameter_slots(module)]  # noqa: SLF001


def flatten_parameters_(module: nn.Module, device=None) -> torch.Tensor:
    """Re-home every parameter of ``module`` into one contiguous fp32 arena (in named_parameters
    order, 16-byte aligned starts, zero pads) and make each ``nn.Parameter`` a view of it.
    Idempotent; returns the arena.  `device`: gather where the parameters are, then move the arena there in
    ONE copy (a host-built network reaches the GPU with one transfer instead of one per tensor)."""
    params = parameters_of(module)
    if not params:
        raise ValueError("module has no parameters")
    arena = getattr(module, "_neosr_arena", None)
    if arena is not None and (device is None or arena.device == torch.device(device)):
        # the steady-state call (once or twice per training step): one pointer comparison per parameter against the
        # offsets computed when the arena was built
        offs = module.__dict__.get("_neosr_offs")
        if offs is not None and len(offs) == len(params) + 1 and offs[-1] == arena.numel():
            base = arena.data_ptr()
            for prm, off in zip(params, offs):
                if prm.data_ptr() != base + 4 * off:
                    break
            else:
                return arena
        elif _is_flat(params, arena):
            return arena
    offs, total = arena_layout(params)
    with torch.no_grad():
        # ONE gather (a `cat` with zero pads), not a copy per parameter: model construction used to issue ~2 100 (esrgan) /
        # ~8 500 (hat_l config) small device copies that show up in whole-process kernel statistics
        parts = []
        for i, p in enumerate(params):
            parts.append(p.data.reshape(-1).to(torch.float32))
            end = offs[i + 1] if i + 1 < len(params) else total
            pad = end - offs[i] - p.numel()
            if pad:
                parts.append(torch.zeros(pad, device=p.device, dtype=torch.float32))
        arena = torch.cat(parts)
        if device is not None:
            arena = arena.to(device)
        for p, off in zip(params, offs):
            p.data = arena[off : off + p.numel()].view(p.shape)
    module._neosr_arena = arena  # noqa: SLF001
    module.__dict__["_neosr_offs"] = [*offs, total]
    return arena


def _is_flat(tensors, arena: torch.Tensor) -> bool:
    offs, total = arena_layout(tensors)
    base = arena.data_ptr()
    for t, off in zip(tensors, offs):
        if t is None or t.data_ptr() != base + 4 * off or not t.is_contiguous():
            return False
    return total == arena.numel()


def flat_view_of(tensors) -> torch.Tensor | None:
    """If `tensors` sit at their `arena_layout` offsets of one buffer, return that buffer as a 1-D
    tensor (no copy); else None."""
    tensors = list(tensors)
    t0 = tensors[0]
    if t0 is None:
        return None
    offs, total = arena_layout(tensors)
    base = t0.data_ptr()
    for t, off in zip(tensors, offs):
        if t is None or t.data_ptr() != base + 4 * off or not t.is_contiguous():
            return None
    if t0.untyped_storage().nbytes() < (t0.storage_offset() + total) * 4:
        return None
    flat = torch.empty(0, device=t0.device, dtype=torch.float32)
    flat.set_(t0.untyped_storage(), t0.storage_offset(), (total,), (1,))
    return flat


def group_arena(cache: dict, plist: list):
    """(flat view of the parameters | None, [parameters with requires_grad]) of an optimizer group.  The fused optimizer
    steps ask this once per group and step; for a layer-composed generator (hat_l: 1 710 parameters) building `.data` views,
    the layout and the contiguity checks from scratch cost ~2 ms per call (tools/host_overhead.py, HOST_PROFILE=1).  The
    layout of a group is cached on the optimizer (`cache`, keyed by the group's own list object); the steady-state call
    is one pass of `requires_grad` + pointer compares against it.  Anything that does not match falls back to the full
    check (a re-homed parameter, a changed `requires_grad`)."""
    ent = cache.get(id(plist))
    if ent is not None and ent[5] is plist and ent[0] == len(plist):   # (the entry keeps the list alive: its id is not re-used)
        _n, params, offs, total, n_frozen, _keep = ent
        base = params[0].data_ptr()
        ok = True
        for prm, off in zip(params, offs):
            if prm.data_ptr() != base + 4 * off or not prm.requires_grad:
                ok = False
                break
        if ok:
            # the list must still hold exactly these objects (a parameter REPLACED in the group with the length unchanged
            # would otherwise keep the old tensors alive and the fused step would go on updating the old arena), and none
            # of its frozen members may have become trainable: one pass over the list
            k = 0
            n_par = len(params)
            for prm in plist:
                if prm.requires_grad:
                    if k >= n_par or prm is not params[k]:
                        ok = False
                        break
                    k += 1
            ok = ok and k == n_par
        if ok:
            t0 = params[0]
            if t0.untyped_storage().nbytes() >= (t0.storage_offset() + total) * 4:
                flat = torch.empty(0, device=t0.device, dtype=torch.float32)
                flat.set_(t0.untyped_storage(), t0.storage_offset(), (total,), (1,))
                return flat, params
    params = [prm for prm in plist if prm.requires_grad]
    flat = flat_view_of([prm.data for prm in params]) if params else None
    if flat is not None:
        offs, total = arena_layout(params)
        cache[id(plist)] = (len(plist), params, offs, total, len(plist) - len(params), plist)
    else:
        cache.pop(id(plist), None)
    return flat, params


def flat_grad_of(params) -> torch.Tensor | None:
    """If all ``p.grad`` sit in one buffer at the arena layout (as our plans emit them), return that
    buffer as a 1-D tensor without copying; else None."""
    return flat_view_of([p.grad for p in params])


_PADS: dict = {}


def pack_grads(params) -> torch.Tensor:
    """Gradients produced tensor by tensor (layer-composed networks) -> one flat arena in the
    parameter layout, zero pads included: a single `cat`."""
    params = list(params)
    offs, total = arena_layout(params)
    parts = []
    for i, p in enumerate(params):
        g = p.grad.reshape(-1)
        parts.append(g)
        end = offs[i + 1] if i + 1 < len(params) else total
        pad = end - offs[i] - g.numel()
        if pad:
            key = (pad, g.device)
            if key not in _PADS:
                _PADS[key] = torch.zeros(pad, device=g.device, dtype=torch.float32)
            parts.append(_PADS[key])
    return torch.cat(parts)


def _alloc_flat_grads(params):
    offs, total = arena_layout(params)
    sizes = [p.numel() for p in params]
    packed = total == sum(sizes)
    flat = (torch.empty if packed else torch.zeros)(total, device=params[0].device, dtype=torch.float32)
    if packed:  # one split instead of a slice per parameter (the views are made on every backward pass)
        views = [v.view(p.shape) for v, p in zip(flat.split_with_sizes(sizes), params)]
    else:
        views = [flat[off : off + n].view(p.shape) for p, off, n in zip(params, offs, sizes)]
    return flat, views


# --------------------------------------------------------------------------------------------
# direct parameter gradients (opt-in)
# --------------------------------------------------------------------------------------------
# A plan network hands autograd one tensor per parameter; per backward pass that is a fresh view per parameter (the
# engine only adopts a gradient nobody else references — a cached view would be CLONED) and one AccumulateGrad node per
# parameter: ~1.9 + ~2.5 us each on the host.  For SRVGGNetCompact (68 parameters, ~0.5 ms of device work per step at
# batch 2) that was 0.3 ms of a 0.8 ms step, which is host-bound.  Inside `direct_param_grads()` — the models put it
# around their own forward + backward (models/image.py), nothing is switched on process-wide — the plan's backward
# writes into ONE persistent gradient arena per network and assigns the cached views to `.grad` itself; autograd sees a
# single 1-element anchor leaf instead of the parameters.  Consequences inside the scope: `torch.autograd.grad(loss,
# params)` does not see these parameters, parameter hooks do not fire, and `.grad` of step k + 1 lives in the memory
# `.grad` of step k lived in (the optimizer has consumed it by then; `zero_grad(set_to_none=True)` or not).  A backward
# pass that finds `.grad` already set (accumulation) adds into it from a temporary arena.  NEOSR_AMD_DIRECT_GRADS=0
# switches the scope off (A/B).
_DIRECT_ENV = os.environ.get("NEOSR_AMD_DIRECT_GRADS")
_DIRECT_SCOPE = 0


@contextlib.contextmanager
def direct_param_grads(on: bool = True):
    global _DIRECT_SCOPE
    if not on or _DIRECT_ENV == "0":
        yield
        return
    _DIRECT_SCOPE += 1
    try:
        yield
    finally:
        _DIRECT_SCOPE -= 1


class DirectGrads:
    """Per-network state of the direct hand-off: parameter list + pointer table, persistent gradient arena + views +
    pointer table, the anchor leaf that ties the plan's output to autograd."""

    __slots__ = ("key", "params", "ptab", "anchor", "flat", "views", "gtab")

    # a cache, not state: a copied / pickled network builds its own on first use (the pointer tables are ctypes arrays,
    # which cannot be pickled, and the arenas belong to the original's parameters)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_direct_state, ())


def _no_direct_state():
    return None


def direct_state(module: nn.Module, params: list):
    """The network's DirectGrads inside a `direct_param_grads()` scope (built on first use, rebuilt when the parameter
    arena moved), else None.  `params` = the flat parameter list (`HipNet._plan_params()`, flatness just checked)."""
    if _DIRECT_SCOPE <= 0 or not torch.is_grad_enabled() or not params[0].is_cuda:
        return None
    for prm in params:
        if not prm.requires_grad:
            return None
    key = (params[0].data_ptr(), len(params))
    st = module.__dict__.get("_neosr_direct")
    if st is None or st.key != key:
        st = DirectGrads()
        st.key = key
        st.params = list(params)
        st.ptab = _C.ptr_table(st.params)
        st.anchor = torch.zeros(1, device=params[0].device, dtype=torch.float32, requires_grad=True)
        st.flat, st.views = _alloc_flat_grads(st.params)
        st.gtab = _C.ptr_table(st.views)
        module.__dict__["_neosr_direct"] = st
    return st


def _direct_targets(st: DirectGrads):
    """(flat, views, pointer table, accumulate) a direct backward pass writes to: the persistent arena, or a temporary
    one when `.grad` is already populated."""
    if st.params[0].grad is None:
        return st.flat, st.views, st.gtab, False
    flat, views = _alloc_flat_grads(st.params)
    return flat, views, _C.ptr_table(views), True


def _direct_assign(st: DirectGrads, views, accumulate: bool) -> None:
    if accumulate:
        torch._foreach_add_([prm.grad for prm in st.params], views)  # noqa: SLF001
    else:
        for prm, v in zip(st.params, views):
            prm.grad = v


# --------------------------------------------------------------------------------------------
# RRDBNet
# --------------------------------------------------------------------------------------------
class RRDBNetFunction(torch.autograd.Function):
    """y = RRDBNet(x; params) on the HIP plan ``neosr_rrdbnet_forward/backward``."""

    @staticmethod
    def forward(ctx, x, hp: dict, *params):
        lib = _C.load()
        _C.require_device(x, "input")
        for p in params:
            _C.require_device(p, "parameter")
        x = x.contiguous()
        B, cin, H, W = x.shape
        training = bool(hp["training"]) and any(ctx.needs_input_grad)
        cfg = _C.RRDBNetCfg(B, H, W, cin, hp["num_out_ch"], hp["num_feat"], hp["num_block"],
                            hp["num_grow_ch"], int(training))
        nexp = lib.neosr_rrdbnet_num_params(C.byref(cfg))
        if nexp != len(params):
            raise _C.NeosrAmdError(f"rrdbnet expects {nexp} parameter tensors, got {len(params)}")
        nbytes = lib.neosr_rrdbnet_workspace_bytes(C.byref(cfg))
        if nbytes < 0:
            _C.check(1, "neosr_rrdbnet_workspace_bytes")
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        y = torch.empty(B, hp["num_out_ch"], 4 * H, 4 * W, device=x.device, dtype=torch.float32)
        ptab = _C.ptr_table(params)
        _C.check(lib.neosr_rrdbnet_forward(C.byref(cfg), ptab, x.data_ptr(), y.data_ptr(),
                                           ws.data_ptr(), _C.stream_ptr()), "neosr_rrdbnet_forward")
        if training:
            ctx.cfg = cfg
            ctx.ws = ws
            ctx.x = x
            ctx.params = params
            ctx.sync = hp.get("sync")
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _C.load()
        params = ctx.params
        gy = gy.contiguous()
        flat, gviews = _alloc_flat_grads(params)
        gx = torch.empty_like(ctx.x) if ctx.needs_input_grad[0] else None
        ptab = _C.ptr_table(params)
        gtab = _C.ptr_table(gviews)
        sync = ctx.sync
        if sync is not None and sync.armed and all(ctx.needs_input_grad[2:]):
            # data-parallel: buckets of the gradient arena are all-reduced while the earlier RRDBs are still in
            # backward (neosr_amd/utils/grad_sync.py); the head of the arena goes when the model calls start()
            blocks = sync.mark_blocks(ctx.cfg.num_block)
            handles = sync.event_handles(len(blocks))
            barr = (C.c_int32 * max(1, len(blocks)))(*blocks)
            earr = (C.c_void_p * max(1, len(blocks)))(*handles)
            _C.check(lib.neosr_rrdbnet_backward_marked(C.byref(ctx.cfg), ptab, gtab, gy.data_ptr(),
                                                       None if gx is None else gx.data_ptr(), ctx.ws.data_ptr(),
                                                       _C.stream_ptr(), len(blocks), barr, earr),
                     "neosr_rrdbnet_backward_marked")
            offs = arena_layout(params)[0]
            sync.begin(flat)
            for i, b in enumerate(blocks):
                sync.reduce_suffix(offs[2 + 30 * b], i)
        else:
            _C.check(lib.neosr_rrdbnet_backward(C.byref(ctx.cfg), ptab, gtab, gy.data_ptr(),
                                                None if gx is None else gx.data_ptr(),
                                                ctx.ws.data_ptr(), _C.stream_ptr()),
                     "neosr_rrdbnet_backward")
        ctx.ws = None
        grads = tuple(g if need else None for g, need in zip(gviews, ctx.needs_input_grad[2:]))
        return (gx, None, *grads)


# --------------------------------------------------------------------------------------------
# SRVGGNetCompact
# --------------------------------------------------------------------------------------------
class CompactFunction(torch.autograd.Function):
    """y = SRVGGNetCompact(x; params) on ``neosr_compact_forward/backward``.  `hp["direct"]` (a DirectGrads, see
    `direct_param_grads`): the only tensor behind `hp` is then the anchor leaf and backward assigns `.grad` itself."""

    @staticmethod
    def forward(ctx, x, hp: dict, *params):
        lib = _C.load()
        _C.require_device(x, "input")
        st = hp.get("direct")
        if st is not None:
            params, ptab = st.params, st.ptab
        else:
            for p in params:
                _C.require_device(p, "parameter")
            ptab = _C.ptr_table(params)
        x = x.contiguous()
        B, cin, H, W = x.shape
        training = bool(hp["training"]) and any(ctx.needs_input_grad)
        cfg = _C.CompactCfg(B, H, W, cin, hp["num_out_ch"], hp["num_feat"], hp["num_conv"],
                            hp["upscale"], hp["act_type"], int(training))
        nexp = lib.neosr_compact_num_params(C.byref(cfg))
        if nexp != len(params):
            raise _C.NeosrAmdError(f"compact expects {nexp} parameter tensors, got {len(params)}")
        nbytes = lib.neosr_compact_workspace_bytes(C.byref(cfg))
        if nbytes < 0:
            _C.check(1, "neosr_compact_workspace_bytes")
        ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
        r = hp["upscale"]
        y = torch.empty(B, hp["num_out_ch"], r * H, r * W, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_compact_forward(C.byref(cfg), ptab, x.data_ptr(), y.data_ptr(),
                                           ws.data_ptr(), _C.stream_ptr()), "neosr_compact_forward")
        if training:
            ctx.cfg = cfg
            ctx.ws = ws
            ctx.params = params
            ctx.direct = st
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _C.load()
        if ctx.needs_input_grad[0]:
            raise _C.NeosrAmdError("compact: gradient w.r.t. the input image is not implemented")
        params = ctx.params
        gy = gy.contiguous()
        st = ctx.direct
        if st is not None:
            _flat, gviews, gtab, acc = _direct_targets(st)
            ptab = st.ptab
        else:
            _flat, gviews = _alloc_flat_grads(params)
            ptab = _C.ptr_table(params)
            gtab = _C.ptr_table(gviews)
        _C.check(lib.neosr_compact_backward(C.byref(ctx.cfg), ptab, gtab, gy.data_ptr(), None,
                                            ctx.ws.data_ptr(), _C.stream_ptr()),
                 "neosr_compact_backward")
        ctx.ws = None
        if st is not None:
            _direct_assign(st, gviews, acc)
            return (None, None, None)
        grads = tuple(g if need else None for g, need in zip(gviews, ctx.needs_input_grad[2:]))
        return (None, None, *grads)


# --------------------------------------------------------------------------------------------
# L1 loss
# --------------------------------------------------------------------------------------------
class L1LossFunction(torch.autograd.Function):
    """loss_weight * mean(|pred - target|) with a fixed-order two-stage reduction."""

    @staticmethod
    def forward(ctx, pred, target, loss_weight: float):
        lib = _C.load()
        _C.require_device(pred, "pred")
        _C.require_device(target, "target")
        pred = pred.contiguous()
        target = target.contiguous()
        if pred.shape != target.shape:
            raise _C.NeosrAmdError(f"L1Loss: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
        n = pred.numel()
        out = torch.empty((), device=pred.device, dtype=torch.float32)
        ws = torch.empty(4096, device=pred.device, dtype=torch.float32)
        _C.check(lib.neosr_l1_loss_fwd(pred.data_ptr(), target.data_ptr(), n, float(loss_weight),
                                       out.data_ptr(), ws.data_ptr(), _C.stream_ptr()),
                 "neosr_l1_loss_fwd")
        ctx.save_for_backward(pred, target)
        ctx.loss_weight = float(loss_weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _C.load()
        pred, target = ctx.saved_tensors
        gout = gout.contiguous().to(torch.float32)
        gp = torch.empty_like(pred)
        _C.check(lib.neosr_l1_loss_bwd(pred.data_ptr(), target.data_ptr(), gout.data_ptr(),
                                       pred.numel(), ctx.loss_weight, gp.data_ptr(),
                                       _C.stream_ptr()), "neosr_l1_loss_bwd")
        gt = -gp if ctx.needs_input_grad[1] else None
        return gp, gt, None
