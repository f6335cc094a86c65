"""Differentiable fronts for the transformer generators (SwinIR): fp32 MFMA GEMM Linear / Mlp with
fused bias / exact GELU / DropPath row scale / residual epilogues, LayerNorm, (shifted-)window
attention, channels-last PixelShuffle.  Tokens are channels-last pixels, i.e. the (B, H*W, C) token
matrix of neosr/archs/swinir_arch.py is the (B, H, W, C) HBM buffer itself.

torch is used for allocation and the autograd graph only; every FLOP runs in libneosr_amd.so.
"""

from __future__ import annotations

import contextlib
import os

import torch

from neosr_amd import _C


def _st():
    return _C.stream_ptr()


def _p(t):
    return None if t is None else t.data_ptr()


def _new(shape, like):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def gemm(mode, A, B, M, N, K, *, out=None, bias=None, res=None, aux_in=None, aux_out=None, row_scale=None,
         rows_per_scale=0, gelu=False, accumulate=False, colsum_a=None, defer_to=None):
    """C = op(A) op(B) with the fused epilogue of `neosr_gemm` (include/neosr_amd.h). Dense operands.
    TN only: `colsum_a` (M floats) also receives sum_k A[k, :] (the bias gradient)."""
    lib = _C.load()
    if out is None:
        out = _new((M, N), A)
    lda = K if mode != _C.GEMM_TN else M
    ldb = K if mode == _C.GEMM_NT else N
    d = _C.GemmDesc(A=A.data_ptr(), B=B.data_ptr(), C=out.data_ptr(), bias=_p(bias), res=_p(res),
                    aux_in=_p(aux_in), aux_out=_p(aux_out), row_scale=_p(row_scale), workspace=None,
                    M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=N, ldres=N, ldaux=N, rows_per_scale=rows_per_scale,
                    mode=mode, gelu=int(gelu), accumulate=int(accumulate), colsum_a=_p(colsum_a))
    ws = None
    if mode == _C.GEMM_TN:
        ws = torch.empty(lib.neosr_gemm_workspace_bytes(d) // 4, device=A.device, dtype=torch.float32)
        d.workspace = ws.data_ptr()
    if defer_to is not None:  # TN: queue the split reduction (see _defer_colsum); `out` must be _wgrad_pair's dW
        d.accumulate = 2
        rc = lib.neosr_gemm(d, _st())
        if rc >= 0:
            _C.check(rc or 1, "neosr_gemm")
        slab = M * N + (M if colsum_a is not None else 0)
        _defer_colsum(ws, -rc, slab, slab, out, defer_to)
        return out
    _C.check(lib.neosr_gemm(d, _st()), "neosr_gemm")
    return out


def colsum(x2d):
    lib = _C.load()
    rows, cols = x2d.shape
    out = _new((cols,), x2d)
    ws = _new((256 * cols,), x2d)
    _C.check(lib.neosr_colsum(x2d.data_ptr(), out.data_ptr(), ws.data_ptr(), rows, cols, cols, 0, _st()),
             "neosr_colsum")
    return out


def row_scale(x2d, scale, rows_per_scale):
    lib = _C.load()
    out = torch.empty_like(x2d)
    _C.check(lib.neosr_row_scale(x2d.data_ptr(), scale.data_ptr(), out.data_ptr(), x2d.shape[0], x2d.shape[1],
                                 rows_per_scale, _st()), "neosr_row_scale")
    return out


# Set by the model on data-parallel runs (GradSync.grad_slot): where the first gradient contribution of a parameter may
# be written directly — its slice of the gradient arena the all-reduce and the optimizer work on — instead of a fresh
# tensor that has to be packed later.  None / returning None: allocate as usual.
GRAD_SLOT = None


def _pair_slots(first, second, n_first: int, n_second: int):
    """arena slices of two parameters whose gradients one kernel writes back to back (weight | bias, gamma | beta), as
    one flat (n_first + n_second) view — or None when they are not adjacent slots"""
    if GRAD_SLOT is None or first is None:
        return None
    a = GRAD_SLOT(first)
    if a is None or not a.is_contiguous() or a.numel() != n_first:
        return None
    if second is None:
        return a.view(-1)
    b = GRAD_SLOT(second)
    if b is None or b.numel() != n_second or b.data_ptr() != a.data_ptr() + 4 * n_first:
        return None
    flat = torch.empty(0, device=a.device, dtype=torch.float32)
    flat.set_(a.untyped_storage(), a.storage_offset(), (n_first + n_second,), (1,))
    return flat


def _wgrad_pair(like, n_out: int, k_in: int, want_bias: bool, wl=None, bl=None):
    """(dW (n_out, k_in), db (n_out)) in one buffer, db right behind dW: the TN GEMM then finishes both with a
    single split-K reduction pass.  `wl` / `bl`: the leaves the gradients are for (GRAD_SLOT)."""
    buf = _pair_slots(wl, bl if want_bias else None, n_out * k_in, n_out)
    if buf is None:
        buf = _new((n_out * k_in + (n_out if want_bias else 0),), like)
    return buf[: n_out * k_in].view(n_out, k_in), (buf[n_out * k_in:] if want_bias else None)


# ------------------------------------------------------------------------------------------------
# Deferred parameter-gradient reductions.  The per-workgroup partials of LayerNorm's dgamma / dbeta (and the other
# fixed-order column sums of a block's backward) are not needed before the optimizer, so instead of two small launches
# per layer the backward pass queues them and ONE batched call (`neosr_colsum_many`) at its end finishes all of them
# and hands the results to `.grad` itself — the Function returns None for those inputs, so autograd never reads an
# unfinished sum.  Only leaf tensors are deferred (a non-leaf weight needs its gradient inside the graph).
_DEFERRED: list = []
_DEFERRED_BYTES = 0
# OPT-IN (ADVICE r2 / VERDICT r3): a deferred gradient bypasses autograd (the Function returns None and `.grad` is
# written by the flush), so `torch.autograd.grad(loss, params)` in user code would see None.  The queue is therefore
# only used inside `deferred_reductions()` — which the models put around their own `backward()` calls
# (models/image.py) — or process-wide with NEOSR_AMD_DEFER_REDUCE=1; NEOSR_AMD_DEFER_REDUCE=0 switches it off everywhere.
_DEFER_ENV = os.environ.get("NEOSR_AMD_DEFER_REDUCE")
DEFER_REDUCTIONS = _DEFER_ENV == "1"
_DEFER_SCOPE = 0
# A queued job pins the WHOLE workspace its partials live in (a view keeps the storage alive: ~100 MB per HAB of hat_l at
# B = 4, the split-K slabs of every Linear) until it is flushed, so the queue is bounded: it is flushed from inside the
# backward pass once it holds DEFER_MAX_JOBS jobs (the launch batch of `neosr_colsum_many`) or DEFER_MAX_BYTES of pinned
# workspaces, whichever comes first — peak memory no longer grows with the depth of the network.
DEFER_MAX_JOBS = 32
DEFER_MAX_BYTES = 1 << 30
# Set by the model on data-parallel runs (GradSync.grads_ready): the flush writes `.grad` outside autograd, so no
# post-accumulate hook fires for these parameters — the gradient exchange is told here that they are final.
GRADS_READY = None


@contextlib.contextmanager
def deferred_reductions(on: bool = True):
    """Scope in which parameter-gradient reductions may be batched at the end of `backward()` (see above)."""
    global _DEFER_SCOPE
    if not on or _DEFER_ENV == "0":
        yield
        return
    _DEFER_SCOPE += 1
    try:
        yield
    finally:
        _DEFER_SCOPE -= 1


def _task_id() -> int:
    fn = getattr(torch._C, "_current_graph_task_id", None)  # noqa: SLF001
    return fn() if fn is not None else -1


def _can_defer(*params) -> bool:
    # (not under hipGraph capture: the captured backward must contain its reductions)
    return ((DEFER_REDUCTIONS or _DEFER_SCOPE > 0) and all(p is not None and p.is_leaf and p.requires_grad for p in params)
            and not torch.cuda.is_current_stream_capturing())


# leaves with a queued reduction (id -> jobs outstanding): their post-accumulate hook — which also fires for a contribution
# that reached `.grad` through autograd — must not report them final to the gradient exchange before the flush has added
# the deferred part (GradSync._on_grad asks `has_pending`)
_PENDING_LEAVES: dict[int, int] = {}


def has_pending(leaf) -> bool:
    return _PENDING_LEAVES.get(id(leaf), 0) > 0


def _defer_colsum(part, rows: int, cols: int, ld: int, out, targets) -> None:
    """queue out[c] = sum_r part[r, c]; afterwards `targets` = [(leaf, view of out), ...] receive their gradients"""
    global _DEFERRED_BYTES
    task = _task_id()
    if _DEFERRED and _DEFERRED[0][6] != task:
        # jobs of ANOTHER backward pass: it died with an exception before its final callback ran.  Their partials must
        # not leak into this pass's gradients
        reset_deferred()
    # one callback per producing backward call, the first one to run does all the work (a pass that dies never runs its
    # callbacks, so "register only when the queue is empty" could strand jobs)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred)  # noqa: SLF001
    # (each pinned workspace counts once, however many jobs have their partials in it: ADVICE r3)
    sid = part.untyped_storage().data_ptr()
    if all(j[0].untyped_storage().data_ptr() != sid for j in _DEFERRED):
        _DEFERRED_BYTES += part.untyped_storage().nbytes()
    for leaf, _g in targets:
        _PENDING_LEAVES[id(leaf)] = _PENDING_LEAVES.get(id(leaf), 0) + 1
    _DEFERRED.append((part, rows, cols, ld, out, targets, task, torch.cuda.current_stream()))
    if len(_DEFERRED) >= DEFER_MAX_JOBS or _DEFERRED_BYTES >= DEFER_MAX_BYTES:
        _flush_deferred()


def reset_deferred() -> None:
    """Drop reductions queued by a backward pass that did not finish (models call this before each step)."""
    global _DEFERRED_BYTES
    _DEFERRED.clear()
    _DEFERRED_BYTES = 0
    _PENDING_LEAVES.clear()


def _flush_deferred() -> None:
    global _DEFERRED_BYTES
    jobs = list(_DEFERRED)
    _DEFERRED.clear()
    _DEFERRED_BYTES = 0
    if not jobs:
        return
    lib = _C.load()
    cur = torch.cuda.current_stream()
    for s in {j[7] for j in jobs}:  # partials produced on another stream (a backward node's forward stream): order after it
        if s != cur:
            cur.wait_stream(s)
    items = (_C.ColsumItem * len(jobs))()
    for it, (part, rows, cols, ld, out, *_r) in zip(items, jobs):
        it.x, it.out, it.rows, it.cols, it.ld, it.accumulate = part.data_ptr(), out.data_ptr(), rows, cols, ld, 0
    ws = torch.empty(lib.neosr_colsum_many_workspace_floats(items, len(jobs)), device=jobs[0][0].device,
                     dtype=torch.float32)
    _C.check(lib.neosr_colsum_many(items, len(jobs), ws.data_ptr(), _st()), "neosr_colsum_many")
    with torch.no_grad():
        for job in jobs:
            for leaf, g in job[5]:
                n = _PENDING_LEAVES.get(id(leaf), 0) - 1
                if n > 0:
                    _PENDING_LEAVES[id(leaf)] = n
                else:
                    _PENDING_LEAVES.pop(id(leaf), None)
                if leaf.grad is None:
                    leaf.grad = g
                else:
                    leaf.grad += g
    if GRADS_READY is not None:
        GRADS_READY([leaf for job in jobs for leaf, _g in job[5]])


def _as2d(x):
    x = _C.require_device(x, "x")
    if not x.is_contiguous():
        x = x.contiguous()
    return x.view(-1, x.shape[-1])


class Linear(torch.autograd.Function):
    """y = (x W^T + b) * row_scale[sample] + res  — nn.Linear with the DropPath + shortcut of
    swinir_arch.py:387 fused into the GEMM epilogue.  W is (N, K) as in nn.Linear."""

    @staticmethod
    def forward(ctx, x, w, b, res, rs, rows_per_scale):
        x2 = _as2d(x)
        ctx.leaves = (w, b)
        w = _C.require_device(w, "weight").contiguous()
        M, K = x2.shape
        N = w.shape[0]
        r2 = None if res is None else _as2d(res)
        y = gemm(_C.GEMM_NT, x2, w, M, N, K, bias=b, res=r2, row_scale=rs, rows_per_scale=rows_per_scale)
        ctx.save_for_backward(x2, w, rs)
        ctx.meta = (M, N, K, rows_per_scale, b is not None, res is not None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w, rs = ctx.saved_tensors
        M, N, K, rps, has_b, has_res = ctx.meta
        g2 = _as2d(g)
        # the DropPath scale of the incoming gradient is applied inside the GEMMs (epilogue of the data gradient,
        # operand rows of the weight gradient): no scaled copy of dY
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(_C.GEMM_NN, g2, w, M, K, N, row_scale=rs, rows_per_scale=rps).view(*g.shape[:-1], K)
        want_b = has_b and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            wl, bl = ctx.leaves
            gw, gb = _wgrad_pair(g2, N, K, want_b, wl, bl)  # bias gradient rides in the weight-gradient GEMM
            if _can_defer(wl, *((bl,) if want_b else ())):
                gemm(_C.GEMM_TN, g2, x2, N, K, M, out=gw, colsum_a=gb, row_scale=rs, rows_per_scale=rps,
                     defer_to=[(wl, gw)] + ([(bl, gb)] if want_b else []))
                gw = gb = None
            else:
                gemm(_C.GEMM_TN, g2, x2, N, K, M, out=gw, colsum_a=gb, row_scale=rs, rows_per_scale=rps)
        elif want_b:
            gb = colsum(g2 if rs is None else row_scale(g2, rs, rps))
        return gx, gw, gb, (g if has_res else None), None, None


def linear(x, w, b=None, res=None, rs=None, rows_per_scale=0):
    return Linear.apply(x, w, b, res, rs, rows_per_scale)


class Mlp(torch.autograd.Function):
    """y = (GELU(x W1^T + b1) W2^T + b2) * row_scale + res   (swinir_arch.py:15-38, :388).
    Bias, exact-erf GELU, DropPath scale and the residual live in the two GEMM epilogues; backward
    multiplies by GELU' in the epilogue of the fc2 data-gradient GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, rs, rows_per_scale):
        x2 = _as2d(x)
        ctx.leaves = (w1, b1, w2, b2)
        w1 = _C.require_device(w1, "fc1.weight").contiguous()
        w2 = _C.require_device(w2, "fc2.weight").contiguous()
        M, K = x2.shape
        Hd, N = w1.shape[0], w2.shape[0]
        keep = any(ctx.needs_input_grad)
        pre = _new((M, Hd), x2) if keep else None
        h = gemm(_C.GEMM_NT, x2, w1, M, Hd, K, bias=b1, aux_out=pre, gelu=True)
        r2 = None if res is None else _as2d(res)
        y = gemm(_C.GEMM_NT, h, w2, M, N, Hd, bias=b2, res=r2, row_scale=rs, rows_per_scale=rows_per_scale)
        if keep:
            ctx.save_for_backward(x2, w1, w2, pre, h, rs)
        ctx.meta = (M, K, Hd, N, rows_per_scale, res is not None)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, w1, w2, pre, h, rs = ctx.saved_tensors
        M, K, Hd, N, rps, has_res = ctx.meta
        g2 = _as2d(g)
        l1, lb1, l2, lb2 = ctx.leaves
        gw2, gb2 = _wgrad_pair(g2, N, Hd, True, l2, lb2)  # bias gradients ride in the weight-gradient GEMMs
        gw1, gb1 = _wgrad_pair(g2, Hd, K, True, l1, lb1)
        # (split reductions of both weight gradients queued for the batched pass at the end of backward when all four
        # parameters are leaves)
        defer = _can_defer(l1, lb1, l2, lb2) and all(ctx.needs_input_grad[1:5])
        # DropPath scale of g: operand rows of the fc2 weight gradient, epilogue of its data gradient
        gemm(_C.GEMM_TN, g2, h, N, Hd, M, out=gw2, colsum_a=gb2, row_scale=rs, rows_per_scale=rps,
             defer_to=[(l2, gw2), (lb2, gb2)] if defer else None)
        gpre = gemm(_C.GEMM_NN, g2, w2, M, Hd, N, aux_in=pre, row_scale=rs, rows_per_scale=rps)  # (g W2) GELU'(pre)
        gemm(_C.GEMM_TN, gpre, x2, Hd, K, M, out=gw1, colsum_a=gb1, defer_to=[(l1, gw1), (lb1, gb1)] if defer else None)
        if defer:
            gw1 = gb1 = gw2 = gb2 = None
        gx = gemm(_C.GEMM_NN, gpre, w1, M, K, Hd).view(*g.shape[:-1], K) if ctx.needs_input_grad[0] else None
        return gx, gw1, gb1, gw2, gb2, (g if has_res else None), None, None


def mlp(x, w1, b1, w2, b2, res=None, rs=None, rows_per_scale=0):
    return Mlp.apply(x, w1, b1, w2, b2, res, rs, rows_per_scale)


class LayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the last dim (swinir_arch.py:284,297,960)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _C.load()
        x2 = _as2d(x)
        rows, C_ = x2.shape
        y = torch.empty_like(x2)
        stats = _new((rows, 2), x2)
        _C.check(lib.neosr_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                         stats.data_ptr(), rows, C_, eps, _st()), "neosr_layernorm_fwd")
        ctx.save_for_backward(x2, gamma, stats)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        x2, gamma, stats = ctx.saved_tensors
        rows, C_ = x2.shape
        g2 = _as2d(g)
        dx = torch.empty_like(x2)
        dgb = _new((2 * C_,), x2)  # dgamma | dbeta adjacent: one reduction launch
        dg, db = dgb[:C_], dgb[C_:]
        ws = _new(((2 * 1024 + 512) * C_,), x2)
        _C.check(lib.neosr_layernorm_bwd(g2.data_ptr(), x2.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                         dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), rows, C_, 0,
                                         _st()), "neosr_layernorm_bwd")
        return dx.view(g.shape), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNorm.apply(x, gamma, beta, eps)


class ResidualLayerNorm(torch.autograd.Function):
    """(x, LayerNorm(x)) for a pre-norm residual block x -> x + f(norm(x)) (swinir_arch.py:343-392): the first output is
    the shortcut.  Autograd then hands BOTH gradients — the one that came over the shortcut and the one through the
    norm — to this backward, which sums them inside the LayerNorm backward kernel (`neosr_layernorm_bwd_res`) instead
    of with an elementwise add per norm."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _C.load()
        x2 = _as2d(x)
        rows, C_ = x2.shape
        y = torch.empty_like(x2)
        stats = _new((rows, 2), x2)
        _C.check(lib.neosr_layernorm_fwd(x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                         stats.data_ptr(), rows, C_, eps, _st()), "neosr_layernorm_fwd")
        ctx.save_for_backward(x2, gamma, stats)
        ctx.gamma_leaf, ctx.beta_leaf = gamma, beta
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, gs, g):
        lib = _C.load()
        x2, gamma, stats = ctx.saved_tensors
        rows, C_ = x2.shape
        if g is None:  # the normed branch was not used
            return gs, None, None, None
        g2 = _as2d(g)
        gs2 = None if gs is None else _as2d(gs)
        dx = torch.empty_like(x2)
        dgb = _pair_slots(ctx.gamma_leaf, ctx.beta_leaf, C_, C_)
        if dgb is None:
            dgb = _new((2 * C_,), x2)  # dgamma | dbeta adjacent: one reduction launch
        dg, db = dgb[:C_], dgb[C_:]
        ws = _new(((2 * 1024 + 512) * C_,), x2)
        if _can_defer(ctx.gamma_leaf, ctx.beta_leaf):
            rc = lib.neosr_layernorm_bwd_res(g2.data_ptr(), x2.data_ptr(), stats.data_ptr(), gamma.data_ptr(), _p(gs2),
                                             dx.data_ptr(), None, None, ws.data_ptr(), rows, C_, 0, _st())
            if rc >= 0:
                _C.check(rc or 1, "neosr_layernorm_bwd_res")
            _defer_colsum(ws, -rc, 2 * C_, 2 * C_, dgb, [(ctx.gamma_leaf, dg), (ctx.beta_leaf, db)])
            return dx.view(g.shape), None, None, None
        _C.check(lib.neosr_layernorm_bwd_res(g2.data_ptr(), x2.data_ptr(), stats.data_ptr(), gamma.data_ptr(),
                                             _p(gs2), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                             rows, C_, 0, _st()), "neosr_layernorm_bwd_res")
        return dx.view(g.shape), dg, db, None


def residual_layer_norm(x, gamma, beta, eps=1e-5):
    """-> (shortcut, norm(x)); use the shortcut (not x) as the residual operand of the block."""
    return ResidualLayerNorm.apply(x, gamma, beta, eps)


class WindowAttention(torch.autograd.Function):
    """softmax(q k^T * scale + rpb + shift-mask) v per (window, head) on the fused qkv matrix
    (B, H, W, 3C) in image order (swinir_arch.py:150-212 inside :343-385)."""

    @staticmethod
    def forward(ctx, qkv, table, heads, ws, shift, scale):
        lib = _C.load()
        ctx.table_leaf = table
        qkv = _C.require_device(qkv, "qkv").contiguous()
        table = _C.require_device(table, "relative_position_bias_table").contiguous()
        B, H, W, C3 = qkv.shape
        C_ = C3 // 3
        nW = (H // ws) * (W // ws) if ws else 0
        out = _new((B, H, W, C_), qkv)
        lse = _new((max(B * nW * heads * ws * ws, 1),), qkv)
        d = _C.WattnDesc(qkv=qkv.data_ptr(), rpb_table=table.data_ptr(), out=out.data_ptr(), lse=lse.data_ptr(),
                         B=B, H=H, W=W, C=C_, heads=heads, ws=ws, shift=shift, accumulate_rpb=0, scale=scale)
        _C.check(lib.neosr_window_attention_fwd(d, _st()), "neosr_window_attention_fwd")
        ctx.save_for_backward(qkv, table, lse, out)  # out: delta = rowsum(dO . O) in the backward kernel
        ctx.meta = (B, H, W, C_, heads, ws, shift, scale, nW)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        qkv, table, lse, out = ctx.saved_tensors
        B, H, W, C_, heads, ws, shift, scale, nW = ctx.meta
        g = g.contiguous()
        dqkv = torch.empty_like(qkv)
        dtab = torch.empty_like(table)
        wsp = _new(((B * nW + 256) * heads * (2 * ws - 1) ** 2,), qkv)
        defer = _can_defer(ctx.table_leaf)
        d = _C.WattnDesc(qkv=qkv.data_ptr(), rpb_table=table.data_ptr(), out=out.data_ptr(), lse=lse.data_ptr(),
                         dout=g.data_ptr(), dqkv=dqkv.data_ptr(), d_rpb_table=dtab.data_ptr(),
                         workspace=wsp.data_ptr(), B=B, H=H, W=W, C=C_, heads=heads, ws=ws, shift=shift,
                         accumulate_rpb=2 if defer else 0, scale=scale)
        rc = lib.neosr_window_attention_bwd(d, _st())
        if defer and rc < 0:  # bias-table partials [rows][bins * heads] at the start of the workspace
            cols = heads * (2 * ws - 1) ** 2
            _defer_colsum(wsp, -rc, cols, cols, dtab, [(ctx.table_leaf, dtab)])
            return dqkv, None, None, None, None, None
        _C.check(rc, "neosr_window_attention_bwd")
        return dqkv, dtab, None, None, None, None


def window_attention(qkv, table, heads, ws, shift, scale):
    return WindowAttention.apply(qkv, table, heads, ws, shift, scale)


# ------------------------------------------------------------------------------------------------
# Whole transformer blocks on the C++ plans of csrc/blocks.hip: TWO library calls per block and step instead of one
# autograd.Function + ctypes call per fused op.  NEOSR_AMD_BLOCK_PLANS=0 keeps the op-by-op composition (same kernels,
# same descriptors: bit-identical; tests/test_hip_blocks.py).
BLOCK_PLANS = os.environ.get("NEOSR_AMD_BLOCK_PLANS", "1") != "0"
# A plan keeps every intermediate of its block in ONE `save` buffer (about M * (7 C + 2 hidden) floats) — what backward
# needs.  Without autograd nothing is kept by the op-by-op composition, which frees as it goes: above this many tokens an
# inference pass (validation of whole images, untiled `test()`) takes that path instead (same kernels, same
# bits).  65 536 tokens = a 256 x 256 LR image; swinir_medium's block then saves ~0.5 GB.
PLAN_NOGRAD_MAX_TOKENS = int(os.environ.get("NEOSR_AMD_PLAN_NOGRAD_MAX_TOKENS", str(1 << 16)))


def use_block_plan(x) -> bool:
    """whether a transformer block should run as its C++ plan for this input (B, H, W, C)"""
    if not BLOCK_PLANS:
        return False
    return torch.is_grad_enabled() or x.shape[0] * x.shape[1] * x.shape[2] <= PLAN_NOGRAD_MAX_TOKENS


def _block_grad_buffer(params, like, meta=None):
    """One flat buffer for the gradients of a block's parameters, laid out like their slice of the network's parameter
    arena (so every (weight, bias) / (gamma, beta) pair is contiguous) — the slice of the data-parallel exchange arena
    itself when GradSync hands out slots (GRAD_SLOT) — and its per-parameter views (one `as_strided` each; the layout is
    cached on `meta`: it depends on the shapes only)."""
    lay = meta.get("_layout") if meta is not None else None
    if lay is None:
        from neosr_amd.hip.nets import arena_layout

        offs, total = arena_layout(params)
        lay = (offs, total, [tuple(p.shape) for p in params], [p.stride() for p in params])
        if meta is not None:
            meta["_layout"] = lay
    offs, total, shapes, strides = lay
    flat = None
    if GRAD_SLOT is not None:
        slots = [GRAD_SLOT(p) for p in params]
        s0 = slots[0]
        if all(s is not None for s in slots) and all(s.data_ptr() == s0.data_ptr() + 4 * o for s, o in zip(slots, offs)):
            flat = torch.empty(0, device=s0.device, dtype=torch.float32)
            flat.set_(s0.untyped_storage(), s0.storage_offset(), (total,), (1,))
    if flat is None:
        flat = _new((total,), like)
    base = flat.storage_offset()
    return flat, [flat.as_strided(sh, st, base + o) for sh, st, o in zip(shapes, strides, offs)]


def _no_plan_meta():
    return None


class PlanMeta(dict):
    """A block module's plan cache (`module._plan_meta`): static descriptor fields plus the cached ctypes descriptor / gradient
    structs (`_desc`, `_G`, `_layout`).  A cache, not state: ctypes structs with pointer fields can be neither deep-copied
    nor pickled, and their addresses belong to the original's parameters — a copied / pickled module gets None and
    rebuilds its own on the next forward (`DirectGrads` in hip/nets.py does the same)."""

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_plan_meta, ())


class TBlock(torch.autograd.Function):
    """SwinTransformerBlock (swinir_arch.py:231-392) / OCAB (hat_arch.py:393-515) / HAB (hat_arch.py:218-350) through
    `neosr_tblock_forward` / `neosr_tblock_backward`.  `params`: the block's parameters in `named_parameters()` order
    (= their order in the network's flat arena), `meta["names"]` the descriptor field of each (`_C.TBLOCK_PARAMS`,
    `_C.TBLOCK_CAB_PARAMS`); `meta["images"]()` -> the packed weight images of the CAB convolutions (HAB only)."""

    @staticmethod
    def _desc(meta, x, params):
        """the block's descriptor, cached on `meta` per (geometry, parameter addresses): ~40 ctypes field writes and 13-21
        device / contiguity checks per call otherwise"""
        B, H, W, C_ = x.shape
        key = (B, H, W, C_) + tuple(p.data_ptr() for p in params)
        hit = meta.get("_desc")
        if hit is not None and hit[0] == key:
            return hit[1]
        names = meta["names"]
        if len(params) != len(names):
            raise _C.NeosrAmdError(f"TBlock: expected {len(names)} parameter tensors, got {len(params)}")
        d = _C.TBlockDesc(B=B, H=H, W=W, C=C_, **meta["ints"], **meta["floats"])
        for n, p in zip(names, params):
            _C.require_device(p, n)
            if not p.is_contiguous():
                raise _C.NeosrAmdError(f"TBlock: parameter {n} must be contiguous")
            setattr(d, n, p.data_ptr())
        nsave = _C.load().neosr_tblock_save_floats(d)
        nws = _C.load().neosr_tblock_bwd_workspace_floats(d)
        if nsave < 0 or nws < 0:
            _C.check(1, "neosr_tblock_save_floats")
        meta["_desc"] = (key, (d, nsave, nws))
        return d, nsave, nws

    @staticmethod
    def forward(ctx, x, rs, rs2, meta, *params):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        d, nsave, nws = TBlock._desc(meta, x, params)
        imgs = meta["images"](*x.shape[:3]) if "images" in meta else None
        if imgs:
            for n, t in imgs.items():
                setattr(d, n, _p(t))
        d.drop_scale, d.drop_scale2 = _p(rs), _p(rs2)
        save = _new((nsave,), x)
        out = torch.empty_like(x)
        _C.check(lib.neosr_tblock_forward(d, x.data_ptr(), out.data_ptr(), save.data_ptr(), _st()), "neosr_tblock_forward")
        if any(ctx.needs_input_grad):
            # (the parameters are leaves the module keeps alive; like the whole-network plans, backward reads them as they
            # are then — nothing may write them between forward and backward, which autograd would only have detected)
            ctx.save_for_backward(x, rs, rs2, save)
            ctx.meta, ctx.leaves, ctx.imgs = meta, params, imgs
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        x, rs, rs2, save = ctx.saved_tensors
        meta, params = ctx.meta, ctx.leaves
        g = g.contiguous()
        d, _nsave, nws = TBlock._desc(meta, x, params)
        if ctx.imgs:
            for n, t in ctx.imgs.items():
                setattr(d, n, _p(t))
        d.drop_scale, d.drop_scale2 = _p(rs), _p(rs2)
        flat, views = _block_grad_buffer(params, x, meta)
        hit = meta.get("_G")   # (the allocator hands a step's buffers out at the same addresses: the struct is reused)
        if hit is not None and hit[0] == flat.data_ptr():
            G = hit[1]
        else:
            G = _C.TBlockGrads()
            base = flat.data_ptr()
            for n, o in zip(meta["names"], meta["_layout"][0]):
                setattr(G, n, base + 4 * o)
            meta["_G"] = (base, G)
        ws = _new((nws,), x)
        dx = torch.empty_like(x)
        tails = lib.neosr_tblock_tails()
        _C.check(lib.neosr_tblock_backward(d, x.data_ptr(), g.data_ptr(), save.data_ptr(), dx.data_ptr(), G,
                                           ws.data_ptr(), _st()), "neosr_tblock_backward")
        if lib.neosr_tblock_tails() != tails:
            _tail_issued(tails, ws, save, g, params)
        grads = [v if need else None for v, need in zip(views, ctx.needs_input_grad[4:])]
        return (dx if ctx.needs_input_grad[0] else None), None, None, None, *grads


# ---- the backward plans' weight-gradient TAILS (csrc/blocks.hip, include/neosr_amd.h: neosr_tblock_tail_join).  A call of
# neosr_tblock_backward may return with its tail in flight on the library's tail stream: the buffers it uses are kept alive
# here until a host-side query says the tail has finished (neosr_tblock_tail_done: nothing waits on the compute stream),
# the caller's stream is joined with the tails at the end of the backward pass, and at once where a gradient would be
# accumulated into an existing `.grad` (autograd would add to it on the caller's stream before the tail has written it).
_TAIL_KEEP: list = []
_TAIL_CALLBACK = False
_TAIL_STREAM = None


def join_tails() -> int:
    """the current stream waits for every tail issued so far; drops the kept buffers (they are only reused behind the wait)"""
    n = int(_C.load().neosr_tblock_tail_join(_st()))
    if n < 0:
        _C.check(1, "neosr_tblock_tail_join")
    _TAIL_KEEP.clear()
    return n


def _tails_end_of_backward() -> None:
    # (an engine callback may run with another current stream than the backward nodes had: join on THEIR stream)
    global _TAIL_CALLBACK, _TAIL_STREAM
    _TAIL_CALLBACK = False
    s, _TAIL_STREAM = _TAIL_STREAM, None
    cur = torch.cuda.current_stream()
    if s is None or s == cur:
        join_tails()
        return
    with torch.cuda.stream(s):
        join_tails()
    cur.wait_stream(s)


def _tail_issued(index, ws, save, g, params) -> None:
    global _TAIL_CALLBACK, _TAIL_STREAM
    _TAIL_STREAM = torch.cuda.current_stream()
    lib = _C.load()
    while _TAIL_KEEP and lib.neosr_tblock_tail_done(_TAIL_KEEP[0][0]):
        del _TAIL_KEEP[0]
    _TAIL_KEEP.append((index, ws, save, g))
    if len(_TAIL_KEEP) > 6:   # (the library answers for its newest seven tails only)
        join_tails()
    if not _TAIL_CALLBACK:
        _TAIL_CALLBACK = True
        torch.autograd.Variable._execution_engine.queue_callback(_tails_end_of_backward)  # noqa: SLF001
    if any(p.grad is not None for p in params):   # (accumulation: AccumulateGrad adds on this stream right after we return)
        join_tails()


def tblock(x, rs, rs2, meta, params):
    return TBlock.apply(x, rs, rs2, meta, *params)


class PixelShuffleNHWC(torch.autograd.Function):
    """nn.PixelShuffle(r) on channels-last tensors: (B,H,W,C*r*r) -> (B,H*r,W*r,C). Bit-exact."""

    @staticmethod
    def forward(ctx, x, r):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, H, W, Crr = x.shape
        C_ = Crr // (r * r)
        out = _new((B, H * r, W * r, C_), x)
        _C.check(lib.neosr_pixel_shuffle_nhwc(x.data_ptr(), out.data_ptr(), B, H, W, C_, r, 0, _st()),
                 "neosr_pixel_shuffle_nhwc")
        ctx.meta = (B, H, W, C_, r)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        B, H, W, C_, r = ctx.meta
        g = g.contiguous()
        out = _new((B, H, W, C_ * r * r), g)
        _C.check(lib.neosr_pixel_shuffle_nhwc(g.data_ptr(), out.data_ptr(), B, H, W, C_, r, 1, _st()),
                 "neosr_pixel_shuffle_nhwc")
        return out, None


class Affine(torch.autograd.Function):
    """(x + shift) * scale."""

    @staticmethod
    def forward(ctx, x, shift, scale):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_affine(x.data_ptr(), out.data_ptr(), x.numel(), shift, scale, _st()), "neosr_affine")
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        g = g.contiguous()
        out = torch.empty_like(g)
        _C.check(lib.neosr_affine(g.data_ptr(), out.data_ptr(), g.numel(), 0.0, ctx.scale, _st()), "neosr_affine")
        return out, None, None


class FlashWindowAttention(torch.autograd.Function):
    """Window attention of HAT on the fused qkv matrix (B, H, W, 3C), windows of `ws` = 16 (hat_s / m / l) or 8:
    `ks = ws` = (shifted-)window self-attention of HAB, `ks = 1.5 ws` = overlapping cross-attention of OCAB
    (hat_arch.py:168-216, 445-516)."""

    @staticmethod
    def forward(ctx, qkv, table, heads, ks, shift, scale, ws):
        lib = _C.load()
        ctx.table_leaf = table
        qkv = _C.require_device(qkv, "qkv").contiguous()
        table = _C.require_device(table, "relative_position_bias_table").contiguous()
        B, H, W, C3 = qkv.shape
        C_ = C3 // 3
        out = _new((B, H, W, C_), qkv)
        lse = _new((B * (H // ws) * (W // ws) * heads * ws * ws,), qkv)
        d = _C.FattnDesc(qkv=qkv.data_ptr(), rpb_table=table.data_ptr(), out=out.data_ptr(), lse=lse.data_ptr(),
                         B=B, H=H, W=W, C=C_, heads=heads, ws=ws, ks=ks, shift=shift, accumulate_rpb=0, scale=scale)
        _C.check(lib.neosr_flash_window_attention_fwd(d, _st()), "neosr_flash_window_attention_fwd")
        ctx.save_for_backward(qkv, table, out, lse)
        ctx.meta = (B, H, W, C_, heads, ks, shift, scale, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        qkv, table, out, lse = ctx.saved_tensors
        B, H, W, C_, heads, ks, shift, scale, ws = ctx.meta
        g = g.contiguous()
        dqkv = torch.empty_like(qkv)
        dtab = torch.empty_like(table)
        d = _C.FattnDesc(qkv=qkv.data_ptr(), rpb_table=table.data_ptr(), out=out.data_ptr(), lse=lse.data_ptr(),
                         dout=g.data_ptr(), dqkv=dqkv.data_ptr(), d_rpb_table=dtab.data_ptr(), workspace=None,
                         B=B, H=H, W=W, C=C_, heads=heads, ws=ws, ks=ks, shift=shift, accumulate_rpb=0, scale=scale)
        wsp = torch.empty(lib.neosr_flash_window_attention_workspace_bytes(d) // 4, device=qkv.device,
                          dtype=torch.float32)
        d.workspace = wsp.data_ptr()
        defer = ks == ws and _can_defer(ctx.table_leaf)  # (the overlapping form gathers its bins from a dense sum)
        if defer:
            d.accumulate_rpb = 2
        rc = lib.neosr_flash_window_attention_bwd(d, _st())
        if defer and rc < 0:
            cols = heads * (2 * ws - 1) ** 2
            part = wsp[B * (H // ws) * (W // ws) * heads * ws * ws:]
            _defer_colsum(part, -rc, cols, cols, dtab, [(ctx.table_leaf, dtab)])
            return dqkv, None, None, None, None, None, None
        _C.check(rc, "neosr_flash_window_attention_bwd")
        return dqkv, dtab, None, None, None, None, None


def flash_window_attention(qkv, table, heads, ks, shift, scale, ws=16):
    return FlashWindowAttention.apply(qkv, table, heads, ks, shift, scale, ws)


class Gelu(torch.autograd.Function):
    """erf-form nn.GELU between the two convs of CAB (hat_arch.py:46); erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7)."""

    @staticmethod
    def forward(ctx, x):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_gelu(x.data_ptr(), None, out.data_ptr(), x.numel(), _st()), "neosr_gelu")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_gelu(x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel(), _st()), "neosr_gelu")
        return out


class ChannelGate(torch.autograd.Function):
    """out = res + alpha * y * sigmoid(W2 relu(W1 mean_hw(y) + b1) + b2): ChannelAttention
    (hat_arch.py:15-37) fused with HAB's `+ conv_x * conv_scale` (:347).  y, res: (B, H, W, C)."""

    trace = None   # tests: a list collects (pooled, w1, b1) of every forward (the bottleneck's ReLU inputs follow from them)

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2, res, alpha):
        lib = _C.load()
        y = _C.require_device(y, "y").contiguous()
        B, H, W, C_ = y.shape
        rows, Cs = H * W, w1.shape[0]
        w1c, w2c = w1.contiguous(), w2.contiguous()
        pooled, attn, hidden = _new((B, C_), y), _new((B, C_), y), _new((B, Cs), y)
        ws = _new((B * 128 * C_,), y)
        _C.check(lib.neosr_batched_colsum(y.data_ptr(), None, pooled.data_ptr(), ws.data_ptr(), B, rows, C_,
                                          1.0 / rows, _st()), "neosr_batched_colsum")
        _C.check(lib.neosr_channel_attention_fwd(pooled.data_ptr(), w1c.data_ptr(), b1.data_ptr(), w2c.data_ptr(),
                                                 b2.data_ptr(), hidden.data_ptr(), attn.data_ptr(), B, C_, Cs, _st()),
                 "neosr_channel_attention_fwd")
        if ChannelGate.trace is not None:
            ChannelGate.trace.append((pooled, w1c, b1))
        out = torch.empty_like(y)
        r = None if res is None else res.contiguous()
        _C.check(lib.neosr_scale_channels_add(y.data_ptr(), attn.data_ptr(), _p(r), out.data_ptr(), B, rows, C_, alpha,
                                              _st()), "neosr_scale_channels_add")
        ctx.save_for_backward(y, w1c, w2c, pooled, attn, hidden)
        ctx.meta = (B, rows, C_, Cs, alpha, res is not None, w1.shape, w2.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        y, w1, w2, pooled, attn, hidden = ctx.saved_tensors
        B, rows, C_, Cs, alpha, has_res, s1, s2 = ctx.meta
        g = g.contiguous()
        dattn, dpooled = _new((B, C_), y), _new((B, C_), y)
        ws = _new((B * 128 * C_,), y)
        _C.check(lib.neosr_batched_colsum(g.data_ptr(), y.data_ptr(), dattn.data_ptr(), ws.data_ptr(), B, rows, C_,
                                          alpha, _st()), "neosr_batched_colsum")
        dw1, db1, dw2, db2 = _new((Cs, C_), y), _new((Cs,), y), _new((C_, Cs), y), _new((C_,), y)
        _C.check(lib.neosr_channel_attention_bwd(dattn.data_ptr(), attn.data_ptr(), hidden.data_ptr(), pooled.data_ptr(),
                                                 w1.data_ptr(), w2.data_ptr(), dpooled.data_ptr(), dw1.data_ptr(),
                                                 db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), B, C_, Cs, _st()),
                 "neosr_channel_attention_bwd")
        dy = torch.empty_like(y)
        _C.check(lib.neosr_scale_channels_bwd(g.data_ptr(), attn.data_ptr(), dpooled.data_ptr(), dy.data_ptr(), B, rows,
                                              C_, alpha, _st()), "neosr_scale_channels_bwd")
        return dy, dw1.view(s1), db1, dw2.view(s2), db2, (g if has_res else None), None
