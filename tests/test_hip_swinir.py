"""GPU parity tests of the SwinIR window-attention path through the C ABI: fp32 MFMA GEMM (all modes
and epilogues), LayerNorm, (shifted-)window attention, channels-last PixelShuffle, whole Swin blocks
and whole nets against fixtures produced by the reference, and the `image` model trajectory with
network_g = swinir_small.

Tolerance (BASELINE.json north_star): 1e-3 relative (per-tensor ||d||/||ref||) in fp32; PixelShuffle
is index math and must be bit-exact.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import group, load_golden, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.array(a))


GEMM_CASES = [(200, 180, 60), (128, 64, 32), (4096, 540, 180), (333, 360, 180), (70, 8, 8), (1000, 180, 360),
              (32768, 540, 180), (32768, 180, 360)]  # (the last two: 128-row NT tiles; the small ones take the 64-row kernel)


@pytest.mark.parametrize("M,N,K", GEMM_CASES)
def test_gemm_modes_vs_float64(M, N, K):
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Y = torch.randn(M, N, generator=g)
    nt = tr.gemm(_C.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K)
    assert rel_err(nt, A.double() @ W.double().t()) < 1e-5
    nn_ = tr.gemm(_C.GEMM_NN, Y.to(DEV), W.to(DEV), M, K, N)
    assert rel_err(nn_, Y.double() @ W.double()) < 1e-5
    tn = tr.gemm(_C.GEMM_TN, Y.to(DEV), A.to(DEV), N, K, M)
    assert rel_err(tn, Y.double().t() @ A.double()) < 1e-5
    cs = tr.colsum(Y.to(DEV))
    assert rel_err(cs, Y.double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(4096, 540, 180), (1000, 180, 360)])
def test_gemm_fast_matmul_tier_is_labelled_precision(M, N, K):
    """`fast_matmul` (reference train.py:168-173: TF32 / "medium" matmul precision) switches the bf16x3 Linear GEMMs to their
    three leading cross terms: the result is within 1e-4 of float64 (TF32 would be ~1e-3), measurably looser than the default
    (~1e-7: the switch is wired), and the default comes back bit for bit when it is switched off."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(3 * M + N + K)
    A, W, Y = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    refs = (A.double() @ W.double().t(), Y.double() @ W.double(), Y.double().t() @ A.double())

    def run():
        return (tr.gemm(_C.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K), tr.gemm(_C.GEMM_NN, Y.to(DEV), W.to(DEV), M, K, N),
                tr.gemm(_C.GEMM_TN, Y.to(DEV), A.to(DEV), N, K, M))

    base = run()
    prev = _C.set_fast_matmul(True)
    try:
        fast = run()
    finally:
        _C.set_fast_matmul(prev)
    again = run()
    for b, f, a2, r in zip(base, fast, again, refs):
        eb, ef = rel_err(b, r), rel_err(f, r)
        assert eb < 1e-6 and 3e-7 < ef < 1e-4, (eb, ef)
        assert torch.equal(b, a2)


@pytest.mark.parametrize("M,N,K", [(4096, 540, 180), (1000, 180, 360), (32768, 180, 180)])
def test_gemm_bf16x3_products_are_fp32_faithful(M, N, K):
    """The default Linear GEMMs (NT forward, NN backward-data) run their products on the bf16 MFMA from three bf16 pieces per
    fp32 operand (six cross products, fp32 accumulate: `neosr_set_gemm_x3`).  Against float64 they must be as accurate as the
    fp32-MFMA kernels they replace (error <= 1.5x theirs, both ~1e-7 relative), on well-scaled operands and on operands that
    span 12 orders of magnitude per row (a piece that underflowed or a dropped cross term would show there), and the two
    forms must really differ in their bits (the switch is wired)."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    lib = _C.load()
    g = torch.Generator().manual_seed(5 * M + N + K)
    for wide in (False, True):
        A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        Y = torch.randn(M, N, generator=g)
        if wide:
            A = A * torch.pow(10.0, torch.randint(-6, 7, (M, K), generator=g).float())
            W = W * torch.pow(10.0, torch.randint(-6, 7, (N, K), generator=g).float())
        ref_nt, ref_nn = A.double() @ W.double().t(), Y.double() @ W.double()
        out = {}
        prev = lib.neosr_set_gemm_x3(1)
        try:
            for x3 in (1, 0):
                lib.neosr_set_gemm_x3(x3)
                out[x3] = (tr.gemm(_C.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K), tr.gemm(_C.GEMM_NN, Y.to(DEV), W.to(DEV), M, K, N))
        finally:
            lib.neosr_set_gemm_x3(prev)
        for i, ref in enumerate((ref_nt, ref_nn)):
            e3, e1 = rel_err(out[1][i], ref), rel_err(out[0][i], ref)
            assert e3 < 1e-6 and e3 <= 1.5 * e1 + 2e-8, (wide, i, e3, e1)
            assert not torch.equal(out[1][i], out[0][i])


@pytest.mark.parametrize("Mo,No,T_", [(96, 64, 16), (180, 180, 1), (540, 180, 32768), (192, 8, 37), (12, 4, 5001),
                                     (360, 180, 8191), (180, 360, 4100), (128, 64, 777)])
def test_gemm_tn_weight_gradient_with_bias_sum(Mo, No, T_):
    """dW = dY^T X and db = sum_t dY through the TN GEMM (register-fed 96 x 64 wave tiles when the ragged last m tile
    splits into whole 3-column lane groups, else the staged 128 x 64 kernel): ragged / odd token counts, token runs
    shorter than one batch, every transformer shape, accumulate on top of an existing gradient, run-to-run bits."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(Mo + No + T_)
    dY, X = torch.randn(T_, Mo, generator=g), torch.randn(T_, No, generator=g)
    ref, refb = dY.double().t() @ X.double(), dY.double().sum(0)
    gw, gb = tr._wgrad_pair(dY.to(DEV), Mo, No, True)
    tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, out=gw, colsum_a=gb)
    assert rel_err(gw, ref) < 1e-5 and rel_err(gb, refb) < 1e-5
    first = (gw.clone(), gb.clone())
    tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, out=gw, colsum_a=gb)
    assert torch.equal(gw, first[0]) and torch.equal(gb, first[1])
    tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, out=gw, colsum_a=gb, accumulate=True)
    assert rel_err(gw, 2 * ref) < 1e-5 and rel_err(gb, 2 * refb) < 1e-5
    sep = torch.empty(Mo, device=DEV)  # bias sum kept away from dW: second reduction pass
    gw2 = tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, colsum_a=sep)
    assert torch.equal(gw2, first[0]) and rel_err(sep, refb) < 1e-5  # (two-stage column sum: other rounding order)


@pytest.mark.parametrize("Mo,No,T_,rps", [(180, 180, 4096 * 3, 4096), (540, 180, 2048 + 640, 256), (360, 180, 5000, 1000),
                                         (128, 64, 777, 100), (180, 360, 4100, 32), (96, 64, 64, 32)])
def test_gemm_tn_scales_gradient_rows_per_sample(Mo, No, T_, rps):
    """DropPath in backward (arch_util.py:118-133 under autograd): dW = (s[t // rps] dY[t])^T X and db = sum_t s dY[t]
    with the scale applied to the operand rows inside the TN GEMM — register-fed kernel when rps is a multiple of its
    32-token batch (scale groups ending inside a run, ragged last group, dropped samples = exact zeros), staged kernel
    otherwise — against a scaled copy through the same GEMM and float64; NN takes the same scale in its epilogue."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(Mo + No + T_ + rps)
    dY, X = torch.randn(T_, Mo, generator=g), torch.randn(T_, No, generator=g)
    ngrp = -(-T_ // rps)
    sc = torch.where(torch.rand(ngrp, generator=g) < 0.3, torch.zeros(ngrp), torch.full((ngrp,), 1 / 0.7))
    rows = sc.repeat_interleave(rps)[:T_, None]
    ref, refb = (dY * rows).double().t() @ X.double(), (dY * rows).double().sum(0)
    gw, gb = tr._wgrad_pair(dY.to(DEV), Mo, No, True)
    tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, out=gw, colsum_a=gb, row_scale=sc.to(DEV), rows_per_scale=rps)
    assert rel_err(gw, ref) < 1e-5 and rel_err(gb, refb) < 1e-5
    first = (gw.clone(), gb.clone())
    tr.gemm(_C.GEMM_TN, dY.to(DEV), X.to(DEV), Mo, No, T_, out=gw, colsum_a=gb, row_scale=sc.to(DEV), rows_per_scale=rps)
    assert torch.equal(gw, first[0]) and torch.equal(gb, first[1])
    # the scaled copy through the unscaled GEMM: same products, the scale (0 or 1/keep) commutes with fp32 rounding
    # only up to the last bit
    gw2, gb2 = tr._wgrad_pair(dY.to(DEV), Mo, No, True)
    tr.gemm(_C.GEMM_TN, (dY * rows).to(DEV), X.to(DEV), Mo, No, T_, out=gw2, colsum_a=gb2)
    assert rel_err(gw, gw2.double().cpu()) < 1e-6 and rel_err(gb, gb2.double().cpu()) < 1e-6
    W = torch.randn(Mo, No, generator=g)
    gx = tr.gemm(_C.GEMM_NN, dY.to(DEV), W.to(DEV), T_, No, Mo, row_scale=sc.to(DEV), rows_per_scale=rps)
    assert rel_err(gx, (dY * rows).double() @ W.double()) < 1e-5


def test_colsum_many_equals_single_calls_bitwise():
    """The batched reduction (one launch pair per 32 jobs) cuts every job into the row slabs `neosr_colsum` uses: same
    bits; 40 jobs of every size class (single slab, slabbed narrow matrices, wide split-K slabs, strided rows)."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    lib = _C.load()
    g = torch.Generator().manual_seed(11)
    shapes = [(1024, 360, 360), (256, 5766, 5766), (56, 97380, 97380), (3, 64, 64), (1, 4, 4), (700, 180, 360),
              (17, 1000, 1024), (43, 33, 40)] * 5
    xs = [torch.randn(r, ld, generator=g).to(DEV) for (r, c, ld) in shapes]
    outs = [torch.empty(c, device=DEV) for (r, c, ld) in shapes]
    items = (_C.ColsumItem * len(shapes))()
    for it, x, o, (r, c, ld) in zip(items, xs, outs, shapes):
        it.x, it.out, it.rows, it.cols, it.ld, it.accumulate = x.data_ptr(), o.data_ptr(), r, c, ld, 0
    ws = torch.empty(lib.neosr_colsum_many_workspace_floats(items, len(shapes)), device=DEV)
    _C.check(lib.neosr_colsum_many(items, len(shapes), ws.data_ptr(), None), "neosr_colsum_many")
    for x, o, (r, c, ld) in zip(xs, outs, shapes):
        ref = torch.empty(c, device=DEV)
        w1 = torch.empty(256 * c + 64, device=DEV)
        _C.check(lib.neosr_colsum(x.data_ptr(), ref.data_ptr(), w1.data_ptr(), r, c, ld, 0, None), "neosr_colsum")
        assert torch.equal(o, ref)
        assert rel_err(o, x[:, :c].double().sum(0)) < 1e-5


def test_deferred_parameter_gradient_reductions_equal_immediate_ones(monkeypatch):
    """LayerNorm dgamma / dbeta and the Linear / Mlp weight + bias gradients with their reductions queued to the end of
    backward (returned as None, handed to `.grad` by the flush) against the immediate path: identical bits, `.grad`
    accumulation over two backward passes included, non-leaf parameters untouched (they keep the immediate path)."""
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(5)
    M, Cc = 4096 + 96, 180
    x0 = torch.randn(2, M // 2, Cc, generator=g).to(DEV)
    P = {k: v.to(DEV) for k, v in dict(
        g1=torch.randn(Cc, generator=g), b1=torch.randn(Cc, generator=g), w=torch.randn(540, Cc, generator=g) * .05,
        wb=torch.randn(540, generator=g), f1=torch.randn(360, 540, generator=g) * .05, fb1=torch.randn(360, generator=g),
        f2=torch.randn(Cc, 360, generator=g) * .05, fb2=torch.randn(Cc, generator=g)).items()}
    rs = (torch.rand(2, generator=g) < 0.7).float().to(DEV) / 0.7

    def run(defer, scale_w):
        monkeypatch.setattr(tr, "DEFER_REDUCTIONS", defer)
        p = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        x = x0.clone().requires_grad_(True)
        for rep in range(2):  # second pass accumulates into existing .grad
            w = p["w"] * 2.0 if scale_w else p["w"]  # non-leaf weight: must stay on the immediate path
            sc, n = tr.residual_layer_norm(x, p["g1"], p["b1"])
            h = tr.linear(n, w, p["wb"])
            y = tr.mlp(h, p["f1"], p["fb1"], p["f2"], p["fb2"], res=sc, rs=rs, rows_per_scale=M // 2)
            (y * (1.0 + rep)).sum().backward()
        assert not tr._DEFERRED
        return [x.grad] + [p[k].grad for k in sorted(p)]

    for scale_w in (False, True):
        a, b = run(True, scale_w), run(False, scale_w)
        for u, v in zip(a, b):
            assert u is not None and torch.equal(u, v)


def test_deferred_reductions_are_opt_in():
    """Outside `deferred_reductions()` (the scope models/image.py puts around its own backward calls) nothing bypasses
    autograd: `torch.autograd.grad` returns every parameter gradient, equal bit for bit to what the scoped, deferred
    pass leaves in `.grad` (VERDICT r3 / ADVICE r2)."""
    from neosr_amd.hip import transformer as tr

    if tr._DEFER_ENV is not None:
        pytest.skip("NEOSR_AMD_DEFER_REDUCE set")
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 2048, 180, generator=g).to(DEV)
    P = [t.to(DEV).requires_grad_(True) for t in (torch.randn(180, generator=g), torch.randn(180, generator=g),
                                                  torch.randn(360, 180, generator=g) * .05, torch.randn(360, generator=g))]

    def loss():
        _sc, n = tr.residual_layer_norm(x, P[0], P[1])
        return tr.linear(n, P[2], P[3]).square().sum()

    grads = torch.autograd.grad(loss(), P)
    assert all(gr is not None for gr in grads) and not tr._DEFERRED
    with tr.deferred_reductions():
        loss().backward()
    assert not tr._DEFERRED
    for p, gr in zip(P, grads):
        assert torch.equal(p.grad, gr)


def test_gemm_random_shapes_all_modes_and_epilogues():
    """Seeded sweep over odd shapes: every GEMM kernel (128- and 64-row NT tiles with trimmed last chunk / skipped column
    tile / LDS bias / prefetched residual, staged NN, register-fed and staged TN) and every epilogue against float64."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    rng = np.random.default_rng(20260928)
    g = torch.Generator().manual_seed(4)
    for case in range(24):
        M = int(rng.choice([4, 60, 128, 516, 1000, 4100, 9000, 20000]))
        N = 4 * int(rng.integers(1, 150))
        K = 4 * int(rng.integers(1, 100))
        A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
        rps = max(1, M // int(rng.integers(1, 5)))
        rs = torch.rand(-(-M // rps), generator=g) + 0.5
        Ad, Wd = A.to(DEV), W.to(DEV)
        ref = A.double() @ W.double().t() + b.double()
        mode = case % 3
        if mode == 0:  # bias + DropPath row scale + residual
            y = tr.gemm(_C.GEMM_NT, Ad, Wd, M, N, K, bias=b.to(DEV), res=res.to(DEV), row_scale=rs.to(DEV), rows_per_scale=rps)
            want = ref * rs.double().repeat_interleave(rps)[:M, None] + res.double()
        elif mode == 1:  # bias + GELU, pre-activation kept
            aux = torch.empty(M, N, device=DEV)
            y = tr.gemm(_C.GEMM_NT, Ad, Wd, M, N, K, bias=b.to(DEV), gelu=True, aux_out=aux)
            want = torch.nn.functional.gelu(ref)
            assert rel_err(aux, ref) < 1e-5, (case, M, N, K)
        else:  # bare
            y = tr.gemm(_C.GEMM_NT, Ad, Wd, M, N, K)
            want = A.double() @ W.double().t()
        assert rel_err(y, want) < 2e-5, (case, "NT", M, N, K)
        G = torch.randn(M, N, generator=g)
        Gd = G.to(DEV)
        gx = tr.gemm(_C.GEMM_NN, Gd, Wd, M, K, N)
        assert rel_err(gx, G.double() @ W.double()) < 1e-5, (case, "NN", M, N, K)
        z = torch.randn(M, K, generator=g)
        gz = tr.gemm(_C.GEMM_NN, Gd, Wd, M, K, N, aux_in=z.to(DEV))  # x GELU'(z)
        zz = z.double().requires_grad_(True)
        torch.nn.functional.gelu(zz).sum().backward()
        assert rel_err(gz, (G.double() @ W.double()) * zz.grad) < 2e-5, (case, "NN gelu'", M, N, K)
        gw, gb = tr._wgrad_pair(Gd, N, K, True)
        tr.gemm(_C.GEMM_TN, Gd, Ad, N, K, M, out=gw, colsum_a=gb)
        assert rel_err(gw, G.double().t() @ A.double()) < 1e-5, (case, "TN", M, N, K)
        assert rel_err(gb, G.double().sum(0)) < 1e-5, (case, "TN bias", M, N, K)


def test_gemm_weight_at_unaligned_arena_offset():
    """weights living at a 4-byte-aligned (not 16-byte) offset of the packed parameter arena"""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(8)
    M, N, K = 300, 180, 60
    A = torch.randn(M, K, generator=g)
    arena = torch.randn(2 + N * K + N, generator=g).to(DEV)
    W, b = arena[2 : 2 + N * K].view(N, K), arena[2 + N * K :]
    assert W.data_ptr() % 16 != 0
    y = tr.gemm(_C.GEMM_NT, A.to(DEV), W, M, N, K, bias=b)
    assert rel_err(y, A.double() @ W.double().cpu().t() + b.double().cpu()) < 1e-5
    gy = torch.randn(M, N, generator=g)
    gx = tr.gemm(_C.GEMM_NN, gy.to(DEV), W, M, K, N)
    assert rel_err(gx, gy.double() @ W.double().cpu()) < 1e-5


def test_gemm_epilogues():
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(3)
    M, N, K, rps = 6 * 50, 120, 60, 50
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2, torch.randn(N, generator=g)
    res, rs = torch.randn(M, N, generator=g), torch.rand(M // rps, generator=g)
    aux = torch.empty(M, N, device=DEV)
    y = tr.gemm(_C.GEMM_NT, A.to(DEV), W.to(DEV), M, N, K, bias=b.to(DEV), res=res.to(DEV), aux_out=aux,
                row_scale=rs.to(DEV), rows_per_scale=rps, gelu=True)
    pre = A.double() @ W.double().t() + b.double()
    ref = F.gelu(pre) * rs.double().repeat_interleave(rps)[:, None] + res.double()
    assert rel_err(aux, pre) < 1e-5 and rel_err(y, ref) < 1e-5
    # backward-data epilogue: (g W) * GELU'(aux_in)
    gy = torch.randn(M, N, generator=g)
    h_pre = torch.randn(M, K, generator=g)
    out = tr.gemm(_C.GEMM_NN, gy.to(DEV), W.to(DEV), M, K, N, aux_in=h_pre.to(DEV))
    hp = h_pre.double().requires_grad_(True)
    F.gelu(hp).backward(gy.double() @ W.double())
    assert rel_err(out, hp.grad) < 1e-5


def test_gelu_forward_backward_vs_float64_erf():
    """`neosr_gelu` (and the GELU epilogues of neosr_gemm / the CAB kernels, which share its `gelu_f` / `gelu_grad_f`) use
    the Abramowitz-Stegun 7.1.26 erf (|err| <= 1.5e-7) with `__expf`, not libm `erff`: forward and derivative
    against float64 erf over [-10, 10], tails included.  Reference: nn.GELU() = x * Phi(x), neosr/archs/hat_arch.py:46,
    swinir_arch.py:23."""
    import math

    from neosr_amd.hip.transformer import Gelu

    x64 = torch.linspace(-10.0, 10.0, 200_001, dtype=torch.float64)
    x64 = torch.cat([x64, torch.tensor([0.0, -0.0, 1e-8, -1e-8, 1e-4, -1e-4], dtype=torch.float64)])
    phi = 0.5 * (1.0 + torch.erf(x64 / math.sqrt(2.0)))
    pdf = torch.exp(-0.5 * x64 * x64) / math.sqrt(2.0 * math.pi)
    y64 = x64 * phi
    d64 = phi + x64 * pdf
    x = x64.float().to(DEV).requires_grad_(True)
    y = Gelu.apply(x)
    g = torch.ones_like(y)
    (dx,) = torch.autograd.grad(y, x, g)
    torch.cuda.synchronize()
    ey = (y.detach().double().cpu() - y64).abs()
    ed = (dx.double().cpu() - d64).abs()
    # absolute bounds: erf error 1.5e-7 enters Phi as 0.75e-7, times |x| <= 10 in the forward; fp32 rounding of y on top
    assert float(ey.max()) <= 2e-6, float(ey.max())
    assert float((ey / y64.abs().clamp_min(1.0)).max()) <= 5e-7, float((ey / y64.abs().clamp_min(1.0)).max())
    assert float(ed.max()) <= 1e-6, float(ed.max())
    # deep negative tail: GELU -> -0 from below, never positive, never NaN
    tail = y.detach()[x.detach() < -6.0]
    assert bool(torch.isfinite(tail).all()) and float(tail.max()) <= 0.0 and float(tail.min()) > -1e-6


def test_linear_and_mlp_autograd_vs_torch():
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(4)
    B, HW, C, Hd = 3, 40, 60, 120
    x = torch.randn(B, HW, C, generator=g)
    w1, b1 = torch.randn(Hd, C, generator=g) * 0.1, torch.randn(Hd, generator=g) * 0.1
    w2, b2 = torch.randn(C, Hd, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    rs = torch.tensor([1 / 0.8, 0.0, 1 / 0.8])
    r = torch.randn(B, HW, C, generator=g)

    def run(dev, dt, fn, r=r):
        ts = [t.to(dev, dt).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = fn(*ts, rs.to(dev, dt))
        (y * r.to(dev, dt)).sum().backward()
        return [y] + [t.grad for t in ts]

    ref = run("cpu", torch.float64, lambda x_, a, b, c, d, s: x_ + s[:, None, None] * F.linear(
        F.gelu(F.linear(x_, a, b)), c, d))
    got = run(DEV, torch.float32, lambda x_, a, b, c, d, s: tr.mlp(x_, a, b, c, d, x_, s, HW))
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 1e-5
    r2 = torch.randn(B, HW, Hd, generator=g)
    ref = run("cpu", torch.float64, lambda x_, a, b, c, d, s: s[:, None, None] * F.linear(x_, a, b), r2)
    got = run(DEV, torch.float32, lambda x_, a, b, c, d, s: tr.linear(x_, a, b, None, s, HW), r2)
    for a, b in zip(got[:4], ref[:4]):
        assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("rows,C", [(1000, 180), (77, 60), (4096, 240), (5, 24)])
def test_layernorm_fwd_bwd(rows, C):
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g) * 2 + 0.5
    gamma, beta, r = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(rows, C, generator=g)

    def run(dev, dt, fn):
        ts = [t.to(dev, dt).requires_grad_(True) for t in (x, gamma, beta)]
        y = fn(*ts)
        (y * r.to(dev, dt)).sum().backward()
        return [y] + [t.grad for t in ts]

    ref = run("cpu", torch.float64, lambda a, b, c: F.layer_norm(a, (C,), b, c, 1e-5))
    got = run(DEV, torch.float32, lambda a, b, c: tr.layer_norm(a, b, c, 1e-5))
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 1e-5


def test_residual_layer_norm_sums_the_shortcut_gradient_in_the_kernel():
    """(shortcut, norm(x)) with both gradients summed inside the LayerNorm backward kernel == torch's LayerNorm with the
    shortcut gradient added by autograd"""
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 8, 16, 180, generator=g)
    gm, bt = torch.randn(180, generator=g), torch.randn(180, generator=g)
    gs, gy = torch.randn(3, 8, 16, 180, generator=g), torch.randn(3, 8, 16, 180, generator=g)
    xr = x.clone().requires_grad_(True)
    gmr, btr = gm.clone().requires_grad_(True), bt.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr.double(), (180,), gmr.double(), btr.double(), 1e-5)
    (xr.double() * gs.double()).sum().backward(retain_graph=True)
    (yr * gy.double()).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    gmd, btd = gm.to(DEV).requires_grad_(True), bt.to(DEV).requires_grad_(True)
    sc, y = tr.residual_layer_norm(xd, gmd, btd, 1e-5)
    assert torch.equal(sc, xd) and rel_err(y, yr) < 1e-5
    ((sc * gs.to(DEV)).sum() + (y * gy.to(DEV)).sum()).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-5 and rel_err(gmd.grad, gmr.grad) < 1e-5 and rel_err(btd.grad, btr.grad) < 1e-5
    # only one of the two outputs used
    xd.grad = None
    sc, y = tr.residual_layer_norm(xd, gmd, btd, 1e-5)
    (sc * gs.to(DEV)).sum().backward()
    assert rel_err(xd.grad, gs) < 1e-6


@pytest.mark.parametrize("shift", [0, 4])
@pytest.mark.parametrize("B,H,W,C,heads", [(2, 16, 24, 60, 6), (1, 8, 8, 24, 2), (2, 32, 16, 180, 6), (1, 16, 16, 64, 2), (1, 16, 8, 54, 6)])
def test_window_attention_fwd_bwd_vs_oracle(B, H, W, C, heads, shift):
    """kernel (addressing-folded roll / partition / head split, analytic index + mask) vs the oracle's
    roll -> window_partition -> softmax(qk^T + bias + mask) v -> window_reverse -> roll."""
    from neosr_amd.hip import transformer as tr
    from oracle import swinir_oracle as sorc

    g = torch.Generator().manual_seed(B * H + C + shift)
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    table = torch.randn(225, heads, generator=g) * 0.5
    r = torch.randn(B, H, W, C, generator=g)
    scale = (C // heads) ** -0.5

    def oracle(qkv_, table_):
        x = torch.roll(qkv_, (-shift, -shift), (1, 2)) if shift else qkv_
        xw = sorc.window_partition(x, 8).view(-1, 64, 3 * C)
        q, k, v = xw.reshape(-1, 64, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        attn = (q * scale) @ k.transpose(-2, -1)
        bias = table_[sorc.relative_position_index(8).view(-1)].view(64, 64, -1).permute(2, 0, 1)
        attn = attn + bias.unsqueeze(0)
        if shift:
            m = sorc.calculate_mask(H, W, 8, shift).to(attn.dtype)
            attn = (attn.view(B, m.shape[0], heads, 64, 64) + m[None, :, None]).view(-1, heads, 64, 64)
        o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, 8, 8, C)
        o = sorc.window_reverse(o, 8, H, W)
        return torch.roll(o, (shift, shift), (1, 2)) if shift else o

    a, t = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref = oracle(a, t)
    (ref * r.double()).sum().backward()
    a2, t2 = qkv.to(DEV).requires_grad_(True), table.to(DEV).requires_grad_(True)
    got = tr.window_attention(a2, t2, heads, 8, shift, scale)
    (got * r.to(DEV)).sum().backward()
    assert rel_err(got, ref) < 1e-5
    assert rel_err(a2.grad, a.grad) < 1e-5
    assert rel_err(t2.grad, t.grad) < 1e-5


def test_window_attention_bwd_is_deterministic():
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(9)
    qkv, table = torch.randn(2, 32, 32, 540, generator=g).to(DEV), torch.randn(225, 6, generator=g).to(DEV)
    r = torch.randn(2, 32, 32, 180, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        a, t = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
        (tr.window_attention(a, t, 6, 8, 4, 30 ** -0.5) * r).sum().backward()
        outs.append((a.grad.clone(), t.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_pixel_shuffle_nhwc_bit_exact():
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(1)
    for r_, c in ((2, 64), (4, 3), (3, 5)):
        x = torch.randn(2, c * r_ * r_, 6, 10, generator=g)
        ref = F.pixel_shuffle(x, r_)
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
        got = tr.PixelShuffleNHWC.apply(xin, r_)
        assert torch.equal(got.detach().permute(0, 3, 1, 2).cpu(), ref)
        got.backward(got.detach())  # adjoint: unshuffle(shuffle(x)) == x
        assert torch.equal(xin.grad.cpu(), xin.detach().cpu())


@pytest.mark.parametrize("shift", [0, 4])
def test_swin_block_vs_reference_fixture(shift):
    from neosr_amd.archs.swinir_arch import SwinTransformerBlock

    fix = load_golden("swinir_prims.npz")
    pre = f"blk_s{shift}"
    blk = SwinTransformerBlock(24, (16, 24), 2, 8, shift, 2.0)
    blk.load_state_dict(group(fix, f"{pre}/p"), strict=False)
    blk = blk.to(DEV).train()
    x = T(fix[f"{pre}/x"]).view(2, 16, 24, 24).to(DEV).requires_grad_(True)
    y = blk(x)
    (y * T(fix[f"{pre}/r"]).view(2, 16, 24, 24).to(DEV)).sum().backward()
    assert rel_err(y.view(2, -1, 24), T(fix[f"{pre}/y"])) < 1e-4
    assert rel_err(x.grad.view(2, -1, 24), T(fix[f"{pre}/gx"])) < 1e-3
    named = dict(blk.named_parameters())
    for k, g in group(fix, f"{pre}/g").items():
        assert rel_err(named[k].grad, g) < 1e-3, k


NET_CFG = {
    "ps": dict(embed_dim=24, upsampler="pixelshuffle", resi_connection="1conv"),
    "psd": dict(embed_dim=24, upsampler="pixelshuffledirect", resi_connection="1conv"),
    "nc": dict(embed_dim=32, upsampler="nearest+conv", resi_connection="3conv"),
}


@pytest.mark.parametrize("tag", list(NET_CFG))
def test_swinir_net_vs_reference_fixture(tag):
    from neosr_amd.archs.swinir_arch import swinir

    fix = load_golden("swinir_nets.npz")
    net = swinir(img_size=16, depths=(2, 2), num_heads=(2, 2), window_size=8, mlp_ratio=2.0, drop_path_rate=0.0,
                 upscale=4, **NET_CFG[tag])
    assert list(net.state_dict().keys()) == [str(k) for k in fix[f"{tag}/keys"]]
    net.load_state_dict(group(fix, f"{tag}/p"), strict=False)
    net = net.to(DEV).train()
    x = T(fix[f"{tag}/x"]).to(DEV).requires_grad_(True)
    y = net(x)
    (y * T(fix[f"{tag}/r"]).to(DEV)).sum().backward()
    assert rel_err(y, T(fix[f"{tag}/y"])) < 1e-4
    assert rel_err(x.grad, T(fix[f"{tag}/gx"])) < 1e-3
    named = dict(net.named_parameters())
    worst = max((rel_err(named[k].grad, g), k) for k, g in group(fix, f"{tag}/g").items())
    assert worst[0] < 1e-3, worst


def test_drop_path_train_mode_scales_rows():
    """train-mode DropPath: each sample's branch is dropped or scaled by 1/keep (arch_util.py:118-133);
    with the draws recorded, the block equals the oracle replaying them."""
    from neosr_amd.archs.swinir_arch import SwinTransformerBlock
    from oracle import swinir_oracle as sorc

    fix = load_golden("swinir_prims.npz")
    blk = SwinTransformerBlock(24, (16, 24), 2, 8, 4, 2.0, drop_path=0.5)
    P = group(fix, "blk_s4/p")
    blk.load_state_dict(P, strict=False)
    blk = blk.to(DEV).train()
    import neosr_amd.archs.swinir_arch as SA

    draws = []
    orig = SA.drop_scale
    SA.drop_scale = lambda p, tr, b, dev: draws.append(orig(p, tr, b, dev)) or draws[-1]
    x = T(fix["blk_s4/x"])
    x = torch.cat([x, x, x, x])  # 8 samples so both outcomes occur
    try:
        y = blk(x.view(8, 16, 24, 24).to(DEV))
    finally:
        SA.drop_scale = orig
    keep = [(d.cpu() * 0.5) for d in draws]
    assert all(set(k.tolist()) <= {0.0, 1.0} for k in keep)
    Pb = {f"b.{k}": v for k, v in P.items()}
    ref = sorc.swin_block(Pb, "b", x, (16, 24), 2, 8, 4, (keep[0], keep[1], 0.5))
    assert rel_err(y.view(8, -1, 24), ref) < 1e-4


def test_drop_path_bank_draws_all_sites_at_once():
    """From the second train-mode forward on, the DropPath scales of every site come from ONE (sites, batch) draw:
    rows are 0 or 1/keep with the site's own keep probability, the forward is reproducible from the torch seed, eval
    mode draws nothing."""
    from neosr_amd.archs.swinir_arch import swinir_small

    torch.manual_seed(3)
    net = swinir_small(upscale=4, drop_path_rate=0.3).to(DEV).train()
    x = torch.rand(4, 3, 16, 16, device=DEV)
    net(x)  # records the site sequence
    bank = net._dp_bank
    assert bank.recorded and len(bank.probs) == 2 * 23  # 24 blocks, the first one has drop_prob 0
    outs = []
    for _ in range(2):
        torch.manual_seed(11)
        outs.append(net(x).detach().clone())
    assert torch.equal(outs[0], outs[1])
    torch.manual_seed(11)
    bank.begin(True, 4, x.device)
    rows = bank._rows.cpu()
    bank.end(True)
    for p, r in zip(bank.probs, rows):
        assert all(abs(v) < 1e-6 or abs(v - 1.0 / (1.0 - p)) < 1e-5 for v in r.tolist())
    assert 0.02 < float((rows == 0).float().mean()) < 0.5
    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert torch.equal(a, b)


def test_image_model_trajectory_swinir_small_vs_reference_fixture():
    """2 x (feed_data + optimize_parameters) of OUR `image` model with network_g = swinir_small from the
    same seeded init as the reference run: loss, output and final weights."""
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden("step_swinir.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_swinir.toml")])
    model = build_model(opt)
    s = np.array([float(v.double().sum()) for v in model.net_g.state_dict().values()])
    np.testing.assert_allclose(s, fix["init/sum"], rtol=1e-5, atol=1e-5)
    for it in (1, 2):
        model.feed_data({"lq": T(fix[f"it{it}/lq"]), "gt": T(fix[f"it{it}/gt"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        ref = float(fix[f"it{it}/log/l_g_pix"])
        assert abs(log["l_g_pix"] - ref) < 1e-4 * ref
        assert rel_err(model.output, T(fix[f"it{it}/output"])) < 1e-3
    sd = model.net_g.state_dict()
    for k in [f for f in fix if f.startswith("final/w/")]:
        assert rel_err(sd[k[len("final/w/"):]], T(fix[k])) < 1e-3, k
