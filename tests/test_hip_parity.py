"""GPU parity tests: every HIP kernel / plan, called through the C ABI, against (a) the golden
fixtures produced by the reference and (b) the CPU oracle / plain PyTorch-CPU fp32 on seeded inputs.

Tolerance (BASELINE.json north_star): 1e-3 relative (per-tensor ||d||/||ref||) in fp32;
bit-exact for pixel-shuffle index math.  Our convs use exact-fp32 MFMA so the observed error is
~1e-6; the asserts use 1e-4 where only re-association differs and 1e-3 end-to-end.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import group, load_golden, rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _nhwc(t):  # (B,C,H,W) cpu -> (B,H,W,C) contiguous on device (canonical strides also when C == 1)
    B, C, H, W = t.shape
    return torch.empty(B, H, W, C).copy_(t.permute(0, 2, 3, 1)).to(DEV)


def _nchw(t):  # (B,H,W,C) device -> (B,C,H,W) cpu
    return t.permute(0, 3, 1, 2).contiguous().cpu()


CONV_CASES = [
    # B, H, W, K, N
    (2, 12, 20, 16, 8),
    (1, 5, 37, 3, 16),      # ragged width, K not multiple of 4 (scalar load path)
    (2, 16, 16, 24, 40),    # two N tiles, second half-empty
    (1, 64, 64, 64, 32),    # RDB conv1 shape
    (1, 32, 32, 192, 64),   # RDB conv5 shape (12 chunks)
    (1, 9, 33, 20, 70),     # 3 N tiles across 2 workgroups, ragged everything
    (2, 40, 70, 3, 64),     # thin-K forward (conv_first / VGG conv1_1), thin-N backward-data
    (2, 40, 70, 64, 3),     # thin-N forward (conv_last), thin-K backward-data
    (1, 33, 65, 64, 1),     # U-Net conv9: one output channel
    (1, 21, 130, 4, 180),   # swinir conv_first-like: 3 n-blocks
    (2, 24, 40, 1, 64),     # one input channel (backward-data of U-Net conv9): channel stride 1
]


@pytest.mark.parametrize("B,H,W,K,N", CONV_CASES)
def test_conv3x3_fwd_bias_lrelu(B, H, W, K, N):
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B * 1000 + K)
    x = torch.randn(B, K, H, W, generator=g)
    w = torch.randn(N, K, 3, 3, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    ref = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    out = ops.conv3x3(_nhwc(x), w.to(DEV), b.to(DEV), act=ops.ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out), ref) < 1e-5


def test_conv3x3_fwd_concat_slice_and_residuals():
    """RDB-style: read first K channels of a wide buffer, write a channel slice, scaled residuals."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(5)
    B, H, W, CC, K, N = 2, 10, 34, 48, 32, 16
    buf = torch.randn(B, CC, H, W, generator=g)
    w = torch.randn(N, K, 3, 3, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    r1 = torch.randn(B, N, H, W, generator=g)
    r2 = torch.randn(B, N, H, W, generator=g)
    ref = (F.conv2d(buf[:, :K], w, b, padding=1) * 0.2 + r1) * 0.2 + r2
    dbuf = _nhwc(buf)
    ops.conv3x3(dbuf, w.to(DEV), b.to(DEV), out=dbuf[..., K:K + N], k_in=K, alpha=0.2,
                res1=_nhwc(r1), alpha2=0.2, res2=_nhwc(r2))
    torch.cuda.synchronize()
    got = _nchw(dbuf)
    assert rel_err(got[:, K:K + N], ref) < 1e-5
    assert torch.equal(got[:, :K], buf[:, :K])            # untouched
    assert torch.equal(got[:, K + N:], buf[:, K + N:])    # untouched


def test_conv3x3_fwd_nearest_upsample_fused():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 16, 9, 17, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) * 0.1
    b = torch.randn(16, generator=g)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1), 0.2)
    out = ops.conv3x3(_nhwc(x), w.to(DEV), b.to(DEV), ups=True, act=ops.ACT_LRELU, slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(_nchw(out), ref) < 1e-5


def test_conv3x3_fwd_prelu_on_load():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(8)
    z = torch.randn(2, 16, 8, 12, generator=g)
    s = torch.rand(16, generator=g) - 0.3
    w = torch.randn(8, 16, 3, 3, generator=g) * 0.1
    ref = F.conv2d(F.prelu(z, s), w, None, padding=1)
    out = ops.conv3x3(_nhwc(z), w.to(DEV), None, in_prelu=s.to(DEV))
    torch.cuda.synchronize()
    assert rel_err(_nchw(out), ref) < 1e-5


@pytest.mark.parametrize("B,H,W,K,N", CONV_CASES)
def test_conv3x3_dgrad_and_wgrad_vs_autograd(B, H, W, K, N):
    """y = lrelu(conv(x)); given dL/dy: dx (mask on load + transposed kernel), dw, db."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B * 77 + N)
    x = torch.randn(B, K, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    y = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dy, dact = _nhwc(gy), _nhwc(y.detach())
    dx = ops.conv3x3(dy, w.detach().to(DEV), None, mode=ops.CONV_DGRAD, in_mask=dact, mask_slope=0.2)
    dw, db = ops.conv3x3_wgrad(_nhwc(x.detach()), dy, N, K, g_mask=dact, mask_slope=0.2)
    torch.cuda.synchronize()
    assert rel_err(_nchw(dx), x.grad) < 1e-5
    assert rel_err(dw.cpu(), w.grad) < 1e-5
    assert rel_err(db.cpu(), b.grad) < 1e-5


PACK_CASES = [
    # B, H, W, K, N  (K, N multiples of 4: the direct-to-LDS kernel moves 16-byte granules)
    (2, 12, 20, 16, 8),
    (1, 9, 37, 20, 44),     # ragged tile edges, K not a multiple of 16, N not a multiple of 32
    (1, 64, 64, 64, 32),    # RDB conv1
    (2, 32, 32, 192, 64),   # RDB conv5: 12 chunks, two n-blocks
    (1, 16, 16, 160, 32),
]


@pytest.mark.parametrize("B,H,W,K,N", PACK_CASES)
def test_conv3x3_packed_fwd_and_dgrad(B, H, W, K, N):
    """direct-to-LDS kernel (w_pack) against autograd, both modes, with the gather-form epilogue:
    g_in = lrelu'(a) * (dgrad + residual)."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B * 31 + K + N)
    x = torch.randn(B, K, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    r = torch.randn(B, N, H, W, generator=g)
    pre = F.conv2d(x, w, b, padding=1)
    ref = F.leaky_relu(pre, 0.2) * 0.2 + r
    wd = w.to(DEV)
    out = ops.conv3x3(_nhwc(x.detach()), wd, b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2,
                      res1=_nhwc(r), w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_FWD))
    torch.cuda.synchronize()
    assert rel_err(_nchw(out), ref.detach()) < 1e-5
    # backward-data of the plain conv, then gated by the derivative of a LeakyReLU fed by `a`
    gy = torch.randn(B, N, H, W, generator=g)
    a = torch.randn(B, K, H, W, generator=g)
    (dx,) = torch.autograd.grad(F.conv2d(x, w, None, padding=1), x, gy)
    ref_g = torch.where(a > 0, dx, dx * 0.2)
    gin = ops.conv3x3(_nhwc(gy), wd, None, mode=ops.CONV_DGRAD, out_mask=_nhwc(a), out_mask_slope=0.2,
                      w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_DGRAD))
    torch.cuda.synchronize()
    assert rel_err(_nchw(gin), ref_g) < 1e-5


@pytest.mark.parametrize("B,H,W,K,N", PACK_CASES + [(1, 13, 21, 96, 12), (3, 8, 16, 128, 32)])
def test_conv3x3_winograd_fwd_and_dgrad(B, H, W, K, N):
    """Winograd F(2x2,3x3) kernel (w_wino) against autograd (float64), both modes, the same epilogues as the
    direct-to-LDS kernel; odd sizes, ragged K / N.  Not the direct form's summation order: 1e-4 (observed ~1e-6)."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B * 37 + K + N)
    x = torch.randn(B, K, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    r = torch.randn(B, N, H, W, generator=g)
    pre = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.leaky_relu(pre, 0.2) * 0.2 + r.double()
    wd = w.to(DEV)
    lib = _C_lib()
    prev_mode = lib.neosr_set_winograd(1)
    out = ops.conv3x3(_nhwc(x.detach()), wd, b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2, res1=_nhwc(r),
                      w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_FWD), w_wino=ops.conv3x3_pack_wino(wd, ops.CONV_FWD))
    direct = ops.conv3x3(_nhwc(x.detach()), wd, b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2, res1=_nhwc(r),
                         w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_FWD))
    torch.cuda.synchronize()
    assert rel_err(_nchw(out), ref.detach()) < 1e-4
    assert not torch.equal(out, direct) or K * N < 200   # it really was the other kernel
    assert rel_err(out.cpu(), direct.cpu()) < 1e-4
    gy = torch.randn(B, N, H, W, generator=g)
    a = torch.randn(B, K, H, W, generator=g)
    (dx,) = torch.autograd.grad(F.conv2d(x.double(), w.double(), None, padding=1), x, gy.double())
    ref_g = torch.where(a > 0, dx, dx * 0.2)
    gin = ops.conv3x3(_nhwc(gy), wd, None, mode=ops.CONV_DGRAD, out_mask=_nhwc(a), out_mask_slope=0.2,
                      w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_DGRAD), w_wino=ops.conv3x3_pack_wino(wd, ops.CONV_DGRAD))
    torch.cuda.synchronize()
    lib.neosr_set_winograd(prev_mode)
    assert rel_err(_nchw(gin), ref_g) < 1e-4


@pytest.mark.parametrize("B,H,W,K,N", PACK_CASES + [(1, 13, 21, 96, 12), (3, 8, 16, 128, 32), (2, 64, 64, 192, 64),
                                                   (1, 16, 16, 16, 32), (2, 35, 50, 48, 40), (1, 24, 40, 64, 96),
                                                   (2, 16, 32, 180, 180)])
def test_conv3x3_winograd4_fwd_and_dgrad(B, H, W, K, N):
    """Winograd F(4x4,3x3) kernel (w_wino4, neosr_set_winograd(2)) against autograd in float64, both modes, the same
    epilogues as the direct-to-LDS kernel; sizes that are not multiples of the 16-pixel tile, ragged K / N, K not a
    multiple of the 32-channel chunk.  The larger transforms amplify rounding ~10x more than F(2x2,3x3): gate 1e-4 of
    the tensor norm (observed ~2e-6), and <= 1e-4 of the output scale element-wise."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B * 41 + K + N)
    x = torch.randn(B, K, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    r = torch.randn(B, N, H, W, generator=g)
    pre = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = F.leaky_relu(pre, 0.2) * 0.2 + r.double()
    wd = w.to(DEV)
    lib = _C_lib()
    prev = lib.neosr_set_winograd(2)
    try:
        pk = ops.conv3x3_pack_weights(wd, ops.CONV_FWD)
        out = ops.conv3x3(_nhwc(x.detach()), wd, b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2, res1=_nhwc(r),
                          w_pack=pk, w_wino4=ops.conv3x3_pack_wino4(wd, ops.CONV_FWD))
        direct = ops.conv3x3(_nhwc(x.detach()), wd, b.to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2, res1=_nhwc(r),
                             w_pack=pk)
        torch.cuda.synchronize()
        assert rel_err(_nchw(out), ref.detach()) < 1e-4
        assert not torch.equal(out, direct) or K * N < 200   # it really was the other kernel
        assert (_nchw(out).double() - ref.detach()).abs().max() < 1e-4 * ref.detach().abs().max()
        gy = torch.randn(B, N, H, W, generator=g)
        a = torch.randn(B, K, H, W, generator=g)
        (dx,) = torch.autograd.grad(F.conv2d(x.double(), w.double(), None, padding=1), x, gy.double())
        ref_g = torch.where(a > 0, dx, dx * 0.2)
        gin = ops.conv3x3(_nhwc(gy), wd, None, mode=ops.CONV_DGRAD, out_mask=_nhwc(a), out_mask_slope=0.2,
                          w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_DGRAD),
                          w_wino4=ops.conv3x3_pack_wino4(wd, ops.CONV_DGRAD))
        torch.cuda.synchronize()
        assert rel_err(_nchw(gin), ref_g) < 1e-4
        assert (_nchw(gin).double() - ref_g).abs().max() < 1e-4 * ref_g.abs().max()
    finally:
        lib.neosr_set_winograd(prev)


def test_conv3x3_winograd4_nearest_upsampled_input():
    """nearest x2 upsampling folded into the F(4x4,3x3) kernel's DMA addresses (esrgan_arch.py:207-212 conv_up1 / conv_up2)"""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(31)
    B, H, W, K, N = 2, 12, 20, 64, 64
    x = torch.randn(B, K, H, W, generator=g)
    w = torch.randn(N, K, 3, 3, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    ref = F.leaky_relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1), 0.2)
    wd = w.to(DEV)
    lib = _C_lib()
    prev = lib.neosr_set_winograd(2)
    try:
        out = ops.conv3x3(_nhwc(x), wd, b.to(DEV), ups=True, act=ops.ACT_LRELU, slope=0.2,
                          w_pack=ops.conv3x3_pack_weights(wd, ops.CONV_FWD), w_wino4=ops.conv3x3_pack_wino4(wd, ops.CONV_FWD))
        direct = ops.conv3x3(_nhwc(x), wd, b.to(DEV), ups=True, act=ops.ACT_LRELU, slope=0.2)
    finally:
        lib.neosr_set_winograd(prev)
    torch.cuda.synchronize()
    assert out.shape == (B, 2 * H, 2 * W, N) and not torch.equal(out, direct)
    assert rel_err(_nchw(out), ref) < 1e-4


def test_conv3x3_winograd4_slices_residuals_accumulate_and_modes():
    """prefix-K read of a wide buffer, slice write, two residuals, accumulate through the F(4x4,3x3) kernel;
    neosr_set_winograd(0 / 1 / 2) routes the same descriptor to the direct / F(2x2) / F(4x4) kernel"""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(29)
    B, H, W, CC, K, N = 2, 20, 36, 96, 64, 32
    buf = _nhwc(torch.randn(B, CC, H, W, generator=g))
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r2 = _nhwc(torch.randn(B, N, H, W, generator=g))
    pack, wino = ops.conv3x3_pack_weights(w, ops.CONV_FWD), ops.conv3x3_pack_wino(w, ops.CONV_FWD)
    wino4 = ops.conv3x3_pack_wino4(w, ops.CONV_FWD)
    lib = _C_lib()
    outs = []
    for on, both in ((0, True), (1, True), (2, False), (2, False), (2, True)):
        prev = lib.neosr_set_winograd(on)
        assert lib.neosr_get_winograd() == on
        o = buf.clone()
        ops.conv3x3(o[..., :K], w, b, out=o[..., K:K + N], alpha=0.2, res1=o[..., :N], alpha2=0.5, res2=r2,
                    accumulate=True, w_pack=pack, w_wino=wino if both else None, w_wino4=wino4)
        outs.append(o)
        assert lib.neosr_set_winograd(prev) == on
    torch.cuda.synchronize()
    assert torch.equal(outs[2], outs[3])                      # deterministic
    assert not torch.equal(outs[0], outs[2]) and not torch.equal(outs[1], outs[2])
    assert rel_err(outs[2].cpu(), outs[0].cpu()) < 1e-5
    assert torch.equal(outs[2][..., :K], buf[..., :K])
    # a launch this small (12 workgroups of 16 x 16 pixels x 32 channels < NEOSR_WINO4_MIN_WGS) keeps F(2x2,3x3) when it
    # is offered both images
    assert torch.equal(outs[4], outs[1])


def test_conv3x3_winograd_slices_residuals_accumulate_and_toggle():
    """prefix-K read of a wide buffer, slice write, two residuals, accumulate; neosr_set_winograd(0) routes the same
    descriptor to the direct kernel"""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(23)
    B, H, W, CC, K, N = 2, 20, 36, 96, 64, 32
    buf = _nhwc(torch.randn(B, CC, H, W, generator=g))
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r2 = _nhwc(torch.randn(B, N, H, W, generator=g))
    pack, wino = ops.conv3x3_pack_weights(w, ops.CONV_FWD), ops.conv3x3_pack_wino(w, ops.CONV_FWD)
    lib = _C_lib()
    outs = []
    for on in (0, 1, 0):
        prev = lib.neosr_set_winograd(on)
        o = buf.clone()
        ops.conv3x3(o[..., :K], w, b, out=o[..., K:K + N], alpha=0.2, res1=o[..., :N], alpha2=0.5, res2=r2,
                    accumulate=True, w_pack=pack, w_wino=wino)
        outs.append(o)
        lib.neosr_set_winograd(prev)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])
    assert rel_err(outs[1].cpu(), outs[0].cpu()) < 1e-5
    assert torch.equal(outs[1][..., :K], buf[..., :K])


def _C_lib():
    from neosr_amd import _C

    return _C.load()


def test_conv3x3_packed_matches_staged_kernel_on_slices():
    """same launch through both kernels: prefix-K read of a wide buffer, slice write, two residuals"""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(21)
    B, H, W, CC, K, N = 2, 20, 36, 96, 64, 32
    buf = _nhwc(torch.randn(B, CC, H, W, generator=g))
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    r2 = _nhwc(torch.randn(B, N, H, W, generator=g))
    outs = []
    for pack in (None, ops.conv3x3_pack_weights(w, ops.CONV_FWD)):
        o = buf.clone()
        ops.conv3x3(o[..., :K], w, b, out=o[..., K:K + N], alpha=0.2, res1=o[..., :N], alpha2=0.5,
                    res2=r2, w_pack=pack)
        outs.append(o)
    torch.cuda.synchronize()
    assert rel_err(outs[1].cpu(), outs[0].cpu()) < 1e-6
    assert torch.equal(outs[1][..., :K], buf[..., :K])


def test_conv3x3_dgrad_accumulate_into_slice():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(9)
    B, H, W, CC, Kc, Nc = 1, 8, 40, 48, 16, 32   # grad of a 32->16 conv accumulated into chans [0,32)
    gb = torch.randn(B, CC, H, W, generator=g)
    w = torch.randn(Kc, Nc, 3, 3, generator=g) * 0.1
    ref = gb.clone()
    ref[:, :Nc] += F.conv_transpose2d(gb[:, 32:48], w, padding=1)
    dgb = _nhwc(gb)
    ops.conv3x3(dgb[..., 32:48], w.to(DEV), None, mode=ops.CONV_DGRAD, out=dgb[..., :Nc], accumulate=True)
    torch.cuda.synchronize()
    assert rel_err(_nchw(dgb), ref) < 1e-5


@pytest.mark.parametrize("B,H,W,K,N", [(2, 40, 70, 64, 3), (1, 33, 65, 64, 1), (2, 40, 70, 3, 64),
                                       (1, 21, 130, 4, 180), (1, 17, 200, 96, 3)])
def test_wgrad_thin_layers_vs_autograd(B, H, W, K, N):
    """first / last layers (3 or 1 channels on one side): the 4x4x1-MFMA weight-gradient path, no mask"""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(K * 7 + N)
    x = torch.randn(B, K, H, W, generator=g)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(N, generator=g).requires_grad_(True)
    gy = torch.randn(B, N, H, W, generator=g)
    F.conv2d(x, w, b, padding=1).backward(gy)
    dw, db = ops.conv3x3_wgrad(_nhwc(x), _nhwc(gy), N, K)
    dw2, db2 = ops.conv3x3_wgrad(_nhwc(x), _nhwc(gy), N, K)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), w.grad) < 1e-5
    assert rel_err(db.cpu(), b.grad) < 1e-5
    assert torch.equal(dw, dw2) and torch.equal(db, db2)  # fixed-order reductions


WGRAD4_CASES = [
    # B, H, W, K, N, ups
    (2, 12, 20, 16, 8, 0), (1, 9, 37, 20, 44, 0), (1, 64, 64, 64, 32, 0), (2, 33, 47, 96, 64, 0), (1, 5, 3, 32, 32, 0),
    (3, 16, 16, 160, 32, 0), (2, 6, 10, 8, 12, 1), (1, 32, 48, 64, 64, 1), (1, 4, 16, 32, 32, 0), (1, 7, 100, 36, 68, 0),
]


@pytest.mark.parametrize("B,H,W,K,N,ups", WGRAD4_CASES)
def test_wgrad_winograd4_vs_autograd_float64(B, H, W, K, N, ups):
    """F(4x4-tile) weight-gradient kernel (neosr_set_winograd(2) + neosr_set_wgrad4(1)) against autograd in float64:
    ragged units, K / N that are no multiple of 32, nearest-upsampled input; its error is ~8e-7 of the gradient's norm
    (the F(2x2) form: ~2e-7); run-to-run bit-identical; neosr_set_wgrad4(0) routes the same call to the F(2x2) kernel"""
    from neosr_amd import _C
    from neosr_amd.hip import ops

    lib = _C.load()
    g = torch.Generator().manual_seed(B * 7 + N)
    x = torch.randn(B, K, H // 2 if ups else H, W // 2 if ups else W, generator=g, dtype=torch.float64)
    w = (torch.randn(N, K, 3, 3, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.randn(N, generator=g, dtype=torch.float64).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    y = F.conv2d(xin, w, b, padding=1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xs, gs = _nhwc(x.float()), _nhwc(gy.float())
    prev_mode = lib.neosr_set_winograd(2)
    prev = lib.neosr_set_wgrad4(1)
    try:
        dw, db = ops.conv3x3_wgrad(xs, gs, N, K, ups=bool(ups))
        dw2, db2 = ops.conv3x3_wgrad(xs, gs, N, K, ups=bool(ups))
        assert lib.neosr_set_wgrad4(0) == 1
        dw_f2, db_f2 = ops.conv3x3_wgrad(xs, gs, N, K, ups=bool(ups))
        torch.cuda.synchronize()
    finally:
        lib.neosr_set_wgrad4(prev)
        lib.neosr_set_winograd(prev_mode)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    assert rel_err(dw.cpu().double(), w.grad) < 2e-6
    assert rel_err(db.cpu().double(), b.grad) < 1e-6
    assert rel_err(dw_f2.cpu().double(), w.grad) < 1e-6
    assert not torch.equal(dw, dw_f2)  # a different kernel ran


def test_wgrad_winograd4_prefix_read_slice_accumulate_scale():
    """prefix-K read of a wider activation buffer, gradient = channel slice of a wider buffer, accumulate, scale"""
    from neosr_amd import _C
    from neosr_amd.hip import ops

    lib = _C.load()
    g = torch.Generator().manual_seed(5)
    xw = torch.randn(2, 24, 40, 96, generator=g).to(DEV)
    gy = torch.randn(2, 24, 40, 64, generator=g).to(DEV)[..., :32]
    wref = torch.zeros(32, 64, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xw[..., :64].permute(0, 3, 1, 2).double().cpu(), wref, None, padding=1).backward(
        gy.permute(0, 3, 1, 2).double().cpu())
    prev_mode = lib.neosr_set_winograd(2)
    prev = lib.neosr_set_wgrad4(1)
    try:
        dw0 = torch.ones(32, 64, 3, 3, device=DEV)
        dw, _ = ops.conv3x3_wgrad(xw, gy, 32, 64, dw=dw0, want_bias=False, accumulate=True, scale=0.5)
        torch.cuda.synchronize()
    finally:
        lib.neosr_set_wgrad4(prev)
        lib.neosr_set_winograd(prev_mode)
    assert rel_err(dw.cpu().double() - 1.0, 0.5 * wref.grad) < 2e-6


def test_wgrad_upsampled_input():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 8, 6, 10, generator=g)
    w = (torch.randn(12, 8, 3, 3, generator=g) * 0.1).requires_grad_(True)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, None, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dw, _ = ops.conv3x3_wgrad(_nhwc(x), _nhwc(gy), 12, 8, ups=True, scale=0.5, want_bias=False)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), 0.5 * w.grad) < 1e-5


def test_wgrad_is_run_to_run_deterministic():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(12)
    x = _nhwc(torch.randn(2, 64, 32, 32, generator=g))
    gy = _nhwc(torch.randn(2, 32, 32, 32, generator=g))
    a, _ = ops.conv3x3_wgrad(x, gy, 32, 64)
    b, _ = ops.conv3x3_wgrad(x, gy, 32, 64)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_pixel_shuffle_bit_exact_vs_reference_fixture():
    from neosr_amd.hip import ops

    fix = load_golden("index.npz")
    for r in (2, 4):
        x = torch.from_numpy(fix[f"ps{r}_in"])
        out = ops.pixel_shuffle(_nhwc(x), r)
        assert np.array_equal(out.cpu().numpy(), fix[f"ps{r}_out"])
        back = ops.pixel_unshuffle(out, r)
        assert torch.equal(_nchw(back), x)


def test_layout_and_pool_kernels():
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 3, 7, 11, generator=g)
    assert torch.equal(_nchw(ops.nchw_to_nhwc(x.to(DEV), 4))[:, :3], x)
    assert torch.equal(ops.nhwc_to_nchw(_nhwc(x)).cpu(), x)
    y = torch.randn(2, 8, 6, 10, generator=g)
    ref = F.avg_pool2d(y, 2) * 4
    assert rel_err(_nchw(ops.pool2x2_sum(_nhwc(y))), ref) < 1e-6


def test_l1_loss_vs_reference_fixture():
    from neosr_amd.losses import build_loss

    fix = load_golden("l1loss.npz")
    pred = torch.from_numpy(fix["pred"]).to(DEV).requires_grad_(True)
    crit = build_loss({"type": "L1Loss", "loss_weight": 0.7})
    out = crit(pred, torch.from_numpy(fix["target"]).to(DEV))
    out.backward()
    assert abs(out.item() - float(fix["loss"])) < 1e-5 * float(fix["loss"])
    assert rel_err(pred.grad, torch.from_numpy(fix["grad"])) < 1e-6


def _load_into(net, params):
    net.load_state_dict(params)
    return net.to(DEV)


def _arch_vs_golden(fixname, net):
    from oracle import neosr_oracle as orc

    fix = load_golden(fixname)
    P = group(fix, "param")
    net = _load_into(net, P).train()
    x, gt = torch.from_numpy(fix["x"]).to(DEV), torch.from_numpy(fix["gt"]).to(DEV)
    y = net(x)
    loss = F.l1_loss(y, gt)
    loss.backward()
    torch.cuda.synchronize()
    assert rel_err(y, torch.from_numpy(fix["y"])) < 1e-4
    assert abs(loss.item() - float(fix["loss"])) < 1e-4 * float(fix["loss"])
    G = group(fix, "grad")
    named = dict(net.named_parameters())
    worst = max(rel_err(named[k].grad, g) for k, g in G.items())
    assert worst < 1e-3, worst
    del orc


def test_esrgan_plan_vs_reference_fixture():
    from neosr_amd.archs import build_network

    net = build_network({"type": "esrgan", "num_in_ch": 3, "num_out_ch": 3, "scale": 4,
                         "num_feat": 16, "num_block": 2, "num_grow_ch": 8})
    _arch_vs_golden("esrgan_small.npz", net)


@pytest.mark.parametrize("act", ["prelu", "leakyrelu", "relu"])
def test_compact_plan_vs_reference_fixture(act):
    from neosr_amd.archs import build_network

    net = build_network({"type": "compact", "num_feat": 16, "num_conv": 3, "upscale": 4,
                         "act_type": act})
    _arch_vs_golden(f"compact_small_{act}.npz", net)


def test_esrgan_inference_ring_matches_training_forward():
    from neosr_amd.archs import build_network

    torch.manual_seed(3)
    net = build_network({"type": "esrgan", "scale": 4, "num_feat": 16, "num_block": 3,
                         "num_grow_ch": 8}).to(DEV)
    x = torch.rand(1, 3, 16, 24, device=DEV)
    net.train()
    y_train = net(x)
    net.eval()
    with torch.no_grad():
        y_eval = net(x)
    assert torch.equal(y_train.detach(), y_eval)


def test_esrgan_full_size_vs_oracle():
    """Default RRDBNet (23 blocks, 64 feat) on one 64x64 LR patch: output + all 702 grads vs CPU oracle."""
    from neosr_amd.archs import build_network
    from oracle import neosr_oracle as orc

    torch.manual_seed(1024)
    net = build_network({"type": "esrgan", "scale": 4})
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.rand(1, 3, 64, 64)
    gt = torch.rand(1, 3, 256, 256)
    y_ref = orc.rrdbnet_forward(P, x, 4)
    orc.l1_loss(y_ref, gt).backward()
    net = net.to(DEV).train()
    y = net(x.to(DEV))
    F.l1_loss(y, gt.to(DEV)).backward()
    torch.cuda.synchronize()
    assert rel_err(y, y_ref) < 1e-4
    named = dict(net.named_parameters())
    errs = {k: rel_err(named[k].grad, P[k].grad) for k in P}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-3, (worst, errs[worst])


@pytest.mark.parametrize("scale", [2, 1])
def test_esrgan_scale_2_and_1_vs_oracle(scale):
    """esrgan with `scale` 2 / 1: pixel-unshuffled input (12 / 48 channels into conv_first, esrgan_arch.py:
    197-200), output + all gradients incl. the input gradient against the CPU oracle."""
    from neosr_amd.archs import build_network
    from oracle import neosr_oracle as orc

    torch.manual_seed(5)
    net = build_network({"type": "esrgan", "num_feat": 32, "num_block": 2, "num_grow_ch": 16, "scale": scale})
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.rand(2, 3, 32, 48).requires_grad_(True)
    y_ref = orc.rrdbnet_forward(P, x, scale)
    gt = torch.rand_like(y_ref)
    orc.l1_loss(y_ref, gt).backward()
    net = net.to(DEV).train()
    xd = x.detach().to(DEV).requires_grad_(True)
    y = net(xd)
    assert y.shape == y_ref.shape == (2, 3, 32 * scale, 48 * scale)
    F.l1_loss(y, gt.to(DEV)).backward()
    torch.cuda.synchronize()
    assert rel_err(y, y_ref) < 1e-4
    assert rel_err(xd.grad, x.grad) < 1e-3
    named = dict(net.named_parameters())
    errs = {k: rel_err(named[k].grad, P[k].grad) for k in P}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-3, (worst, errs[worst])


@pytest.mark.parametrize("upscale,act", [(2, "prelu"), (3, "leakyrelu"), (1, "relu"), (3, "prelu"), (1, "prelu")])
def test_compact_other_scales_vs_oracle(upscale, act):
    """compact with upscale 1 / 2 / 3 (PixelShuffle factor, nearest-upsampled residual; compact_arch.py:56-85)."""
    from neosr_amd.archs import build_network
    from oracle import neosr_oracle as orc

    torch.manual_seed(9)
    net = build_network({"type": "compact", "num_feat": 16, "num_conv": 3, "upscale": upscale, "act_type": act})
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.rand(2, 3, 20, 36)
    y_ref = orc.compact_forward(P, x, upscale, act)
    gt = torch.rand_like(y_ref)
    orc.l1_loss(y_ref, gt).backward()
    net = net.to(DEV).train()
    y = net(x.to(DEV))
    F.l1_loss(y, gt.to(DEV)).backward()
    torch.cuda.synchronize()
    assert y.shape == y_ref.shape and rel_err(y, y_ref) < 1e-4
    named = dict(net.named_parameters())
    errs = {k: rel_err(named[k].grad, P[k].grad) for k in P}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-3, (worst, errs[worst])


def test_esrgan_launch_chains_do_not_change_results():
    """`neosr_set_num_streams`: the two batch-half chains (+ the weight-gradient stream) give bit-identical
    outputs and gradients to the single-stream schedule, run after run (odd batch: halves of 1 and 2)."""
    from neosr_amd import _C
    from neosr_amd.archs import build_network

    lib = _C.load()
    torch.manual_seed(3)
    net = build_network({"type": "esrgan", "num_feat": 32, "num_block": 2, "num_grow_ch": 16, "scale": 4}).to(DEV).train()
    x = torch.rand(3, 3, 24, 40, device=DEV)
    gy = torch.randn(3, 3, 96, 160, device=DEV)
    runs = []
    prev = lib.neosr_set_num_streams(2)
    try:
        for ns in (1, 2, 2):
            lib.neosr_set_num_streams(ns)
            net.zero_grad(set_to_none=True)
            y = net(x)
            y.backward(gy)
            torch.cuda.synchronize()
            runs.append((y.detach().clone(), [p.grad.clone() for p in net.parameters()]))
    finally:
        lib.neosr_set_num_streams(prev)
    for y, grads in runs[1:]:
        assert torch.equal(y, runs[0][0])
        assert all(torch.equal(a, b) for a, b in zip(grads, runs[0][1]))


def test_adamw_clip_ema_step_vs_oracle():
    from neosr_amd.hip.nets import arena_layout, flatten_parameters_
    from neosr_amd.optimizers import AdamW
    from oracle import neosr_oracle as orc

    torch.manual_seed(4)
    shapes = [(16, 3, 3, 3), (16,), (8, 16, 3, 3), (8,), (5,)]
    ps = [torch.randn(s) for s in shapes]
    gs = [torch.randn(s) * 3 for s in shapes]
    mod = torch.nn.ParameterList([torch.nn.Parameter(p.clone()) for p in ps]).to(DEV)
    flatten_parameters_(mod)
    opt = AdamW(list(mod.parameters()), lr=1e-2, betas=(0.9, 0.99), weight_decay=0.05)
    offs, total_elems = arena_layout(ps)  # 16-byte aligned starts; the (5,) tensor leaves a 3-element pad
    ema = torch.zeros(total_elems, device=DEV)
    ref_p = [p.clone() for p in ps]
    ref_m = [torch.zeros_like(p) for p in ps]
    ref_v = [torch.zeros_like(p) for p in ps]
    ref_e = [torch.zeros_like(p) for p in ps]
    for step in range(1, 4):
        for p, g in zip(mod.parameters(), gs):
            p.grad = (g * step).to(DEV)
        opt.set_clip(1.0)
        opt.set_ema(ema, 0.9, first=step == 1)
        opt.step()
        rg = [(g * step).clone() for g in gs]
        total = orc.clip_grad_norm_(rg, 1.0)
        orc.adamw_step(ref_p, rg, ref_m, ref_v, step, 1e-2, (0.9, 0.99), 1e-8, 0.05)
        orc.ema_update(ref_e, ref_p, 0.9, first=step == 1)
        assert abs(opt.last_grad_norm.item() - total) < 1e-5 * total
    torch.cuda.synchronize()
    for p, r in zip(mod.parameters(), ref_p):
        assert rel_err(p, r) < 1e-5
    for e, off in zip(ref_e, offs):
        assert rel_err(ema[off: off + e.numel()], e.flatten()) < 1e-5


@pytest.mark.parametrize("arch", ["compact", "esrgan"])
def test_image_model_trajectory_vs_reference_fixture(arch):
    """3 x (feed_data + optimize_parameters) of OUR `image` model from the reference's initial
    weights and batches: per-iteration loss, output, final weights and EMA vs the reference run."""
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden(f"step_{arch}.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / f"golden_{arch}.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    for it in range(1, 4):
        model.feed_data({"lq": torch.from_numpy(fix[f"lq{it}"]), "gt": torch.from_numpy(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert abs(log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-4 * fix["log"][it - 1, 0]
        assert rel_err(model.output, torch.from_numpy(fix[f"out{it}"])) < 1e-3
    final, ema = group(fix, "final"), group(fix, "ema")
    sd = model.net_g.state_dict()
    esd = model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in final.items()) < 1e-3
    assert max(rel_err(esd[k], v) for k, v in ema.items() if k != "n_averaged") < 1e-3
    assert int(esd["n_averaged"]) == 3


def test_image_model_eco_trajectory_vs_reference_fixture():
    """`train.eco` (image.py:393-418): 5 iterations around eco_init = 2 / eco_iters = 4 — a plain step, three ECO steps
    (no-grad prediction, GT and LQ centroids, step on the prediction from the LQ centroid), a plain step — vs the reference"""
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden("step_eco.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_eco.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    for it in range(1, 6):
        model.feed_data({"lq": torch.from_numpy(fix[f"lq{it}"]), "gt": torch.from_numpy(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert abs(log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-4 * fix["log"][it - 1, 0], it
        assert rel_err(model.gt, torch.from_numpy(fix[f"gt_used{it}"])) < 1e-5, it  # the GT centroid of iterations 2-4
        assert rel_err(model.output, torch.from_numpy(fix[f"out{it}"])) < 1e-3, it
    final, ema = group(fix, "final"), group(fix, "ema")
    sd, esd = model.net_g.state_dict(), model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in final.items()) < 1e-3
    assert max(rel_err(esd[k], v) for k, v in ema.items() if k != "n_averaged") < 1e-3


def test_pack_many_equals_per_layer_packing():
    """neosr_conv3x3_pack_many (both image kinds, both modes, many layers per launch) writes exactly what the per-layer
    entry points write; the cached images of a layer stack are refreshed by one batched call after a parameter change"""
    import ctypes as C_

    from neosr_amd import _C
    from neosr_amd.hip import layers, ops

    lib = _C.load()
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 64), (32, 96), (180, 60), (60, 180), (8, 12), (64, 192)] * 6  # 36 layers: two launches per kind
    ws = [torch.randn(co, ci, 3, 3, generator=g).to(DEV) for co, ci in shapes]
    items, want = [], []
    for w in ws:
        for mode in (ops.CONV_FWD, ops.CONV_DGRAD):
            for kind, single in ((0, ops.conv3x3_pack_weights), (1, ops.conv3x3_pack_wino)):
                ref = single(w, mode)
                dst = torch.full_like(ref, float("nan"))
                items.append(_C.PackItem(w=w.data_ptr(), dst=dst.data_ptr(), w_cout=w.shape[0], w_cin=w.shape[1],
                                         mode=mode, kind=kind))
                want.append((ref, dst))
    arr = (_C.PackItem * len(items))(*items)
    _C.check(lib.neosr_conv3x3_pack_many(arr, len(items), _C.stream_ptr()), "pack_many")
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in want)
    # the cache: every registered image of the device is rebuilt by the first miss after a parameter change
    ps = [torch.nn.Parameter(w.clone()) for w in ws[:6]]
    first = [layers.packed_weights(p, ops.CONV_FWD).clone() for p in ps]
    with torch.no_grad():
        for p in ps:
            p.mul_(2.0)  # moves every parameter's _version
    again = layers.packed_weights(ps[0], ops.CONV_FWD)  # one miss ...
    assert all(p.__dict__["_neosr_packs"][(0, ops.CONV_FWD)][0] == layers._pack_key(p) for p in ps)  # ... refreshed all
    assert torch.equal(again, 2.0 * first[0])
    assert all(torch.equal(layers.packed_weights(p, ops.CONV_FWD), 2.0 * f) for p, f in zip(ps, first))
