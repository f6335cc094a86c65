"""GPU: the RRDB trunk as one launch per RRDB (conv3x3_wino4_chain_kernel, neosr_set_conv_chain) against the same trunk
as one launch per convolution.  The chain kernel runs the SAME arithmetic per layer (transforms, MFMA order, epilogue),
so outputs and every parameter gradient must agree BIT FOR BIT wherever both paths pick the same workgroup shapes
(batch 16: the fill estimate picks the 64-channel shape for conv5 / the RDB-input gradient, as the chain always does);
any difference is a tile-to-tile hand-off bug (stale read, missed flag).  Repeated runs check the flag protocol under
whatever timing the box produces; the status word must stay 0 (no flag wait hit its bound)."""

from __future__ import annotations

import pytest
import torch

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def lib():
    from neosr_amd import _C

    return _C.load()


def _net(num_block):
    from neosr_amd.archs import build_network

    torch.manual_seed(7)
    return build_network({"type": "esrgan", "scale": 4, "num_block": num_block}).to(DEV).train()


def _fwd_bwd(net, x, gy):
    net.zero_grad(set_to_none=True)
    y = net(x)
    y.backward(gy)
    torch.cuda.synchronize()
    return y.detach().clone(), [p.grad.detach().clone() for p in net.parameters()]


@pytest.mark.parametrize("batch,hw,blocks", [(16, 64, 3), (4, 64, 2), (3, 48, 2), (5, 80, 1), (20, 64, 1)])
def test_chain_is_bit_identical_to_per_layer_launches(lib, batch, hw, blocks):
    net = _net(blocks)
    g = torch.Generator().manual_seed(batch * 100 + hw)
    x = torch.rand(batch, 3, hw, hw, generator=g).to(DEV)
    gy = (torch.randn(batch, 3, 4 * hw, 4 * hw, generator=g) * 1e-3).to(DEV)
    prev_n64 = lib.neosr_set_wino4_n64(1)   # both paths: 64-channel workgroups for the 64-channel layers
    prev_wg = lib.neosr_set_wgrad_rrdb(0)   # weight gradients: one launch per RDB on both paths (same pixel splits)
    prev = lib.neosr_set_conv_chain(0)
    try:
        y0, g0 = _fwd_bwd(net, x, gy)
        lib.neosr_set_conv_chain(1)
        for rep in range(3):
            y1, g1 = _fwd_bwd(net, x, gy)
            assert torch.equal(y0, y1), (rep, rel_err(y1, y0))
            bad = [i for i, (a, b) in enumerate(zip(g0, g1)) if not torch.equal(a, b)]
            assert not bad, (rep, bad[:8], max(rel_err(g1[i], g0[i]) for i in bad))
        # the default: the fifteen weight gradients of an RRDB in one launch = another split of the pixel range, i.e.
        # the same sums in another order; run-to-run identical
        lib.neosr_set_wgrad_rrdb(1)
        y2, g2 = _fwd_bwd(net, x, gy)
        y3, g3 = _fwd_bwd(net, x, gy)
        assert torch.equal(y0, y2) and torch.equal(y2, y3)
        assert all(torch.equal(a, b) for a, b in zip(g2, g3))
        worst = max(rel_err(a, b) for a, b in zip(g2, g0))
        assert worst < 5e-6, worst
        assert lib.neosr_conv_chain_status() == 0
    finally:
        lib.neosr_set_conv_chain(prev)
        lib.neosr_set_wgrad_rrdb(prev_wg)
        lib.neosr_set_wino4_n64(prev_n64)


def test_chain_full_size_repeated(lib):
    """BASELINE configs[1] geometry (23 RRDBs, batch 16, one tile per CU): the chain against the per-layer launches, and
    twenty chain runs against each other (a hand-off race would show up as a run-to-run difference)."""
    net = _net(23)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(16, 3, 64, 64, generator=g).to(DEV)
    gy = (torch.randn(16, 3, 256, 256, generator=g) * 1e-3).to(DEV)
    prev = lib.neosr_set_conv_chain(0)
    prev_wg = lib.neosr_set_wgrad_rrdb(0)
    try:
        y0, g0 = _fwd_bwd(net, x, gy)
        lib.neosr_set_conv_chain(1)
        for rep in range(20):
            y1, g1 = _fwd_bwd(net, x, gy)
            assert torch.equal(y0, y1), (rep, rel_err(y1, y0))
            assert all(torch.equal(a, b) for a, b in zip(g0, g1)), rep
        assert lib.neosr_conv_chain_status() == 0
    finally:
        lib.neosr_set_conv_chain(prev)
        lib.neosr_set_wgrad_rrdb(prev_wg)


def test_chain_inference_ring(lib):
    """eval(): the activation ring of four concat buffers is reused across the RDBs of a chain launch"""
    net = _net(4).eval()
    x = torch.rand(8, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(DEV)
    prev = lib.neosr_set_conv_chain(0)
    prev_n64 = lib.neosr_set_wino4_n64(1)
    try:
        with torch.no_grad():
            y0 = net(x).clone()
            lib.neosr_set_conv_chain(1)
            for _ in range(3):
                assert torch.equal(net(x), y0)
        assert lib.neosr_conv_chain_status() == 0
    finally:
        lib.neosr_set_conv_chain(prev)
        lib.neosr_set_wino4_n64(prev_n64)


@pytest.mark.parametrize("B,H,W,K,N,opts", [
    (8, 128, 128, 64, 64, "plain"),
    (8, 128, 128, 128, 32, "lrelu_res"),
    (32, 64, 64, 64, 64, "mask"),
    (4, 256, 256, 64, 64, "dgrad"),
])
def test_sample_strips_of_one_convolution_are_bit_identical(lib, B, H, W, K, N, opts):
    """A convolution with at least twice as many pixel tiles as CUs runs as chain launches over strips of samples
    (conv_wino4_chain.hip, launch_wino4_strips: independent layers, no flags); the same launch through the one-layer kernel
    (neosr_set_conv_chain(0)) must give the same bits."""
    from neosr_amd.hip import ops

    g = torch.Generator().manual_seed(B + H + K + N)
    x = torch.randn(B, H, W, K + 8, generator=g).to(DEV)
    mode = ops.CONV_DGRAD if opts == "dgrad" else ops.CONV_FWD
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.05).to(DEV) if mode == ops.CONV_FWD else (torch.randn(K, N, 3, 3, generator=g) * 0.05).to(DEV)
    kw = dict(mode=mode, k_in=K, w_pack=ops.conv3x3_pack_weights(w, mode), w_wino4=ops.conv3x3_pack_wino4(w, mode))
    if opts == "lrelu_res":
        kw.update(bias=torch.randn(N, generator=g).to(DEV), act=ops.ACT_LRELU, slope=0.2, alpha=0.2,
                  res1=torch.randn(B, H, W, N, generator=g).to(DEV))
    if opts == "mask":
        kw.update(out_mask=torch.randn(B, H, W, N, generator=g).to(DEV), out_mask_slope=0.2)
    prev_n64 = lib.neosr_set_wino4_n64(1)
    prev = lib.neosr_set_conv_chain(0)
    try:
        y0 = ops.conv3x3(x, w, **kw).clone()
        lib.neosr_set_conv_chain(1)
        for _ in range(3):
            y1 = ops.conv3x3(x, w, **kw)
            assert torch.equal(y0, y1), rel_err(y1, y0)
        assert lib.neosr_conv_chain_status() == 0
    finally:
        lib.neosr_set_conv_chain(prev)
        lib.neosr_set_wino4_n64(prev_n64)
