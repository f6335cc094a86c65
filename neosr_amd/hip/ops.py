"""Thin, typed Python fronts for the primitive entry points of the C ABI.

Tensors are torch tensors used purely as HBM handles (``data_ptr()``); all arithmetic happens in
the HIP kernels.  Activations are channels-last ``(B, H, W, C)`` contiguous tensors, optionally a
channel *slice* of a wider buffer (``cs`` = channel stride = ``buf.shape[-1]``).
"""

from __future__ import annotations

import ctypes as C

import torch

from neosr_amd import _C
from neosr_amd._C import ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU, CONV_DGRAD, CONV_FWD  # noqa: F401


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _cs(t: torch.Tensor) -> int:
    """channel stride of a channels-last tensor or channel-slice view"""
    assert t.dim() == 4 and t.stride(3) == 1, "expected (B,H,W,C) with unit channel stride"
    cs = t.stride(2)
    assert t.stride(1) == t.shape[2] * cs and t.stride(0) == t.shape[1] * t.shape[2] * cs
    return cs


def conv3x3(x, w, bias=None, *, out=None, n_out=None, mode=CONV_FWD, ups=False, act=ACT_NONE,
            slope=0.0, prelu=None, alpha=1.0, res1=None, res1_nch=None, alpha2=1.0, res2=None,
            res2_nch=None, accumulate=False, in_mask=None, mask_slope=1.0, mask_slopes=None,
            in_prelu=None, k_in=None, w_pack=None, out_mask=None, out_mask_slope=1.0, s2d_c=0, w_wino=None,
            w_wino4=None, out2=None, out_mask_gelu=False):
    """out = epilogue(conv3x3(x', w)).  See ``neosr_conv3x3`` in include/neosr_amd.h."""
    lib = _C.load()
    _C.require_device(x, "x")
    _C.require_device(w, "w")
    B, Hin, Win, Cx = x.shape
    H, W = (Hin * 2, Win * 2) if ups else (Hin, Win)
    w_cout, w_cin = w.shape[0], w.shape[1]
    assert w.is_contiguous() and w.shape[2:] == (3, 3)
    if mode == CONV_FWD:
        K = w_cin if k_in is None else k_in
        N = w_cout if n_out is None else n_out
    else:
        K = w_cout if k_in is None else k_in
        N = w_cin if n_out is None else n_out
    assert Cx >= K
    if out is None:
        out = torch.empty(B, H, W, N, device=x.device, dtype=torch.float32)
    d = _C.ConvDesc()
    d.in_ = x.data_ptr()
    d.in_cs = _cs(x)
    if in_mask is not None:
        d.in_mask = in_mask.data_ptr()
        d.mask_cs = _cs(in_mask)
    d.mask_slopes = _ptr(mask_slopes)
    d.in_prelu = _ptr(in_prelu)
    d.w = w.data_ptr()
    d.bias = _ptr(bias)
    d.prelu = _ptr(prelu)
    if res1 is not None:
        d.res1 = res1.data_ptr()
        d.res1_cs = _cs(res1)
        d.res1_nch = N if res1_nch is None else res1_nch
    if res2 is not None:
        d.res2 = res2.data_ptr()
        d.res2_cs = _cs(res2)
        d.res2_nch = N if res2_nch is None else res2_nch
    d.out = out.data_ptr()
    d.out_cs = _cs(out)
    d.B, d.H, d.W, d.K, d.N = B, H, W, K, N
    d.w_cout, d.w_cin = w_cout, w_cin
    d.mode, d.ups, d.act, d.accumulate = mode, int(ups), act, int(accumulate)
    d.mask_slope, d.slope, d.alpha, d.alpha2 = mask_slope, slope, alpha, alpha2
    d.w_pack = _ptr(w_pack)
    d.w_wino = _ptr(w_wino)
    d.w_wino4 = _ptr(w_wino4)
    d.s2d_c = int(s2d_c)
    if out_mask is not None:
        d.out_mask = out_mask.data_ptr()
        d.out_mask_cs = _cs(out_mask)
        d.out_mask_slope = out_mask_slope
        d.out_mask_gelu = int(bool(out_mask_gelu))
    if out2 is not None:   # (F(4x4,3x3) kernel only: conv + bias before the activation)
        d.out2 = out2.data_ptr()
        d.out2_cs = _cs(out2)
    _C.check(lib.neosr_conv3x3(C.byref(d), _C.stream_ptr()), "neosr_conv3x3")
    return out


def conv3x3_pack_weights(w, mode=CONV_FWD):
    """Packed image of ``w`` for the direct-to-LDS kernel (``neosr_conv3x3_pack_weights``)."""
    lib = _C.load()
    _C.require_device(w, "w")
    assert w.is_contiguous() and w.shape[2:] == (3, 3)
    cout, cin = w.shape[0], w.shape[1]
    N, K = (cout, cin) if mode == CONV_FWD else (cin, cout)
    nbytes = lib.neosr_conv3x3_pack_bytes(N, K)
    dst = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    _C.check(lib.neosr_conv3x3_pack_weights(w.data_ptr(), cout, cin, mode, dst.data_ptr(), _C.stream_ptr()),
             "neosr_conv3x3_pack_weights")
    return dst


def conv3x3_pack_wino(w, mode=CONV_FWD):
    """Winograd F(2x2,3x3) image of ``w`` (``neosr_conv3x3_pack_wino``), for `conv3x3(..., w_pack=, w_wino=)`."""
    lib = _C.load()
    _C.require_device(w, "w")
    assert w.is_contiguous() and w.shape[2:] == (3, 3)
    cout, cin = w.shape[0], w.shape[1]
    N, K = (cout, cin) if mode == CONV_FWD else (cin, cout)
    nbytes = lib.neosr_conv3x3_pack_wino_bytes(N, K)
    dst = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    _C.check(lib.neosr_conv3x3_pack_wino(w.data_ptr(), cout, cin, mode, dst.data_ptr(), _C.stream_ptr()),
             "neosr_conv3x3_pack_wino")
    return dst


def conv3x3_pack_wino4(w, mode=CONV_FWD):
    """Winograd F(4x4,3x3) image of ``w`` (``neosr_conv3x3_pack_wino4``), for `conv3x3(..., w_pack=, w_wino4=)`."""
    lib = _C.load()
    _C.require_device(w, "w")
    assert w.is_contiguous() and w.shape[2:] == (3, 3)
    cout, cin = w.shape[0], w.shape[1]
    N, K = (cout, cin) if mode == CONV_FWD else (cin, cout)
    nbytes = lib.neosr_conv3x3_pack_wino4_bytes(N, K)
    dst = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    _C.check(lib.neosr_conv3x3_pack_wino4(w.data_ptr(), cout, cin, mode, dst.data_ptr(), _C.stream_ptr()),
             "neosr_conv3x3_pack_wino4")
    return dst


def conv3x3_wgrad(x, g, n_out, k_in, *, ups=False, g_mask=None, mask_slope=1.0, mask_slopes=None,
                  in_prelu=None, scale=1.0, dw=None, db=None, want_bias=True, accumulate=False, s2d_c=0):
    """(dw, db) of the 3x3 conv.  See ``neosr_conv3x3_wgrad``."""
    lib = _C.load()
    _C.require_device(x, "x")
    _C.require_device(g, "g")
    B, H, W, _ = g.shape
    if dw is None:
        dw = torch.empty(n_out, k_in, 3, 3, device=x.device, dtype=torch.float32)
    if db is None and want_bias:
        db = torch.empty(n_out, device=x.device, dtype=torch.float32)
    nbytes = lib.neosr_conv3x3_wgrad_workspace_bytes(B, H, W, k_in, n_out)
    ws = torch.empty(nbytes // 4 + 64, device=x.device, dtype=torch.float32)
    d = _C.WgradDesc()
    d.in_ = x.data_ptr()
    d.in_cs = _cs(x)
    d.in_prelu = _ptr(in_prelu)
    d.g = g.data_ptr()
    d.g_cs = _cs(g)
    if g_mask is not None:
        d.g_mask = g_mask.data_ptr()
        d.mask_cs = _cs(g_mask)
    d.mask_slopes = _ptr(mask_slopes)
    d.dw = dw.data_ptr()
    d.db = _ptr(db)
    d.workspace = ws.data_ptr()
    d.B, d.H, d.W, d.K, d.N = B, H, W, k_in, n_out
    d.ups, d.accumulate = int(ups), int(accumulate)
    d.mask_slope, d.scale = mask_slope, scale
    d.s2d_c = int(s2d_c)
    _C.check(lib.neosr_conv3x3_wgrad(C.byref(d), _C.stream_ptr()), "neosr_conv3x3_wgrad")
    return dw, db


def nchw_to_nhwc(x: torch.Tensor, cs: int | None = None) -> torch.Tensor:
    lib = _C.load()
    _C.require_device(x, "x")
    B, Cc, H, W = x.shape
    cs = cs or Cc
    out = torch.zeros(B, H, W, cs, device=x.device, dtype=torch.float32)
    _C.check(lib.neosr_nchw_to_nhwc(x.contiguous().data_ptr(), out.data_ptr(), B, Cc, H, W, cs,
                                    _C.stream_ptr()), "neosr_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, c: int | None = None) -> torch.Tensor:
    lib = _C.load()
    _C.require_device(x, "x")
    B, H, W, _ = x.shape
    c = c or x.shape[3]
    out = torch.empty(B, c, H, W, device=x.device, dtype=torch.float32)
    _C.check(lib.neosr_nhwc_to_nchw(x.data_ptr(), out.data_ptr(), B, c, H, W, _cs(x),
                                    _C.stream_ptr()), "neosr_nhwc_to_nchw")
    return out


def pool2x2_sum(x: torch.Tensor) -> torch.Tensor:
    lib = _C.load()
    _C.require_device(x, "x")
    B, H2, W2, Cc = x.shape
    out = torch.empty(B, H2 // 2, W2 // 2, Cc, device=x.device, dtype=torch.float32)
    _C.check(lib.neosr_pool2x2_sum(x.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, Cc, _cs(x),
                                   Cc, 0, _C.stream_ptr()), "neosr_pool2x2_sum")
    return out


def pixel_shuffle(x_nhwc: torch.Tensor, r: int, base: torch.Tensor | None = None) -> torch.Tensor:
    """PixelShuffle(r) of a channels-last tensor into planar NCHW (+ optional nearest-upsampled base)."""
    lib = _C.load()
    _C.require_device(x_nhwc, "x")
    B, H, W, Crr = x_nhwc.shape
    Cc = Crr // (r * r)
    out = torch.empty(B, Cc, H * r, W * r, device=x_nhwc.device, dtype=torch.float32)
    _C.check(lib.neosr_pixel_shuffle_nhwc_to_nchw(x_nhwc.data_ptr(), _ptr(base), out.data_ptr(), B,
                                                  Cc, H, W, r, _cs(x_nhwc), _C.stream_ptr()),
             "neosr_pixel_shuffle_nhwc_to_nchw")
    return out


def pixel_unshuffle(g_nchw: torch.Tensor, r: int) -> torch.Tensor:
    lib = _C.load()
    _C.require_device(g_nchw, "g")
    B, Cc, Ho, Wo = g_nchw.shape
    H, W = Ho // r, Wo // r
    out = torch.empty(B, H, W, Cc * r * r, device=g_nchw.device, dtype=torch.float32)
    _C.check(lib.neosr_pixel_unshuffle_nchw_to_nhwc(g_nchw.contiguous().data_ptr(), out.data_ptr(),
                                                    B, Cc, H, W, r, Cc * r * r, _C.stream_ptr()),
             "neosr_pixel_unshuffle_nchw_to_nhwc")
    return out
