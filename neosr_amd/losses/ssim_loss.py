"""``mssim_loss`` — multi-scale SSIM loss (drop-in for neosr/losses/ssim_loss.py:66-163) on HIP kernels.

Five scales; per scale ONE kernel applies the separable 11-tap Gaussian window to x, y, x^2, y^2 and xy
from an LDS halo tile and reduces the cs / ssim maps (the reference runs 5 depthwise 11x11 convolutions
and ~15 elementwise passes per scale); 2x2 average pools in between; the scalar combination
`1 - prod cs_i^w_i * ssim_4^w_4` and the per-scale gradient scalars are computed on the device (no host
sync).  Backward filters three derivative maps per scale with the same (self-adjoint) window and chains the
scales through the pooling adjoint.
"""

from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from neosr_amd import _C
from neosr_amd.utils.registry import LOSS_REGISTRY

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _window(window_size: int, sigma: float):
    """GaussianFilter2D._get_gaussian_window1d (ssim_loss.py:44-49), float32 like the reference buffer."""
    x = torch.arange(-(window_size // 2), window_size // 2 + 1)
    w = torch.exp(-0.5 * x**2 / (sigma * sigma))
    w /= w.sum()
    return (C.c_float * window_size)(*[float(v) for v in w.float()])


class _Msssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, win, c1, c2, loss_weight):
        lib, st = _C.load(), _C.stream_ptr()
        x = _C.require_device(x, "x").contiguous()
        y = _C.require_device(y, "y").contiguous()
        B, Cc, H, W = x.shape
        P = B * Cc
        want_grad = ctx.needs_input_grad[0]
        xs, ys, dmaps, parts = [x], [y], [], []
        d = _C.MsssimDesc()
        for s in range(5):
            h, w = H >> s, W >> s
            if s > 0:
                if (H >> (s - 1)) % 2 or (W >> (s - 1)) % 2:
                    raise _C.NeosrAmdError("mssim_loss: spatial size must be divisible by 16 (no odd-size pooling path)")
                for src, dst in ((xs, xs), (ys, ys)):
                    nxt = torch.empty(B, Cc, h, w, device=x.device, dtype=torch.float32)
                    _C.check(lib.neosr_avgpool2_planes(src[-1].data_ptr(), nxt.data_ptr(), P, h * 2, w * 2, st),
                             "neosr_avgpool2_planes")
                    dst.append(nxt)
            nblk = lib.neosr_ssim_tiles(P, h, w)
            part = torch.empty(nblk * 2, device=x.device, dtype=torch.float32)
            dm = torch.empty(3 * P * h * w, device=x.device, dtype=torch.float32) if want_grad else None
            _C.check(lib.neosr_ssim_fwd(xs[s].data_ptr(), ys[s].data_ptr(), win, None if dm is None else dm.data_ptr(),
                                        part.data_ptr(), P, h, w, c1, c2, int(s == 4), st), "neosr_ssim_fwd")
            parts.append(part)
            dmaps.append(dm)
            d.partial[s], d.nblk[s], d.npix[s], d.weights[s] = part.data_ptr(), nblk, P * h * w, _WEIGHTS[s]
        loss = torch.empty(1, device=x.device, dtype=torch.float32)
        gscal = torch.empty(5, device=x.device, dtype=torch.float32)
        d.loss_weight, d.nscales, d.loss, d.gscal = loss_weight, 5, loss.data_ptr(), gscal.data_ptr()
        _C.check(lib.neosr_msssim_finalize(C.byref(d), st), "neosr_msssim_finalize")
        if want_grad:
            ctx.save_for_backward(gscal, *xs, *ys, *dmaps)
            ctx.meta = (win, P, H, W)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib, st = _C.load(), _C.stream_ptr()
        gscal, *rest = ctx.saved_tensors
        xs, ys, dmaps = rest[:5], rest[5:10], rest[10:]
        win, P, H, W = ctx.meta
        g = g.contiguous().reshape(1)
        coarse = None
        for s in range(4, -1, -1):
            h, w = H >> s, W >> s
            dx = torch.empty_like(xs[s])
            _C.check(lib.neosr_ssim_bwd(dmaps[s].data_ptr(), xs[s].data_ptr(), ys[s].data_ptr(), win,
                                        gscal[s:].data_ptr(), g.data_ptr(), None if coarse is None else coarse.data_ptr(),
                                        dx.data_ptr(), P, h, w, st), "neosr_ssim_bwd")
            coarse = dx
        return coarse, None, None, None, None, None


@LOSS_REGISTRY.register()
class mssim_loss(nn.Module):
    def __init__(self, window_size: int = 11, in_channels: int = 3, sigma: float = 1.5, K1: float = 0.01,
                 K2: float = 0.03, L: int = 1, padding: int | None = None, loss_weight: float = 1.0) -> None:
        super().__init__()
        if window_size != 11 or (padding is not None and padding != 5):
            raise NotImplementedError("mssim_loss: the HIP kernel implements window_size 11 with padding 5 (the defaults)")
        self.window_size, self.in_channels = window_size, in_channels
        self.C1, self.C2, self.loss_weight = (K1 * L) ** 2, (K2 * L) ** 2, loss_weight
        self._win = _window(window_size, sigma)

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        assert x.shape == y.shape, f"x: {x.shape} and y: {y.shape} must be the same"
        assert x.ndim == y.ndim == 4, f"x: {x.ndim} and y: {y.ndim} must be 4"
        return _Msssim.apply(x, y, self._win, float(self.C1), float(self.C2), float(self.loss_weight))
