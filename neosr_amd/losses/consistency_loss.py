"""``consistency_loss`` — colour / luma consistency (drop-in for neosr/losses/consistency_loss.py:14-192).

clamp -> [GaussianBlur(21, 3), reflect] -> CIE L* luma and Oklab chroma -> chc criterion on both, plus the
cosine-similarity terms that the reference adds only `if cosim < 1e-3`.  Every stage is one HIP kernel with an
exact backward; the `cosim < 1e-3` test is evaluated on the device (`torch.where` on 0-d tensors) so the
iteration has no host sync (the reference's `if` reads the value back, consistency_loss.py:189).
"""

from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from neosr_amd import _C
from neosr_amd.hip.layers import ChcLoss
from neosr_amd.hip.nets import L1LossFunction
from neosr_amd.utils.registry import LOSS_REGISTRY


def _st():
    return _C.stream_ptr()


def _gaussian_taps(kernel_size: int, sigma: float):
    """torchvision `_get_gaussian_kernel1d` (float32)."""
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k = pdf / pdf.sum()
    return (C.c_float * kernel_size)(*[float(v) for v in k])


class _Clamp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_clamp(x.data_ptr(), None, out.data_ptr(), x.numel(), lo, hi, _st()), "neosr_clamp")
        ctx.save_for_backward(x)
        ctx.lim = (lo, hi)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_clamp(x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel(), *ctx.lim, _st()), "neosr_clamp")
        return out, None, None


class _Blur(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, taps, ntaps):
        ctx.cfg = (taps, ntaps)
        return _Blur._run(x, taps, ntaps, 0)

    @staticmethod
    def _run(x, taps, ntaps, adjoint):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, Cc, H, W = x.shape
        out, tmp = torch.empty_like(x), torch.empty_like(x)
        _C.check(lib.neosr_gaussian_blur_reflect(x.data_ptr(), out.data_ptr(), tmp.data_ptr(), taps, ntaps, B * Cc, H, W,
                                                 adjoint, _st()), "neosr_gaussian_blur_reflect")
        return out

    @staticmethod
    def backward(ctx, g):
        return _Blur._run(g, *ctx.cfg, 1), None, None


class _ColorMap(torch.autograd.Function):
    """kind 0: rgb -> luma (B,H,W); kind 1: rgb -> Oklab chroma (B,2,H,W)"""

    @staticmethod
    def forward(ctx, rgb, kind, mul):
        lib = _C.load()
        rgb = _C.require_device(rgb, "rgb").contiguous()
        B, Cc, H, W = rgb.shape
        if Cc != 3:
            raise ValueError(f"Input size must have a shape of (*, 3, H, W). Got {tuple(rgb.shape)}")
        out = torch.empty((B, H, W) if kind == 0 else (B, 2, H, W), device=rgb.device, dtype=torch.float32)
        fn = lib.neosr_rgb_to_luma if kind == 0 else lib.neosr_rgb_to_oklab_chroma
        _C.check(fn(rgb.data_ptr(), None, out.data_ptr(), B, H, W, mul, _st()), "neosr_rgb_to_*")
        ctx.save_for_backward(rgb)
        ctx.cfg = (kind, mul)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        (rgb,) = ctx.saved_tensors
        kind, mul = ctx.cfg
        B, _, H, W = rgb.shape
        g = g.contiguous()
        out = torch.empty_like(rgb)
        fn = lib.neosr_rgb_to_luma if kind == 0 else lib.neosr_rgb_to_oklab_chroma
        _C.check(fn(rgb.data_ptr(), g.data_ptr(), out.data_ptr(), B, H, W, mul, _st()), "neosr_rgb_to_*")
        return out, None, None


class _CosDist(torch.autograd.Function):
    """1 - nn.CosineSimilarity(dim=1, eps)(a, b).mean()"""

    @staticmethod
    def forward(ctx, a, b, eps):
        lib = _C.load()
        a, b = _C.require_device(a, "a").contiguous(), _C.require_device(b, "b").contiguous()
        groups, L = a.shape[0], a.shape[1]
        inner = a.numel() // (groups * L)
        stats = torch.empty(3 * groups * inner, device=a.device, dtype=torch.float32)
        partial = torch.empty(1024, device=a.device, dtype=torch.float32)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        _C.check(lib.neosr_cosine_dist_fwd(a.data_ptr(), b.data_ptr(), stats.data_ptr(), partial.data_ptr(), out.data_ptr(),
                                           groups, L, inner, eps, _st()), "neosr_cosine_dist_fwd")
        ctx.save_for_backward(a, b, stats)
        ctx.cfg = (groups, L, inner, eps)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        a, b, stats = ctx.saved_tensors
        g = g.contiguous().reshape(1)
        da = torch.empty_like(a)
        _C.check(lib.neosr_cosine_dist_bwd(a.data_ptr(), b.data_ptr(), stats.data_ptr(), g.data_ptr(), da.data_ptr(),
                                           *ctx.cfg, _st()), "neosr_cosine_dist_bwd")
        return da, None, None


@LOSS_REGISTRY.register()
class consistency_loss(nn.Module):
    def __init__(self, criterion: str = "chc", blur: bool = True, cosim: bool = True, saturation: float = 1.0,
                 brightness: float = 1.0, loss_weight: float = 1.0) -> None:
        super().__init__()
        if criterion not in ("l1", "chc"):
            raise NotImplementedError(f"{criterion} criterion has not been supported.")
        self.use_blur, self.cosim, self.criterion_type = blur, cosim, criterion
        self.saturation, self.brightness, self.loss_weight = saturation, brightness, loss_weight
        self._taps = _gaussian_taps(21, 3.0)

    def _crit(self, a: Tensor, b: Tensor) -> Tensor:
        if self.criterion_type == "l1":
            return L1LossFunction.apply(a, b, 1.0)
        # chc_loss(loss_lambda=0, clip_min=0, clip_max=1): mean(clamp(sqrt(d^2 + 1e-12), 0, 1))
        return ChcLoss.apply(a, b, 1.0, True, 0.0, 1.0, 1.0)

    def forward(self, net_output: Tensor, gt: Tensor) -> Tensor:
        x = _Clamp.apply(net_output, 1 / 255, 1.0)
        with torch.no_grad():
            t = _Clamp.apply(gt, 1 / 255, 1.0)
        if self.use_blur:
            xb = _Clamp.apply(_Blur.apply(x, self._taps, 21), 0.0, 1.0)
            with torch.no_grad():
                tb = _Clamp.apply(_Blur.apply(t, self._taps, 21), 0.0, 1.0)
        else:
            xb, tb = x, t
        in_luma = _ColorMap.apply(xb, 0, 1.0)
        in_chroma = _ColorMap.apply(x, 1, 1.0)
        with torch.no_grad():
            tg_luma = _ColorMap.apply(tb, 0, float(self.brightness))
            tg_chroma = _ColorMap.apply(t, 1, float(self.saturation))
        loss = self._crit(in_luma, tg_luma) + self._crit(in_chroma, tg_chroma)
        if self.cosim:
            cos = 0.5 * _CosDist.apply(in_chroma, tg_chroma, 1e-20) + 0.5 * _CosDist.apply(in_luma, tg_luma, 1e-20)
            loss = torch.where(cos < 1e-3, loss + cos, loss)  # consistency_loss.py:186-190, decided on the device
        return loss * self.loss_weight
