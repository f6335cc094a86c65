"""Python fronts of the degradation-bank entry points (`neosr_filter2d`, `neosr_resize`,
`neosr_gaussian_noise`, `neosr_poisson_*`, `neosr_diffjpeg`, …).  Tensors are planar (B,C,H,W) fp32
HBM handles; all arithmetic is in `csrc/degrade.hip`."""

from __future__ import annotations

import math

import torch

from neosr_amd import _C

RESIZE_MODES = {"area": 0, "bilinear": 1, "bicubic": 2}


def _img(t: torch.Tensor, name: str = "image") -> torch.Tensor:
    _C.require_device(t, name)
    if t.dim() != 4:
        raise _C.NeosrAmdError(f"{name} must be (B,C,H,W), got {tuple(t.shape)}")
    return t.contiguous()


def filter2d(img: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """`filter2D(img, kernel)`; kernel (B,k,k) or (1,k,k)."""
    lib = _C.load()
    img = _img(img)
    kernel = _C.require_device(kernel, "kernel").contiguous()
    B, C_, H, W = img.shape
    k = kernel.shape[-1]
    if k % 2 != 1:
        raise ValueError("Wrong kernel size")  # same error as the reference
    if kernel.shape[0] not in (1, B):
        raise _C.NeosrAmdError("kernel batch must be 1 or B")
    out = torch.empty_like(img)
    _C.check(lib.neosr_filter2d(img.data_ptr(), kernel.data_ptr(), out.data_ptr(), B, C_, H, W, k,
                                int(kernel.shape[0] != 1), _C.stream_ptr()), "neosr_filter2d")
    return out


def resize(img: torch.Tensor, *, size: tuple[int, int] | None = None,
           scale_factor: float | None = None, mode: str = "bilinear") -> torch.Tensor:
    """`F.interpolate(img, size=|scale_factor=, mode=area|bilinear|bicubic)` (no antialias)."""
    lib = _C.load()
    img = _img(img)
    B, C_, H, W = img.shape
    if (size is None) == (scale_factor is None):
        raise ValueError("exactly one of size / scale_factor")
    if scale_factor is not None:
        Ho, Wo = int(math.floor(H * scale_factor)), int(math.floor(W * scale_factor))
        rs_h = rs_w = 1.0 / scale_factor  # ATen uses the given factor, not in/out
    else:
        Ho, Wo = int(size[0]), int(size[1])
        rs_h, rs_w = H / Ho, W / Wo
    out = torch.empty(B, C_, Ho, Wo, device=img.device, dtype=torch.float32)
    _C.check(lib.neosr_resize(img.data_ptr(), out.data_ptr(), B * C_, H, W, Ho, Wo,
                              RESIZE_MODES[mode], float(rs_h), float(rs_w), _C.stream_ptr()),
             "neosr_resize")
    return out


def gaussian_noise(img, noise, noise_gray, sigma, gray) -> torch.Tensor:
    """clamp(img + noise_c*(1-g) + noise_gray*g, 0, 1) with per-sample sigma/255 scaling."""
    lib = _C.load()
    img = _img(img)
    B, C_, H, W = img.shape
    out = torch.empty_like(img)
    _C.check(lib.neosr_gaussian_noise(img.data_ptr(), noise.contiguous().data_ptr(),
                                      None if noise_gray is None else noise_gray.contiguous().data_ptr(),
                                      sigma.contiguous().data_ptr(),
                                      None if gray is None else gray.contiguous().data_ptr(),
                                      out.data_ptr(), B, C_, H, W, _C.stream_ptr()),
             "neosr_gaussian_noise")
    return out


def poisson_rate(img: torch.Tensor, gray: bool) -> tuple[torch.Tensor, torch.Tensor]:
    """(rate, vals): rate = q(img or gray(img)) * vals[b], vals = 2^ceil(log2(#distinct 8-bit levels))."""
    lib = _C.load()
    img = _img(img)
    B, C_, H, W = img.shape
    if C_ != 3:
        raise _C.NeosrAmdError("poisson noise expects RGB images")
    rate = torch.empty(B, 1 if gray else 3, H, W, device=img.device, dtype=torch.float32)
    vals = torch.empty(B, device=img.device, dtype=torch.float32)
    ws = torch.empty(B * 8, device=img.device, dtype=torch.int32)
    _C.check(lib.neosr_poisson_rate(img.data_ptr(), ws.data_ptr(), vals.data_ptr(), rate.data_ptr(),
                                    B, H, W, int(gray), _C.stream_ptr()), "neosr_poisson_rate")
    return rate, vals


def poisson_sample(rate: torch.Tensor, seed: int, offset: int) -> torch.Tensor:
    """P ~ Poisson(rate), a pure function of (seed, offset, rate): `neosr_poisson_sample`."""
    lib = _C.load()
    rate = _C.require_device(rate, "rate").contiguous()
    out = torch.empty_like(rate)
    _C.check(lib.neosr_poisson_sample(rate.data_ptr(), out.data_ptr(), rate.numel(), seed & (2**64 - 1),
                                      offset & (2**64 - 1), _C.stream_ptr()), "neosr_poisson_sample")
    return out


def normal_sample(shape, seed: int, offset: int, device="cuda") -> torch.Tensor:
    """N(0, 1) field, a pure function of (seed, offset): `neosr_normal_sample` (Philox4x32-10 + Box-Muller)."""
    lib = _C.load()
    out = torch.empty(*shape, device=device, dtype=torch.float32)
    _C.check(lib.neosr_normal_sample(out.data_ptr(), out.numel(), seed & (2**64 - 1), offset & (2**64 - 1),
                                     _C.stream_ptr()), "neosr_normal_sample")
    return out


def blur_kernels(params: torch.Tensor) -> torch.Tensor:
    """(n, 8) float64 parameter table on the device -> (n, 21, 21) float32 kernels (`neosr_blur_kernels`)."""
    lib = _C.load()
    if not params.is_cuda or params.dtype != torch.float64 or params.dim() != 2 or params.shape[1] != 8:
        raise _C.NeosrAmdError("blur_kernels: params must be a (n, 8) float64 tensor on a HIP device")
    params = params.contiguous()
    out = torch.empty(params.shape[0], 21, 21, device=params.device, dtype=torch.float32)
    _C.check(lib.neosr_blur_kernels(params.data_ptr(), params.shape[0], out.data_ptr(), _C.stream_ptr()),
             "neosr_blur_kernels")
    return out


def poisson_noise(img, P, vals, P_gray, vals_gray, scale, gray) -> torch.Tensor:
    lib = _C.load()
    img = _img(img)
    B, _, H, W = img.shape
    out = torch.empty_like(img)
    opt = lambda t: None if t is None else t.contiguous().data_ptr()  # noqa: E731
    _C.check(lib.neosr_poisson_noise(img.data_ptr(), P.contiguous().data_ptr(), vals.data_ptr(),
                                     opt(P_gray), opt(vals_gray), scale.contiguous().data_ptr(),
                                     opt(gray), out.data_ptr(), B, H, W, _C.stream_ptr()),
             "neosr_poisson_noise")
    return out


def diffjpeg(img: torch.Tensor, quality: torch.Tensor) -> torch.Tensor:
    """`DiffJPEG(differentiable=False)(img, quality=quality)`; quality (B) in (0,100]; not mutated."""
    lib = _C.load()
    img = _img(img)
    B, C_, H, W = img.shape
    if C_ != 3:
        raise _C.NeosrAmdError("DiffJPEG expects RGB images")
    quality = _C.require_device(quality, "quality").contiguous()
    out = torch.empty_like(img)
    _C.check(lib.neosr_diffjpeg(img.data_ptr(), quality.data_ptr(), out.data_ptr(), B, H, W,
                                _C.stream_ptr()), "neosr_diffjpeg")
    return out


def quantize_u8(img: torch.Tensor) -> torch.Tensor:
    lib = _C.load()
    img = _C.require_device(img, "image").contiguous()
    out = torch.empty_like(img)
    _C.check(lib.neosr_quantize_u8(img.data_ptr(), out.data_ptr(), img.numel(), _C.stream_ptr()),
             "neosr_quantize_u8")
    return out


def clamp01(img: torch.Tensor) -> torch.Tensor:
    lib = _C.load()
    img = _C.require_device(img, "image").contiguous()
    out = torch.empty_like(img)
    _C.check(lib.neosr_clamp01(img.data_ptr(), out.data_ptr(), img.numel(), _C.stream_ptr()),
             "neosr_clamp01")
    return out


def crop(img: torch.Tensor, top: int, left: int, h: int, w: int) -> torch.Tensor:
    lib = _C.load()
    img = _img(img)
    B, C_, H, W = img.shape
    out = torch.empty(B, C_, h, w, device=img.device, dtype=torch.float32)
    _C.check(lib.neosr_crop(img.data_ptr(), out.data_ptr(), B * C_, H, W, top, left, h, w,
                            _C.stream_ptr()), "neosr_crop")
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src[idx] along dim 0 (idx int64 on the device)."""
    lib = _C.load()
    src = _C.require_device(src, "src").contiguous()
    if idx.dtype != torch.int64 or not idx.is_cuda:
        raise _C.NeosrAmdError("idx must be a device int64 tensor")
    out = torch.empty((idx.numel(), *src.shape[1:]), device=src.device, dtype=torch.float32)
    row = src[0].numel()
    _C.check(lib.neosr_gather_rows(src.data_ptr(), idx.contiguous().data_ptr(), out.data_ptr(),
                                   idx.numel(), row, _C.stream_ptr()), "neosr_gather_rows")
    return out
