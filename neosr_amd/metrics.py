"""Validation metrics and image output for `image.validation` (host side; the reference runs them on the CPU too:
neosr/metrics/calculate.py:15-160, neosr/utils/img_util.py:60-129).  numpy only — cv2 is not in this image, so the SSIM
window filter is restated (cv2.filter2D + [5:-5] crop == valid-mode correlation with the 11-tap sigma-1.5 Gaussian) and
"parity unpinned" by a reference run; PSNR and the BT.601 Y conversion are closed formulas."""

from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np
import torch

from neosr_amd.utils.registry import Registry

METRIC_REGISTRY = Registry("metric")


def tensor2img(t: torch.Tensor, rgb2bgr: bool = True, min_max=(0, 1)) -> np.ndarray:
    """(1, C, H, W) | (C, H, W) tensor in [0, 1] -> uint8 HWC (BGR like the reference) or HW (img_util.py:60-129)."""
    t = t.squeeze(0).float().detach().cpu().clamp(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 2:
        img = t.numpy()
    elif t.dim() == 3:
        img = t.numpy().transpose(1, 2, 0)
        if img.shape[2] == 1:
            img = img[..., 0]
        elif rgb2bgr:
            img = img[..., ::-1]
    else:
        raise TypeError(f"Only support 3D or 2D tensors after squeeze, got {t.dim()}D")
    return (img * 255.0).round().astype(np.uint8)


def _to_y(img: np.ndarray) -> np.ndarray:
    """BGR [0, 255] -> Y of YCbCr (ITU-R BT.601), float, no rounding (metric_util.py:35-51)."""
    img = img.astype(np.float32) / 255.0
    if img.ndim == 3 and img.shape[2] == 3:
        img = (np.dot(img, [24.966, 128.553, 65.481]) + 16.0) / 255.0
        img = img[..., None].astype(np.float32)
    return img * 255.0


def _prep(img, img2, crop_border, input_order, test_y_channel):
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    if input_order not in {"HWC", "CHW"}:
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    out = []
    for a in (img, img2):
        if a.ndim == 2:
            a = a[..., None]
        elif input_order == "CHW":
            a = a.transpose(1, 2, 0)
        if crop_border != 0:
            a = a[crop_border:-crop_border, crop_border:-crop_border, ...]
        if test_y_channel:
            a = _to_y(a)
        out.append(a.astype(np.float64))
    return out


@METRIC_REGISTRY.register()
def calculate_psnr(img, img2, crop_border: int = 4, input_order: str = "HWC", test_y_channel: bool = False, **kwargs) -> float:  # noqa: ARG001
    a, b = _prep(img, img2, crop_border, input_order, test_y_channel)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else float(10.0 * np.log10(255.0 * 255.0 / mse))


def _valid_filter(x: np.ndarray, k: np.ndarray) -> np.ndarray:
    """valid-mode separable correlation with the symmetric 1-D kernel k"""
    n = len(k)
    rows = sum(k[i] * x[i: x.shape[0] - n + 1 + i, :] for i in range(n))
    return sum(k[i] * rows[:, i: rows.shape[1] - n + 1 + i] for i in range(n))


def _ssim(a: np.ndarray, b: np.ndarray) -> float:
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5**2))
    k /= k.sum()
    mu1, mu2 = _valid_filter(a, k), _valid_filter(b, k)
    s1 = _valid_filter(a * a, k) - mu1 * mu1
    s2 = _valid_filter(b * b, k) - mu2 * mu2
    s12 = _valid_filter(a * b, k) - mu1 * mu2
    return float((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))).mean())


@METRIC_REGISTRY.register()
def calculate_ssim(img, img2, crop_border: int = 4, input_order: str = "HWC", test_y_channel: bool = False, **kwargs) -> float:  # noqa: ARG001
    a, b = _prep(img, img2, crop_border, input_order, test_y_channel)
    return float(np.mean([_ssim(a[..., i], b[..., i]) for i in range(a.shape[2])]))


def calculate_metric(data: dict, opt: dict) -> float:
    """neosr/metrics/__init__.py: `type` names the registered function, the rest are its kwargs."""
    opt = dict(opt)
    fn = METRIC_REGISTRY.get(opt.pop("type"))
    opt.pop("better", None)
    return fn(**data, **opt)


def imwrite_png(img: np.ndarray, path) -> None:
    """uint8 HW / HWC(BGR) -> PNG file (what cv2.imwrite does in the reference, img_util.py `imwrite`); zlib only."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    if img.ndim == 3:
        img = img[..., ::-1]  # BGR -> RGB
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    color = 0 if img.ndim == 2 else 2
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
                     + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
