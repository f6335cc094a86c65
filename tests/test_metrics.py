"""CPU: validation metrics and image output (neosr_amd/metrics.py) — PSNR / SSIM against independent float64
restatements (scipy), BT.601 Y conversion, tensor2img rounding / channel order, PNG writer round trip (decoded by
hand with zlib)."""

from __future__ import annotations

import struct
import zlib

import numpy as np
import torch
from scipy import ndimage

from neosr_amd import metrics as M


def _imgs(seed=0, h=40, w=52):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(h, w, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-12, 13, size=a.shape), 0, 255).astype(np.uint8)
    return a, b


def test_psnr_and_ssim_against_independent_restatement():
    a, b = _imgs()
    x, y = a[4:-4, 4:-4].astype(np.float64), b[4:-4, 4:-4].astype(np.float64)
    assert abs(M.calculate_psnr(a, b, crop_border=4) - 10 * np.log10(255.0**2 / np.mean((x - y) ** 2))) < 1e-12
    assert M.calculate_psnr(a, a) == float("inf")
    # SSIM: scipy correlate with the outer-product window, cropped to the valid region (cv2.filter2D(...)[5:-5, 5:-5])
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5**2))
    k /= k.sum()
    win = np.outer(k, k)
    f = lambda t: ndimage.correlate(t, win, mode="mirror")[5:-5, 5:-5]  # noqa: E731
    ref = []
    for c in range(3):
        p, q = x[..., c], y[..., c]
        mu1, mu2 = f(p), f(q)
        s1, s2, s12 = f(p * p) - mu1**2, f(q * q) - mu2**2, f(p * q) - mu1 * mu2
        c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        ref.append((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1**2 + mu2**2 + c1) * (s1 + s2 + c2))).mean())
    assert abs(M.calculate_ssim(a, b, crop_border=4) - np.mean(ref)) < 1e-10
    assert abs(M.calculate_ssim(a, a) - 1.0) < 1e-12
    # Y channel: BGR order, 16 + (24.966 B + 128.553 G + 65.481 R) / 255
    yv = M.calculate_psnr(a, b, crop_border=0, test_y_channel=True)
    ya = 16.0 + (a.astype(np.float32) / 255.0) @ np.array([24.966, 128.553, 65.481], dtype=np.float64)
    yb = 16.0 + (b.astype(np.float32) / 255.0) @ np.array([24.966, 128.553, 65.481], dtype=np.float64)
    assert abs(yv - 10 * np.log10(255.0**2 / np.mean((ya - yb) ** 2))) < 1e-4
    assert abs(M.calculate_metric({"img": a, "img2": b}, {"type": "calculate_psnr", "crop_border": 4, "better": "higher"})
               - M.calculate_psnr(a, b, crop_border=4)) == 0


def test_tensor2img_and_png_round_trip(tmp_path):
    t = torch.tensor([[[0.0, 0.5], [1.2, -0.3]], [[0.25, 0.75], [0.1, 0.9]], [[1.0, 0.0], [0.499, 0.501]]])  # RGB, CHW
    img = M.tensor2img(t.unsqueeze(0))
    assert img.dtype == np.uint8 and img.shape == (2, 2, 3)
    assert img[0, 0].tolist() == [255, 64, 0] and img[1, 0].tolist() == [127, 26, 255]   # BGR, clamp, round-half-even
    assert M.tensor2img(torch.rand(1, 1, 5, 4)).shape == (5, 4)
    a, _ = _imgs(3, 9, 7)
    M.imwrite_png(a, tmp_path / "sub" / "x.png")
    raw = (tmp_path / "sub" / "x.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, tag = struct.unpack(">I", raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        if tag == b"IDAT":
            idat += data
        pos += 12 + n
    assert hdr == (7, 9, 8, 2, 0, 0, 0)
    rows = zlib.decompress(idat)
    dec = np.frombuffer(rows, np.uint8).reshape(9, 1 + 7 * 3)[:, 1:].reshape(9, 7, 3)
    assert np.array_equal(dec, a[..., ::-1])   # file holds RGB
