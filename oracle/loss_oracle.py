"""CPU oracle for the remaining template losses — TEST INFRASTRUCTURE ONLY.

PyTorch-CPU fp32 restatement of `mssim_loss` (neosr/losses/ssim_loss.py:11-163).

Parity status: PINNED against tests/golden/mssim.npz (reference run on CPU, tests/golden/gen_golden_losses.py).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def gaussian_window(window_size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """ssim_loss.py:44-55: normalised 1-D window and its outer product."""
    x = torch.arange(-(window_size // 2), window_size // 2 + 1)
    w = torch.exp(-0.5 * x**2 / (sigma * sigma))
    w /= w.sum()
    w = w.reshape(1, 1, 1, window_size)
    return torch.matmul(w.transpose(-1, -2), w)


def _ssim(x, y, win, c1, c2):
    """ssim_loss.py:142-163."""
    f = lambda t: F.conv2d(t, win, stride=1, padding=win.shape[-1] // 2, groups=t.shape[1])  # noqa: E731
    mu_x, mu_y = f(x), f(y)
    s2x, s2y, sxy = f(x * x) - mu_x * mu_x, f(y * y) - mu_y * mu_y, f(x * y) - mu_x * mu_y
    a1, a2 = 2 * mu_x * mu_y + c1, 2 * sxy + c2
    b1, b2 = mu_x.pow(2) + mu_y.pow(2) + c1, s2x + s2y + c2
    cs = a2 / b2
    return (a1 / b1) * cs, cs


def mssim_loss(x, y, window_size=11, sigma=1.5, k1=0.01, k2=0.03, L=1, loss_weight=1.0):
    """mssim_loss.forward / msssim (ssim_loss.py:110-140)."""
    win = gaussian_window(window_size, sigma).repeat(x.shape[1], 1, 1, 1).to(x.dtype)
    c1, c2 = (k1 * L) ** 2, (k2 * L) ** 2
    comps = []
    for i, w in enumerate((0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
        ssim, cs = _ssim(x, y, win, c1, c2)
        if i == 4:
            comps.append(ssim.mean() ** w)
        else:
            comps.append(cs.mean() ** w)
            pad = [s % 2 for s in x.shape[2:]]
            x, y = F.avg_pool2d(x, 2, 2, padding=pad), F.avg_pool2d(y, 2, 2, padding=pad)
    return loss_weight * (1 - math.prod(comps))


# --------------------------------------------------------------------------------------------
# consistency_loss (neosr/losses/consistency_loss.py:14-192)
# --------------------------------------------------------------------------------------------


def gaussian_blur_reflect(img, kernel_size=21, sigma=3.0):
    """torchvision GaussianBlur as the reference calls it (kernel 21, sigma 3): restated from torchvision's
    published `_get_gaussian_kernel1d` + reflect pad + depthwise conv (torchvision itself is not installed:
    this part is "parity unpinned" by the reference; the shim in tests/golden/_shims uses the same algorithm)."""
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :]).to(img.dtype).expand(img.shape[-3], 1, kernel_size, kernel_size)
    p = kernel_size // 2
    return F.conv2d(F.pad(img, [p, p, p, p], mode="reflect"), k2, groups=img.shape[-3])


def _lin_rgb(img):
    return torch.where(img <= 0.04045, img / 12.92, torch.pow((img + 0.055) / 1.055, 2.4))


def rgb_to_oklab_chroma(img):
    img = _lin_rgb(img)
    r, g, b = img[:, 0], img[:, 1], img[:, 2]
    l = 0.4122214708 * r + 0.5363325363 * g + 0.0514459929 * b
    m = 0.2119034982 * r + 0.6806995451 * g + 0.1073969566 * b
    s = 0.0883024619 * r + 0.2817188376 * g + 0.6299787005 * b
    l_, m_, s_ = (t.sign() * t.abs().pow(1 / 3) for t in (l, m, s))
    a = 1.9779984951 * l_ - 2.4285922050 * m_ + 0.4505937099 * s_
    b2 = 0.0259040371 * l_ + 0.7827717662 * m_ - 0.8086757660 * s_
    return torch.stack([a, b2], dim=1)


def rgb_to_l_star(img):
    img = _lin_rgb(img.permute(0, 2, 3, 1)) @ torch.tensor([0.2126, 0.7152, 0.0722], dtype=img.dtype)
    img = torch.where(img <= (216 / 24389), img * (img * (24389 / 27)), img.sign() * img.abs().pow(1 / 3) * 116 - 16)
    return torch.clamp(img / 100, 0, 1)


def _chc01(a, b):
    """chc_loss(loss_lambda=0, clip_min=0, clip_max=1), huber criterion (basic_loss.py:192-219)."""
    return torch.mean(torch.clamp(torch.sqrt((a - b) ** 2 + 1e-12), 0, 1))


def consistency_loss(x, gt, blur=True, cosim=True, saturation=1.0, brightness=1.0, loss_weight=1.0):
    x, gt = torch.clamp(x, 1 / 255, 1), torch.clamp(gt, 1 / 255, 1)
    if blur:
        in_luma = rgb_to_l_star(torch.clamp(gaussian_blur_reflect(x), 0, 1))
        tg_luma = rgb_to_l_star(torch.clamp(gaussian_blur_reflect(gt), 0, 1)) * brightness
    else:
        in_luma, tg_luma = rgb_to_l_star(x), rgb_to_l_star(gt) * brightness
    in_ch = torch.clamp(rgb_to_oklab_chroma(x) + 0.5, 0, 1)
    tg_ch = torch.clamp(rgb_to_oklab_chroma(gt) * saturation + 0.5, 0, 1)
    loss = _chc01(in_luma, tg_luma) + _chc01(in_ch, tg_ch)
    if cosim:
        cs = lambda a, b: 1 - F.cosine_similarity(a, b, dim=1, eps=1e-20).mean()  # noqa: E731
        cos = 0.5 * cs(in_ch, tg_ch) + 0.5 * cs(in_luma, tg_luma)
        if cos < 1e-3:
            loss = loss + cos
    return loss * loss_weight
