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
