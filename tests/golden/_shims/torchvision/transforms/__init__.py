"""Shim for the one torchvision transform the reference's losses use (torchvision is not installed here).

`GaussianBlur` restates torchvision's published algorithm (torchvision/transforms/_functional_tensor.py
`_get_gaussian_kernel1d/2d`, `gaussian_blur`): kernel1d = normalised exp(-0.5 (x / sigma)^2) on
linspace(-(k-1)/2, (k-1)/2, k), 2-D kernel = outer product, reflect padding k//2, depthwise conv2d.
Because this is a restatement of a third-party dependency, anything that depends on it is "parity unpinned"
by the reference itself.
"""

import torch
import torch.nn.functional as F

from . import functional  # noqa: F401


def _kernel1d(kernel_size: int, sigma: float) -> torch.Tensor:
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


class GaussianBlur(torch.nn.Module):
    def __init__(self, kernel_size, sigma=(0.1, 2.0)):
        super().__init__()
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.sigma = (float(sigma), float(sigma)) if isinstance(sigma, (int, float)) else tuple(sigma)

    def forward(self, img):
        # torchvision draws sigma ~ U(sigma_min, sigma_max) from the global torch RNG on every call
        sigma = torch.empty(1).uniform_(self.sigma[0], self.sigma[1]).item()
        kx, ky = self.kernel_size
        k2 = torch.mm(_kernel1d(ky, sigma)[:, None], _kernel1d(kx, sigma)[None, :]).to(img.dtype)
        k2 = k2.expand(img.shape[-3], 1, ky, kx)
        pad = [kx // 2, kx // 2, ky // 2, ky // 2]
        x = F.pad(img, pad, mode="reflect")
        return F.conv2d(x, k2, groups=img.shape[-3])
