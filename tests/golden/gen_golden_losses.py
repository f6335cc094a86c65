#!/usr/bin/env python
"""Golden fixtures for the remaining template losses, produced by RUNNING THE REFERENCE on CPU (build
container only):  python tests/golden/gen_golden_losses.py  ->  mssim.npz

  mssim.npz   reference `mssim_loss` on two smooth-ish image pairs (2x3x64x64 and 1x3x96x128): loss value and
              gradient wrt the prediction
"""

from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, save  # noqa: E402


def smooth(gen, b, h, w):
    t = torch.rand(b, 3, h // 4, w // 4, generator=gen)
    t = F.interpolate(t, size=(h, w), mode="bicubic", align_corners=False).clamp(0, 1)
    return (t + 0.05 * torch.randn(b, 3, h, w, generator=gen)).clamp(0, 1)


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_losses.toml"
    tmp.write_text((HERE / "golden_esrgan.toml").read_text())
    install_reference(str(tmp))
    from neosr.losses.ssim_loss import mssim_loss

    gen = torch.Generator().manual_seed(13)
    out = {}
    for tag, (b, h, w), lw in (("a", (2, 64, 64), 1.0), ("b", (1, 96, 128), 0.7)):
        gt = smooth(gen, b, h, w)
        x = (gt + 0.1 * torch.randn(b, 3, h, w, generator=gen)).clamp(0, 1).requires_grad_(True)
        loss = mssim_loss(loss_weight=lw)(x, gt)
        loss.backward()
        out[f"{tag}/x"], out[f"{tag}/gt"] = x.detach().numpy(), gt.numpy()
        out[f"{tag}/loss"], out[f"{tag}/gx"] = loss.detach().numpy(), x.grad.numpy().copy()
        out[f"{tag}/loss_weight"] = torch.tensor(lw).numpy()
    save("mssim.npz", **out)


if __name__ == "__main__":
    main()
