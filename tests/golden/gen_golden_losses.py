#!/usr/bin/env python
"""Golden fixtures for the remaining template losses, produced by RUNNING THE REFERENCE on CPU (build
container only):  python tests/golden/gen_golden_losses.py  ->  mssim.npz, consistency.npz

  mssim.npz   reference `mssim_loss` on two smooth-ish image pairs (2x3x64x64 and 1x3x96x128): loss value and
              gradient wrt the prediction
  consistency.npz  reference `consistency_loss` (blur + cosim, saturation 1.1, brightness 0.95) on a pair close
              enough for the cosine term to be added and on one where it is not: loss and gradient
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

    # ---- consistency_loss (torchvision's GaussianBlur comes from tests/golden/_shims: restated algorithm)
    from neosr.losses.consistency_loss import consistency_loss

    out = {}
    cases = {"near": 0.01, "far": 0.25}  # "near": cosine term below its 1e-3 threshold -> added; "far": not
    for tag, noise in cases.items():
        gt = smooth(gen, 2, 64, 80)
        x = (gt + noise * torch.randn(2, 3, 64, 80, generator=gen)).clamp(0, 1).requires_grad_(True)
        crit = consistency_loss(saturation=1.1, brightness=0.95, loss_weight=0.8)
        torch.manual_seed(0)
        loss = crit(x, gt)
        loss.backward()
        out[f"{tag}/x"], out[f"{tag}/gt"] = x.detach().numpy(), gt.numpy()
        out[f"{tag}/loss"], out[f"{tag}/gx"] = loss.detach().numpy(), x.grad.numpy().copy()
    save("consistency.npz", **out)


if __name__ == "__main__":
    main()
