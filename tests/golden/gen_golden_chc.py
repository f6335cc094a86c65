#!/usr/bin/env python
"""chc_loss with its cosine-similarity term and the "sum" reductions of the pixel losses, produced by RUNNING
THE REFERENCE on CPU (build container only):  python tests/golden/gen_golden_chc.py  ->  chc_lambda.npz

  chc/<crit>_<lam>   loss value and d loss / d pred of neosr.losses.basic_loss.chc_loss(criterion, loss_lambda)
                     for criterion in {l1, huber}, loss_lambda in {0, 5/255, 0.5} on (2,3,10,12) and (1,8,6,5)
  sum/<name>         L1Loss / MSELoss / HuberLoss with reduction="sum"
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, save  # noqa: E402


def main():
    install_reference(str(HERE / "golden_compact.toml"))
    from neosr.losses.basic_loss import HuberLoss, L1Loss, MSELoss, chc_loss

    gen = torch.Generator().manual_seed(19)
    A = {}
    for tag, shape in (("a", (2, 3, 10, 12)), ("b", (1, 8, 6, 5))):
        x = torch.rand(shape, generator=gen)
        y = (x + 0.3 * torch.randn(shape, generator=gen)).clamp(0, 1)
        A[f"{tag}/x"], A[f"{tag}/y"] = x.numpy(), y.numpy()
        for crit in ("l1", "huber"):
            for lam in (0.0, 5 / 255, 0.5):
                t = x.clone().requires_grad_(True)
                v = chc_loss(loss_weight=0.8, criterion=crit, loss_lambda=lam)(t, y)
                (v * 1.7).backward()
                A[f"{tag}/chc/{crit}_{lam:.6f}"] = np.float64(v.item())
                A[f"{tag}/chc/{crit}_{lam:.6f}/g"] = t.grad.numpy().copy()
        for name, cls in (("L1Loss", L1Loss), ("MSELoss", MSELoss), ("HuberLoss", HuberLoss)):
            t = x.clone().requires_grad_(True)
            v = cls(loss_weight=0.6, reduction="sum")(t, y)
            v.backward()
            A[f"{tag}/sum/{name}"] = np.float64(v.item())
            A[f"{tag}/sum/{name}/g"] = t.grad.numpy().copy()
    save("chc_lambda.npz", **A)


if __name__ == "__main__":
    main()
