#!/usr/bin/env python
"""Golden fixtures for the remaining optimizers, produced by RUNNING THE REFERENCE on CPU (build container
only):  python tests/golden/gen_golden_optim.py  ->  optim.npz

  <name>/...   reference `adan` (prox and no_prox), `adamw_sf` (warmup 3, incl. an eval()/train() switch) and
               `adamw_win` (win, win2 and the plain path) on two tensors, 5 steps with given gradients: the
               parameters after every step.  (Adam / NAdam are torch.optim classes: tested against torch itself.)
"""

from __future__ import annotations

import sys
import tempfile
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, save  # noqa: E402


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_optim.toml"
    tmp.write_text((HERE / "golden_esrgan.toml").read_text())
    install_reference(str(tmp))
    from neosr.optimizers.adamw_sf import adamw_sf
    from neosr.optimizers.adamw_win import adamw_win
    from neosr.optimizers.adan import adan

    cases = {
        "adan_prox": (adan, dict(lr=2e-3, betas=(0.98, 0.92, 0.99), weight_decay=0.02)),
        "adan_noprox": (adan, dict(lr=2e-3, betas=(0.98, 0.92, 0.99), weight_decay=0.02, no_prox=True)),
        "adamw_sf": (adamw_sf, dict(lr=2.5e-3, betas=(0.9, 0.999), weight_decay=0.01, warmup_steps=3)),
        "adamw_win": (adamw_win, dict(lr=5e-4, weight_decay=0.02, acceleration_mode="win")),
        "adamw_win2": (adamw_win, dict(lr=5e-4, weight_decay=0.02, acceleration_mode="win2")),
        "adamw_plain": (adamw_win, dict(lr=5e-4, weight_decay=0.02, acceleration_mode="none")),
    }
    gen = torch.Generator().manual_seed(17)
    A = {}
    for tag, (cls, kw) in cases.items():
        ps = [torch.randn(6, 5, generator=gen).requires_grad_(True), torch.randn(9, generator=gen).requires_grad_(True)]
        for i, p in enumerate(ps):
            A[f"{tag}/p0/{i}"] = p.detach().numpy().copy()
        opt = cls(ps, **kw)
        if hasattr(opt, "train"):
            opt.train()
        for step in range(1, 6):
            for i, p in enumerate(ps):
                g = torch.randn(p.shape, generator=gen) * (0.5 + 0.1 * step)
                A[f"{tag}/g{step}/{i}"] = g.numpy().copy()
                p.grad = g.clone()
            opt.step()
            for i, p in enumerate(ps):
                A[f"{tag}/p{step}/{i}"] = p.detach().numpy().copy()
            if tag == "adamw_sf" and step == 3:
                opt.eval()
                for i, p in enumerate(ps):
                    A[f"{tag}/p_eval/{i}"] = p.detach().numpy().copy()
                opt.train()
    save("optim.npz", **A)


if __name__ == "__main__":
    main()
