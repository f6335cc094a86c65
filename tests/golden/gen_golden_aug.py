#!/usr/bin/env python
"""Golden fixtures for the batch augmentations, produced by RUNNING THE REFERENCE on CPU (build container
only):  python tests/golden/gen_golden_aug.py  ->  aug.npz

  resize/*   F.interpolate(antialias=True) exactly as apply_augment / resizemix call it: bilinear and bicubic
             x4 up of a 2x3x16x16 batch, bicubic x1/4 down of 2x3x64x64, bicubic to an odd box size
  fn/<name>  mixup / cutmix / resizemix / cutblur on a 4x3x32x32 pair with their draws recorded
  run/<k>    apply_augment(gt 4x3x64x64, lq 4x3x16x16, scale 4, template augs / probs) for 16 seeds —
             both the single and the multi-augmentation branch occur — with EVERY draw recorded in order
"""

from __future__ import annotations

import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, save  # noqa: E402
from gen_golden_otf import pack_draws  # noqa: E402

AUGS = ["none", "mixup", "cutmix", "resizemix", "cutblur"]
PROB = [0.5, 0.1, 0.1, 0.1, 0.5]


class Recorder:
    def __init__(self):
        self.log = []

    def install(self, mod):
        R, py_random, np_rng = self, random, mod.rng

        class PyRandom:
            def choice(self, seq):
                v = py_random.choice(seq)
                R.log.append(("choice", v))
                return v

            def choices(self, pop, weights=None, **kw):
                v = py_random.choices(pop, weights, **kw)
                v0 = v[0]
                R.log.append(("choices", v0[0] if isinstance(v0, tuple) else AUGS[v0]))
                return v

        class NpRng:
            def uniform(self, *a, **k):
                v = np_rng.uniform(*a, **k)
                R.log.append(("uniform", float(v)))
                return v

            def random(self, *a, **k):
                v = np_rng.random(*a, **k)
                R.log.append(("random", float(v)))
                return v

            def integers(self, *a, **k):
                v = np_rng.integers(*a, **k)
                R.log.append(("integers", int(v)))
                return v

        mod.random, mod.rng = PyRandom(), NpRng()
        orig = torch.randperm

        def randperm(*a, **k):
            v = orig(*a, **k)
            R.log.append(("randperm", v.numpy().copy()))
            return v

        mod.torch.randperm = randperm

    def drain(self):
        log, self.log = self.log, []
        return log


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_aug.toml"
    tmp.write_text((HERE / "golden_esrgan.toml").read_text())
    install_reference(str(tmp))
    from neosr.data import augmentations as A

    rec = Recorder()
    rec.install(A)
    gen = torch.Generator().manual_seed(21)
    out = {}
    a16, a64 = torch.rand(2, 3, 16, 16, generator=gen), torch.rand(2, 3, 64, 64, generator=gen)
    out["resize/in16"], out["resize/in64"] = a16.numpy(), a64.numpy()
    out["resize/bilinear_up4"] = F.interpolate(a16, scale_factor=4, mode="bilinear", antialias=True).numpy()
    out["resize/bicubic_up4"] = F.interpolate(a16, scale_factor=4, mode="bicubic", antialias=True).numpy()
    out["resize/bicubic_down4"] = F.interpolate(a64, scale_factor=0.25, mode="bicubic", antialias=True).numpy()
    out["resize/bicubic_23x37"] = F.interpolate(a64, (23, 37), mode="bicubic", antialias=True).numpy()
    out["resize/bicubic_50x9"] = F.interpolate(a64, (50, 9), mode="bicubic", antialias=True).numpy()

    for name in ("mixup", "cutmix", "resizemix", "cutblur"):
        gt, lq = torch.rand(4, 3, 32, 32, generator=gen), torch.rand(4, 3, 32, 32, generator=gen)
        out[f"fn/{name}/gt"], out[f"fn/{name}/lq"] = gt.numpy().copy(), lq.numpy().copy()
        random.seed(5)
        torch.manual_seed(5)
        g2, l2 = getattr(A, name)(gt.clone(), lq.clone())
        out[f"fn/{name}/gt_out"], out[f"fn/{name}/lq_out"] = g2.numpy().copy(), l2.numpy().copy()
        pack_draws(out, f"fn/{name}/draws", rec.drain())

    kinds_seen = set()
    for k in range(16):
        random.seed(100 + k)
        torch.manual_seed(100 + k)
        gt, lq = torch.rand(4, 3, 64, 64, generator=gen), torch.rand(4, 3, 16, 16, generator=gen)
        out[f"run/{k}/gt"], out[f"run/{k}/lq"] = gt.numpy().copy(), lq.numpy().copy()
        g2, l2 = A.apply_augment(gt.clone(), lq.clone(), scale=4, augs=AUGS, prob=PROB)
        out[f"run/{k}/gt_out"], out[f"run/{k}/lq_out"] = g2.numpy().copy(), l2.numpy().copy()
        log = rec.drain()
        kinds_seen.add(tuple(v for kk, v in log if kk == "choices"))
        pack_draws(out, f"run/{k}/draws", log)
    print("augmentation picks seen:", sorted(kinds_seen))
    save("aug.npz", **out)


if __name__ == "__main__":
    main()
