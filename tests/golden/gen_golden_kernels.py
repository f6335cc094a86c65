#!/usr/bin/env python
"""Blur/sinc kernel fixtures from the reference's own generators (build container only):
    python tests/golden/gen_golden_kernels.py   ->  tests/golden/kernels.npz
* parametric families on fixed parameter grids (sizes 7..21)
* 24 seeded draws of `random_mixed_kernels` (+ circular low-pass) with python `random` seeded 1234 and
  the reference's module Generator (default_rng(manual_seed=1024), fresh at import)."""
from __future__ import annotations

import random
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import TOML_TMPL, install_reference, save  # noqa: E402


def main():
    tmp = Path(tempfile.mkdtemp()) / "k.toml"
    tmp.write_text(TOML_TMPL.format(arch="compact", net='type = "compact"'))
    install_reference(str(tmp))
    import neosr.data.degradations as degr

    A = {}
    for k in (7, 13, 21):
        A[f"gauss_iso_{k}"] = degr.bivariate_Gaussian(k, 1.7, 1.7, 0, isotropic=True)
        A[f"gauss_aniso_{k}"] = degr.bivariate_Gaussian(k, 2.3, 0.8, 0.6, isotropic=False)
        A[f"gen_iso_{k}"] = degr.bivariate_generalized_Gaussian(k, 1.4, 1.4, 0, 0.7, isotropic=True)
        A[f"gen_aniso_{k}"] = degr.bivariate_generalized_Gaussian(k, 2.0, 1.1, -1.1, 2.5, isotropic=False)
        A[f"plat_iso_{k}"] = degr.bivariate_plateau(k, 1.9, 1.9, 0, 1.6, isotropic=True)
        A[f"plat_aniso_{k}"] = degr.bivariate_plateau(k, 2.6, 0.9, 2.2, 1.2, isotropic=False)
        A[f"sinc_{k}"] = degr.circular_lowpass_kernel(np.pi / 2.5, k, pad_to=21)
    kinds = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
    prob = [0.45, 0.25, 0.12, 0.03, 0.12, 0.03]
    random.seed(1234)
    for i in range(24):
        k = random.choice([7, 9, 11, 13, 15, 17, 19, 21])
        noise = [0.75, 1.25] if i % 3 == 0 else None
        A[f"mixed_{i:02d}"] = degr.random_mixed_kernels(kinds, prob, k, [0.2, 3], [0.2, 3], [-np.pi, np.pi],
                                                         [0.5, 4], [1, 2], noise_range=noise)
    save("kernels.npz", **A)


if __name__ == "__main__":
    main()
