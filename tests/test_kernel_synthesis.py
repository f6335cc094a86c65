"""Blur / sinc kernel synthesis (neosr_amd/data/degradations.py) vs the reference's generators
(fixture tests/golden/kernels.npz).  CPU only; fp64, tolerance 1e-6 abs as per SURVEY §8c — in practice
bit-identical because the same numpy expressions are evaluated on the same draws."""

from __future__ import annotations

import random

import numpy as np

from neosr_amd.data import degradations as K
from tests.conftest import load_golden


def test_parametric_families_match_reference():
    fix = load_golden("kernels.npz")
    for k in (7, 13, 21):
        pairs = {
            f"gauss_iso_{k}": K.bivariate_gaussian(k, 1.7, 1.7, 0, True),
            f"gauss_aniso_{k}": K.bivariate_gaussian(k, 2.3, 0.8, 0.6, False),
            f"gen_iso_{k}": K.bivariate_generalized_gaussian(k, 1.4, 1.4, 0, 0.7, True),
            f"gen_aniso_{k}": K.bivariate_generalized_gaussian(k, 2.0, 1.1, -1.1, 2.5, False),
            f"plat_iso_{k}": K.bivariate_plateau(k, 1.9, 1.9, 0, 1.6, True),
            f"plat_aniso_{k}": K.bivariate_plateau(k, 2.6, 0.9, 2.2, 1.2, False),
            f"sinc_{k}": K.circular_lowpass_kernel(np.pi / 2.5, k, pad_to=21),
        }
        for name, got in pairs.items():
            assert got.shape == fix[name].shape, name
            assert np.abs(got - fix[name]).max() < 1e-12, name
            assert abs(got.sum() - 1) < 1e-12


def test_seeded_random_mixed_kernels_match_reference_stream():
    """same python-random seed + same Generator seed => same draw order => same kernels"""
    fix = load_golden("kernels.npz")
    s = K.KernelSampler(np.random.default_rng(1024))
    kinds = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
    prob = [0.45, 0.25, 0.12, 0.03, 0.12, 0.03]
    random.seed(1234)
    for i in range(24):
        k = random.choice([7, 9, 11, 13, 15, 17, 19, 21])
        noise = [0.75, 1.25] if i % 3 == 0 else None
        got = s.mixed(kinds, prob, k, [0.2, 3], [0.2, 3], [-np.pi, np.pi], [0.5, 4], [1, 2], noise_range=noise)
        assert np.abs(got - fix[f"mixed_{i:02d}"]).max() < 1e-12, i


def test_otf_kernel_batch_contract():
    opt = {"blur_kernel_size": 7, "kernel_list": ["iso", "aniso"], "kernel_prob": [0.5, 0.5], "sinc_prob": 0.3,
           "blur_sigma": [0.2, 3], "betag_range": [0.5, 4], "betap_range": [1, 2],
           "kernel_list2": ["generalized_iso", "plateau_aniso"], "kernel_prob2": [0.5, 0.5], "sinc_prob2": 0.3,
           "blur_sigma2": [0.2, 1.5], "betag_range2": [0.5, 4], "betap_range2": [1, 2], "final_sinc_prob": 0.5}
    random.seed(0)
    batch = K.KernelSampler(np.random.default_rng(0)).otf_kernel_batch(opt, 6)
    for name in ("kernel1", "kernel2", "sinc_kernel"):
        t = batch[name]
        assert tuple(t.shape) == (6, 21, 21) and t.dtype.is_floating_point
        assert np.allclose(t.sum((1, 2)).numpy(), 1.0, atol=1e-5)
