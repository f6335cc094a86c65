"""Pins oracle/degrade_oracle.py to fixtures produced by the reference's own degradation code
(tests/golden/gen_golden_otf.py).  CPU only."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from neosr_amd.data.draws import ReplayDraws
from oracle import degrade_oracle as dorc
from tests.conftest import DEG_OPT, load_draws, load_golden, rel_err


@pytest.fixture(scope="module")
def prims():
    return load_golden("degrade_prims.npz")


def T(a):
    return torch.from_numpy(np.array(a))


def test_filter2d(prims):
    img = T(prims["f2d_img"])
    assert rel_err(dorc.filter2d(img, T(prims["f2d_k"])), T(prims["f2d_out"])) < 1e-6
    assert rel_err(dorc.filter2d(img, T(prims["f2d_k1"])), T(prims["f2d_out1"])) < 1e-6


def test_resize_forms(prims):
    img = T(prims["f2d_img"])
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.5, 0.73, 1.37):
            assert torch.equal(dorc.resize(img, scale_factor=s, mode=mode), T(prims[f"rs_sf_{mode}_{s}"]))
        for size in ((25, 35), (32, 32), (61, 90)):
            assert torch.equal(dorc.resize(img, size=size, mode=mode), T(prims[f"rs_sz_{mode}_{size[0]}x{size[1]}"]))


def _replay_gaussian(img, draws, lo, hi, gray_prob):
    d = ReplayDraws(draws)
    b, _, h, w = img.shape
    sigma = d.rand(b) * (hi - lo) + lo
    gray = (d.rand(b) < gray_prob).float()
    ngray = d.randn(h, w) if float(gray.sum()) > 0 else None
    noise = d.randn(b, 3, h, w)
    assert d.exhausted()
    return dorc.add_gaussian_noise(img, noise, ngray, sigma, gray)


def _replay_poisson(img, draws, lo, hi, gray_prob):
    d = ReplayDraws(draws)
    b = img.size(0)
    scale = d.rand(b) * (hi - lo) + lo
    gray = (d.rand(b) < gray_prob).float()
    Pg = vg = None
    if float(gray.sum()) > 0:
        rate_g, vg = dorc.poisson_rate(img, gray=True)
        Pg = d.poisson(rate_g)
    rate, vals = dorc.poisson_rate(img, gray=False)
    P = d.poisson(rate)
    assert d.exhausted()
    return dorc.add_poisson_noise(img, P, vals, Pg, vg, scale, gray)


def test_noise_given_reference_draws(prims):
    img = T(prims["f2d_img"])
    assert rel_err(_replay_gaussian(img, load_draws(prims, "gn_draws"), 1, 30, 0.6), T(prims["gn_out"])) < 1e-6
    assert rel_err(_replay_gaussian(img, load_draws(prims, "gn0_draws"), 1, 30, 0.0), T(prims["gn0_out"])) < 1e-6
    assert rel_err(_replay_poisson(img, load_draws(prims, "pn_draws"), 0.05, 3, 0.6), T(prims["pn_out"])) < 1e-6
    assert rel_err(_replay_poisson(img, load_draws(prims, "pn0_draws"), 0.05, 3, 0.0), T(prims["pn0_out"])) < 1e-6


def test_diffjpeg(prims):
    for name in ("a", "b"):
        out = dorc.diffjpeg(T(prims[f"jpg_{name}_img"]), T(prims[f"jpg_{name}_q"]))
        ref = T(prims[f"jpg_{name}_out"])
        assert rel_err(out, ref) < 1e-5
        assert float((out - ref).abs().max()) < 1e-4


def test_quantise_and_quality_factor(prims):
    x = T(prims["q_in"])
    assert torch.equal(torch.clamp((x * 255.0).round(), 0, 255) / 255.0, T(prims["q_out"]))
    assert torch.allclose(dorc.quality_to_factor(T(prims["qf_q"])), T(prims["qf_f"]), rtol=1e-6)


def test_full_feed_data_matches_reference():
    """3 feed_data calls incl. the pair pool (fills at call 2, shuffles at call 3)."""
    fix = load_golden("otf_feed.npz")
    pool = dorc.PairPool(4, 2)
    for it in (1, 2, 3):
        d = ReplayDraws(load_draws(fix, f"it{it}/draws"))
        g = lambda k: T(fix[f"it{it}/{k}"])  # noqa: E731
        lq, gt = dorc.otf_feed_data(g("gt"), g("kernel1"), g("kernel2"), g("sinc_kernel"), DEG_OPT, 4, 16, d)
        lq, gt = pool.step(lq, gt, d)
        assert d.exhausted()
        ref_lq = g("lq")
        assert lq.shape == ref_lq.shape
        # 8-bit quantised output: allow isolated 1/255 flips from fp re-association upstream
        diff = (lq - ref_lq).abs()
        assert float(diff.max()) <= 1.0 / 255 + 1e-6
        assert float((diff > 1e-6).float().mean()) < 0.01
        assert torch.equal(gt, g("gt_out"))
