"""GPU parity of the degradation-bank kernels (through the C ABI) against fixtures produced by the
reference's own `filter2D`, `F.interpolate`, `random_add_*_noise_pt`, `DiffJPEG` and `otf.feed_data`
(stochastic parts replay the reference's recorded draws).  Tolerance 1e-3 rel (north_star); DiffJPEG is bit-exact
and the 8-bit quantised LQ of `otf.feed_data` has zero 1/255 rounding flips against the reference's."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import DEG_OPT, GOLDEN, ROOT, load_draws, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def prims():
    return load_golden("degrade_prims.npz")


def G(a):
    return torch.from_numpy(np.array(a)).to(DEV)


def test_filter2d_vs_reference(prims):
    from neosr_amd.hip import degrade as D

    img = G(prims["f2d_img"])
    assert rel_err(D.filter2d(img, G(prims["f2d_k"])), torch.from_numpy(prims["f2d_out"])) < 1e-5
    assert rel_err(D.filter2d(img, G(prims["f2d_k1"])), torch.from_numpy(prims["f2d_out1"])) < 1e-5
    with pytest.raises(ValueError, match="Wrong kernel size"):
        D.filter2d(img, torch.ones(1, 4, 4, device=DEV))


def test_filter2d_full_size_vs_oracle():
    """blur1 shape of config 3 on one sample: (1,3,512,512), 21x21 anisotropic kernel."""
    from neosr_amd.hip import degrade as D
    from oracle import degrade_oracle as dorc

    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, 512, 512, generator=g)
    k = torch.rand(1, 21, 21, generator=g)
    k /= k.sum()
    assert rel_err(D.filter2d(img.to(DEV), k.to(DEV)), dorc.filter2d(img, k)) < 1e-5


def test_resize_all_forms_vs_reference(prims):
    from neosr_amd.hip import degrade as D

    img = G(prims["f2d_img"])
    worst = 0.0
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.5, 0.73, 1.37):
            ref = torch.from_numpy(prims[f"rs_sf_{mode}_{s}"])
            out = D.resize(img, scale_factor=s, mode=mode)
            assert out.shape == ref.shape
            worst = max(worst, rel_err(out, ref))
        for size in ((25, 35), (32, 32), (61, 90)):
            ref = torch.from_numpy(prims[f"rs_sz_{mode}_{size[0]}x{size[1]}"])
            worst = max(worst, rel_err(D.resize(img, size=size, mode=mode), ref))
    assert worst < 1e-5, worst


def test_noise_kernels_replaying_reference_draws(prims):
    from neosr_amd.data.draws import ReplayDraws
    from neosr_amd.hip import degrade as D

    img = G(prims["f2d_img"])
    b, _, h, w = img.shape
    for tag, gray_prob in (("gn", 0.6), ("gn0", 0.0)):
        d = ReplayDraws(load_draws(prims, f"{tag}_draws"), DEV)
        sigma = d.rand(b) * (30 - 1) + 1
        gray = (d.rand(b) < gray_prob).float()
        ngray = d.randn(h, w) if float(gray.sum()) > 0 else None
        noise = d.randn(b, 3, h, w)
        assert d.exhausted()
        assert rel_err(D.gaussian_noise(img, noise, ngray, sigma, gray), torch.from_numpy(prims[f"{tag}_out"])) < 1e-5
    for tag, gray_prob in (("pn", 0.6), ("pn0", 0.0)):
        d = ReplayDraws(load_draws(prims, f"{tag}_draws"), DEV)
        scale = d.rand(b) * (3 - 0.05) + 0.05
        gray = (d.rand(b) < gray_prob).float()
        pg = vg = None
        if float(gray.sum()) > 0:
            rate_g, vg = D.poisson_rate(img, gray=True)
            pg = d.poisson(rate_g)
        rate, vals = D.poisson_rate(img, gray=False)
        p = d.poisson(rate)
        assert d.exhausted()
        assert rel_err(D.poisson_noise(img, p, vals, pg, vg, scale, gray), torch.from_numpy(prims[f"{tag}_out"])) < 1e-5


def test_poisson_rate_level_count_vs_oracle():
    """2^ceil(log2(#distinct 8-bit levels)) from the bitmap kernel == torch.unique on the CPU."""
    from neosr_amd.hip import degrade as D
    from oracle import degrade_oracle as dorc

    g = torch.Generator().manual_seed(3)
    img = torch.rand(3, 3, 40, 56, generator=g)
    img[1] = (img[1] * 5).round() / 5          # few levels
    img[2] = 0.5                                # a single level
    for gray in (False, True):
        rate, vals = D.poisson_rate(img.to(DEV), gray)
        rate_ref, vals_ref = dorc.poisson_rate(img, gray)
        assert torch.equal(vals.cpu(), vals_ref)
        assert rel_err(rate, rate_ref) < 1e-6


def test_diffjpeg_vs_reference(prims):
    """BIT-EXACT since round 6: the kernel evaluates the colour transforms, the 8x8 DCT / IDCT (fmaf chains in
    the index order of the reference's `tensordot`, the reference's float32(float64 product) cosine table), the quantiser and
    the final division in the reference's operations and order — no quantised coefficient rounds the other way
    (neosr/utils/diffjpeg.py:65-555)."""
    from neosr_amd.hip import degrade as D

    for name in ("a", "b"):
        q = G(prims[f"jpg_{name}_q"])
        out = D.diffjpeg(G(prims[f"jpg_{name}_img"]), q).cpu()
        ref = torch.from_numpy(prims[f"jpg_{name}_out"])
        assert torch.equal(out, ref), (name, int((out != ref).sum()), float((out - ref).abs().max()))
        assert torch.equal(q.cpu(), torch.from_numpy(prims[f"jpg_{name}_q"]))  # quality not mutated


def test_quantise_crop_gather_bit_exact(prims):
    from neosr_amd.hip import degrade as D

    assert torch.equal(D.quantize_u8(G(prims["q_in"])).cpu(), torch.from_numpy(prims["q_out"]))
    x = torch.arange(2 * 3 * 9 * 11, dtype=torch.float32).reshape(2, 3, 9, 11)
    assert torch.equal(D.crop(x.to(DEV), 2, 3, 5, 6).cpu(), x[:, :, 2:7, 3:9])
    idx = torch.tensor([3, 0, 2, 1, 1])
    rows = torch.arange(4 * 6, dtype=torch.float32).reshape(4, 2, 3)
    assert torch.equal(D.gather_rows(rows.to(DEV), idx.to(DEV)).cpu(), rows[idx])


def test_otf_feed_data_replaying_reference_draws():
    """OUR otf.feed_data (HIP kernels) on the reference's inputs and recorded draws, 3 calls incl. the
    pair pool filling and shuffling: same LQ/GT pair as the reference produced."""
    from neosr_amd.data.draws import ReplayDraws
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options

    fix = load_golden("otf_feed.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_otf.toml")])
    assert opt["degradations"] == DEG_OPT
    model = build_model(opt)
    for it in (1, 2, 3):
        d = ReplayDraws(load_draws(fix, f"it{it}/draws"), DEV)
        model.draws = d
        model.feed_data({k: torch.from_numpy(fix[f"it{it}/{k}"]) for k in ("gt", "kernel1", "kernel2", "sinc_kernel")})
        assert d.exhausted()
        ref_lq = torch.from_numpy(fix[f"it{it}/lq"])
        diff = (model.lq.cpu() - ref_lq).abs()
        # zero flips: no LQ pixel is a quantisation step away from the reference's
        assert float(diff.max()) <= 1e-6, (it, float(diff.max()), int((diff > 1e-6).sum()))
        assert torch.equal(model.gt.cpu(), torch.from_numpy(fix[f"it{it}/gt_out"]))
    model.optimize_parameters(1)  # the degraded pair feeds the HIP training step
    assert np.isfinite(model.get_current_log()["l_g_pix"])


@pytest.mark.parametrize("lam", [0.3, 3.0, 9.9, 10.1, 50.0, 800.0])
def test_poisson_sample_distribution(lam):
    """`neosr_poisson_sample` against the Poisson pmf (scipy): mean / variance and a chi-square over the
    central bins; pure function of (seed, offset)."""
    from scipy import stats

    from neosr_amd.hip import degrade as D

    n = 1 << 20
    rate = torch.full((n,), lam, device="cuda")
    a = D.poisson_sample(rate, 1234, 0)
    b = D.poisson_sample(rate, 1234, 0)
    c = D.poisson_sample(rate, 1234, 4 * n)
    assert torch.equal(a, b) and not torch.equal(a, c)
    x = a.cpu().numpy().astype(np.int64)
    assert x.min() >= 0
    assert abs(x.mean() - lam) < 5 * np.sqrt(lam / n)
    assert abs(x.var() - lam) < 8 * lam * np.sqrt(2.0 / n) + 5 * np.sqrt(lam / n)
    lo, hi = int(stats.poisson.ppf(1e-4, lam)), int(stats.poisson.ppf(1 - 1e-4, lam))
    ks = np.arange(lo, hi + 1)
    exp = stats.poisson.pmf(ks, lam) * n
    obs = np.array([(x == k).sum() for k in ks], dtype=np.float64)
    keep = exp > 20
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    dof = keep.sum() - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (chi2, dof)


def test_poisson_sample_mixed_rates_and_zero():
    from neosr_amd.hip import degrade as D

    rate = torch.tensor([0.0, 0.0, 1e-6, 5.0, 20.0, 3000.0], device="cuda").repeat(4096)
    p = D.poisson_sample(rate, 7, 0).view(4096, 6).cpu()
    assert (p[:, :2] == 0).all() and (p >= 0).all() and torch.equal(p, p.round())
    assert abs(p[:, 3].mean() - 5.0) < 0.2 and abs(p[:, 4].mean() - 20.0) < 0.4 and abs(p[:, 5].mean() - 3000.0) < 5


def test_blur_kernels_on_device_vs_reference_fixture_and_host_path():
    """`neosr_blur_kernels` (device, float64) against (i) the reference generators' outputs for fixed parameters
    (tests/golden/kernels.npz) and (ii) the host numpy path for 64 seeded otf draws, which consume the RNG streams
    identically (SURVEY §8c `kernels/`: 1e-6 abs; observed ~1e-9)."""
    import random

    from neosr_amd.data import degradations as K
    from neosr_amd.hip import degrade as D
    from tools.bench_degrade import DEG_TABLE

    fix = load_golden("kernels.npz")
    rows, refs = [], []
    for k in (7, 13, 21):
        for name, row in ((f"gauss_iso_{k}", [0, k, 1.7, 1.7, 0, 1, 1, 0]), (f"gauss_aniso_{k}", [0, k, 2.3, 0.8, 0.6, 1, 0, 0]),
                          (f"gen_iso_{k}", [1, k, 1.4, 1.4, 0, 0.7, 1, 0]), (f"gen_aniso_{k}", [1, k, 2.0, 1.1, -1.1, 2.5, 0, 0]),
                          (f"plat_iso_{k}", [2, k, 1.9, 1.9, 0, 1.6, 1, 0]), (f"plat_aniso_{k}", [2, k, 2.6, 0.9, 2.2, 1.2, 0, 0]),
                          (f"sinc_{k}", [3, k, 0, 0, 0, np.pi / 2.5, 1, 0])):
            ref = np.array(fix[name], dtype=np.float64)
            p = (21 - ref.shape[0]) // 2
            refs.append(np.pad(ref, ((p, p), (p, p))))
            rows.append([float(v) for v in row])
    out = D.blur_kernels(torch.tensor(rows, dtype=torch.float64, device=DEV)).cpu().numpy()
    assert np.abs(out - np.stack(refs)).max() < 1e-7

    class Py:
        def __init__(self, seed):
            self.r = random.Random(seed)

        def choices(self, *a, **k):
            return self.r.choices(*a, **k)

        def choice(self, a):
            return self.r.choice(a)

    host = K.KernelSampler(np.random.default_rng(11), Py(5)).otf_kernel_batch(DEG_TABLE, 64)
    dev = K.KernelSampler(np.random.default_rng(11), Py(5)).otf_kernel_batch_device(DEG_TABLE, 64, DEV)
    for name in ("kernel1", "kernel2", "sinc_kernel"):
        assert float((dev[name].cpu() - host[name]).abs().max()) < 1e-7, name
        assert dev[name].shape == (64, 21, 21)


def test_normal_sample_distribution_and_determinism():
    """`neosr_normal_sample` (Philox + Box-Muller): a pure function of (seed, offset); mean / variance / kurtosis and a
    Kolmogorov-Smirnov test against N(0, 1); LiveDraws.randn advances the device generator like a torch draw"""
    from scipy import stats

    from neosr_amd.data.draws import LiveDraws
    from neosr_amd.hip import degrade as D

    n = 1 << 20
    a = D.normal_sample((n,), 99, 0, DEV)
    assert torch.equal(a, D.normal_sample((n,), 99, 0, DEV)) and not torch.equal(a, D.normal_sample((n,), 99, n, DEV))
    x = a.cpu().double().numpy()
    assert abs(x.mean()) < 5 / np.sqrt(n) and abs(x.var() - 1) < 8 * np.sqrt(2 / n)
    assert abs(stats.kurtosis(x)) < 0.03
    assert stats.kstest(x[:200000], "norm").pvalue > 1e-3
    odd = D.normal_sample((3, 5, 7), 1, 2, DEV)
    assert odd.shape == (3, 5, 7) and torch.isfinite(odd).all()
    torch.manual_seed(5)
    d = LiveDraws(5, DEV)
    u, v = d.randn(2, 3, 8, 8), d.randn(2, 3, 8, 8)
    torch.manual_seed(5)
    assert torch.equal(u, LiveDraws(5, DEV).randn(2, 3, 8, 8)) and not torch.equal(u, v)
