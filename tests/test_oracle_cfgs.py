"""Pins oracle/step_oracle.py (`ConfigTrainer`: the whole iteration of a BASELINE config restated on CPU) to the
reference-run trajectories tests/golden/step_cfg{2,3,4}.npz (gen_golden_cfgs.py), and the swinir_medium
restatement to the reference's forward + backward of the network AS NAMED (cfg3_swinir_medium.npz)."""

from __future__ import annotations

import random
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import gan_oracle as gorc
from oracle import hat_oracle as horc
from oracle import swinir_oracle as sorc
from oracle.step_oracle import ConfigTrainer
from tests.conftest import GOLDEN, ROOT, group, load_draws, load_golden, rel_err

T = lambda a: torch.from_numpy(np.array(a))  # noqa: E731


def _opt(name):
    from neosr_amd.utils.options import parse_options

    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / f"golden_{name}.toml")])
    return opt


def _init_g(opt, fix):
    """initial generator weights: shipped for the small nets; for swinir_small / hat_s the product arch's seeded init,
    which reproduces the reference's draw for draw (checked against the fixture's per-tensor checksums)"""
    if "init_g/keys" not in fix:
        return group(fix, "init_g")
    from neosr_amd.archs import build_network

    torch.manual_seed(1024)
    random.seed(1024)
    net = build_network(dict(opt["network_g"]))
    sd = net.state_dict()
    keys = [str(k) for k in fix["init_g/keys"]]
    s = np.array([float(sd[k].double().sum()) for k in keys])
    np.testing.assert_allclose(s, fix["init_g/sum"], rtol=1e-6, atol=1e-5)
    return OrderedDict((k, v.detach().clone()) for k, v in sd.items())


def werr(a, b) -> float:
    """||a - b|| / max(||b||, 1e-3 sqrt(n)): zero-initialised biases have moved by ~1e-6 after three adan_sf steps, where
    the direction of m / sqrt(n) amplifies 1e-7 gradient differences; their error is judged against an RMS floor of 1e-3"""
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm()) / max(float(b.norm()), 1e-3 * b.numel() ** 0.5)


def check_final(fix, G, D, tol):
    if "final_g/sum" in fix:
        keys = [str(k) for k in fix["init_g/keys"]]
        s = np.array([float(G[k].double().sum()) for k in keys])
        n = np.array([G[k].numel() for k in keys])
        bad = np.abs(s - fix["final_g/sum"]) > tol * np.maximum(fix["final_g/abs"], 1e-3 * n)  # floor: see werr()
        assert not bad.any(), [keys[i] for i in np.nonzero(bad)[0]]
        for k in [f for f in fix if f.startswith("final_g/w/")]:
            assert werr(G[k[len("final_g/w/"):]], T(fix[k])) < tol, k
    else:
        for k, v in group(fix, "final_g").items():
            assert werr(G[k], v) < tol, k
    for k, v in group(fix, "final_d").items():
        assert werr(D[k], v) < tol, k


@pytest.mark.parametrize("name", ["cfg3", "cfg2", "cfg4"])
def test_config_trainer_vs_reference_trajectory(name):
    from neosr_amd.data.draws import ReplayDraws

    fix = load_golden(f"step_{name}.npz")
    opt = _opt(name)
    gp = _init_g(opt, fix)
    dp = group(fix, "init_d") or None
    tr = ConfigTrainer(opt, gp, dp, gorc.vgg_seeded_weights())
    keys = [str(k) for k in fix["log_keys"]]
    otf = opt["model_type"] == "otf"
    for it in range(1, fix["log"].shape[0] + 1):
        if otf:
            tr.draws = ReplayDraws(load_draws(fix, f"it{it}/draws"))
            tr.feed_data({k: T(fix[f"it{it}/{k}"]) for k in ("gt", "kernel1", "kernel2", "sinc_kernel")})
            assert tr.draws.exhausted()
            assert float((tr.lq - T(fix[f"it{it}/lq"])).abs().max()) <= 1.0 / 255 + 1e-6
            assert torch.equal(tr.gt, T(fix[f"it{it}/gt_out"]))
            tr.lq = T(fix[f"it{it}/lq"])  # a JPEG rounding flip must not leak into the step comparison
        else:
            tr.feed_data({"lq": T(fix[f"it{it}/lq"]), "gt": T(fix[f"it{it}/gt"])})
        tr.optimize_parameters()
        assert list(tr.log.keys()) == keys
        for j, k in enumerate(keys):
            ref = fix["log"][it - 1, j]
            assert abs(tr.log[k] - ref) < 1e-4 * max(abs(ref), 1e-2), (it, k, tr.log[k], ref)
        assert rel_err(tr.output, T(fix[f"it{it}/out"])) < 1e-4
    check_final(fix, tr.G, tr.D or {}, 1e-3)  # north_star tolerance; see werr() on the near-zero biases


def test_swinir_medium_oracle_forward_backward_vs_reference():
    """configs[3]'s generator as named: seeded init + seeded perturbation rebuilt here, then y, dx and every
    parameter gradient (checksums for all 550, full tensors for a sample) against the reference run."""
    from neosr_amd.archs import swinir_arch as A

    fix = load_golden("cfg3_swinir_medium.npz")
    seed = int(fix["seed"])
    torch.manual_seed(seed)
    net = A.swinir_medium(upscale=4, drop_path_rate=0.0)
    sgen = torch.Generator().manual_seed(7000 + seed)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
    P = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in net.named_parameters())
    keys = [str(k) for k in fix["p/keys"]]
    assert keys == list(P)
    np.testing.assert_allclose(np.array([float(P[k].double().sum()) for k in keys]), fix["p/sum"], rtol=1e-6, atol=1e-5)
    x = T(fix["x"]).requires_grad_(True)
    cfg = dict(sorc.VARIANTS["swinir_medium"], upscale=4)
    y = sorc.swinir_forward(P, x, **cfg)
    assert rel_err(y, T(fix["y"])) < 1e-5
    (y * T(fix["r"])).sum().backward()
    assert rel_err(x.grad, T(fix["gx"])) < 1e-4
    l2 = np.array([float(P[k].grad.double().norm()) for k in keys])
    np.testing.assert_allclose(l2, fix["g/l2"], rtol=1e-4, atol=1e-7)
    for k in [f for f in fix if f.startswith("gfull/")]:
        assert rel_err(P[k[len("gfull/"):]].grad, T(fix[k])) < 1e-4, k


@pytest.mark.skipif(torch.get_num_threads() < 4, reason="hat_l forward + backward on CPU wants a few threads")
def test_hat_l_oracle_forward_backward_vs_reference():
    """configs[4]'s generator as named (hat_l, train mode, drop_path_rate 0): seeded init + seeded perturbation rebuilt
    here, then y, dx and every parameter gradient (checksums for all 1710, full tensors for a sample) against the
    reference run (cfg4_hat_l.npz, gen_golden_cfgs.py hat_l)."""
    from neosr_amd.archs import hat_arch as A

    fix = load_golden("cfg4_hat_l.npz")
    seed = int(fix["seed"])
    torch.manual_seed(seed)
    net = A.hat_l(upscale=4, drop_path_rate=0.0)
    sgen = torch.Generator().manual_seed(9000 + seed)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
    P = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in net.named_parameters())
    keys = [str(k) for k in fix["p/keys"]]
    assert keys == list(P)
    np.testing.assert_allclose(np.array([float(P[k].double().sum()) for k in keys]), fix["p/sum"], rtol=1e-6, atol=1e-5)
    x = T(fix["x"]).requires_grad_(True)
    cfg = {k: v for k, v in horc.VARIANTS["hat_l"].items() if k not in ("compress_ratio", "squeeze_factor")}
    y = horc.hat_forward(P, x, upscale=4, **cfg)
    assert rel_err(y, T(fix["y"])) < 1e-5
    (y * T(fix["r"])).sum().backward()
    assert rel_err(x.grad, T(fix["gx"])) < 1e-4
    l2 = np.array([float(P[k].grad.double().norm()) for k in keys])
    np.testing.assert_allclose(l2, fix["g/l2"], rtol=1e-4, atol=1e-7)
    for k in [f for f in fix if f.startswith("gfull/")]:
        assert rel_err(P[k[len("gfull/"):]].grad, T(fix[k])) < 1e-4, k
