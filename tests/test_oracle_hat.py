"""Pins oracle/hat_oracle.py to fixtures produced by the reference's hat_arch.py (index / mask tables,
CAB / HAB / OCAB blocks and a tiny net forward + backward, hat_l forward) and checks that the product
arch reproduces the reference's state-dict layout and seeded initialisation.  CPU only."""

from __future__ import annotations

from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hat_oracle as ho
from oracle import swinir_oracle as so
from tests.conftest import group, load_golden, rel_err


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture(scope="module")
def prims():
    return load_golden("hat_prims.npz")


def test_index_and_mask_tables(prims):
    assert np.array_equal(ho.rpi_sa(16).numpy(), prims["rpi_sa_16"])
    assert np.array_equal(ho.rpi_oca(16, 0.5).numpy(), prims["rpi_oca_16"])
    assert prims["rpi_oca_16"].min() < 0  # the reference's negative (wrap-around) indices are part of the contract
    assert np.array_equal(so.calculate_mask(32, 48, 16, 8).numpy(), prims["mask_32x48_s8"])
    from neosr_amd.archs import hat_arch as A

    assert np.array_equal(A._rpi_sa(16).numpy(), prims["rpi_sa_16"])
    assert np.array_equal(A._rpi_oca(16, 0.5).numpy(), prims["rpi_oca_16"])


def _check_block(prims, pre, fn, tol=1e-4):
    P = group(prims, f"{pre}/p")
    for v in P.values():
        v.requires_grad_(True)
    x = T(prims[f"{pre}/x"]).requires_grad_(True)
    y = fn(OrderedDict((f"b.{k}", v) for k, v in P.items()), x)
    assert rel_err(y, T(prims[f"{pre}/y"])) < 1e-5
    (y * T(prims[f"{pre}/r"])).sum().backward()
    assert rel_err(x.grad, T(prims[f"{pre}/gx"])) < tol
    for k, g in group(prims, f"{pre}/g").items():
        assert rel_err(P[k].grad, g) < tol, k


def test_cab_forward_backward(prims):
    _check_block(prims, "cab", lambda P, x: ho.cab(P, "b", x))


@pytest.mark.parametrize("shift", [0, 8])
def test_hab_forward_backward(prims, shift):
    _check_block(prims, f"hab_s{shift}", lambda P, x: ho.hab(P, "b", x, (32, 48), 2, 16, shift, 0.01))


def test_ocab_forward_backward(prims):
    _check_block(prims, "ocab", lambda P, x: ho.ocab(P, "b", x, (32, 48), 2, 16, 0.5))


def test_tiny_hat_forward_backward():
    fix = load_golden("hat_net.npz")
    P = group(fix, "p")
    for v in P.values():
        v.requires_grad_(True)
    x = T(fix["x"]).requires_grad_(True)
    y = ho.hat_forward(P, x, depths=(2,), num_heads=(2,), embed_dim=24)
    assert rel_err(y, T(fix["y"])) < 1e-5
    (y * T(fix["r"])).sum().backward()
    assert rel_err(x.grad, T(fix["gx"])) < 1e-4
    for k, g in group(fix, "g").items():
        assert rel_err(P[k].grad, g) < 2e-4, k


def test_tiny_hat_window8_forward_backward():
    """window_size 8 (shift 4, 12x12 overlapping key windows), two RHAGs, 16x24 tokens: the oracle against the reference run"""
    fix = load_golden("hat_w8.npz")
    P = group(fix, "p")
    for v in P.values():
        v.requires_grad_(True)
    x = T(fix["x"]).requires_grad_(True)
    y = ho.hat_forward(P, x, depths=(2, 2), num_heads=(2, 2), embed_dim=24, window_size=8)
    assert rel_err(y, T(fix["y"])) < 1e-5
    (y * T(fix["r"])).sum().backward()
    assert rel_err(x.grad, T(fix["gx"])) < 1e-4
    for k, g in group(fix, "g").items():
        assert rel_err(P[k].grad, g) < 2e-4, k


def _sums(sd):
    return np.array([float(v.double().sum()) for v in sd.values()])


def test_product_arch_state_dict_and_seeded_init():
    from neosr_amd.archs import hat_arch as A
    from neosr_amd.utils import options

    fix = load_golden("hat_init.npz")
    options.set_global_opt({"manual_seed": 1024, "rank": 0, "scale": 4, "datasets": {"train": {}}})
    try:
        torch.manual_seed(1024)
        net = A.hat_s(upscale=4)
    finally:
        options.set_global_opt(None)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in fix["hat_s/keys"]]
    np.testing.assert_allclose(_sums(sd), fix["hat_s/sum"], rtol=1e-6, atol=1e-6)


@pytest.mark.skipif(torch.get_num_threads() < 4, reason="hat_l forward on CPU wants a few threads")
def test_hat_l_forward_b1():
    """the oracle on the product arch's seeded hat_l weights reproduces the reference's B=1 forward"""
    from neosr_amd.archs import hat_arch as A
    from neosr_amd.utils import options

    fix = load_golden("hat_l_fwd.npz")
    options.set_global_opt({"manual_seed": 1024, "rank": 0, "scale": 4, "datasets": {"train": {}}})
    try:
        torch.manual_seed(1024)
        net = A.hat_l(upscale=4)
    finally:
        options.set_global_opt(None)
    np.testing.assert_allclose(_sums(net.state_dict()), fix["init_sum"], rtol=1e-6, atol=1e-6)
    P = OrderedDict((k, v.detach()) for k, v in net.named_parameters())
    with torch.no_grad():
        y = ho.hat_forward(P, T(fix["x"]), **{k: v for k, v in ho.VARIANTS["hat_l"].items()
                                              if k in ("depths", "num_heads", "embed_dim")})
    assert rel_err(y, T(fix["y"])) < 1e-5
