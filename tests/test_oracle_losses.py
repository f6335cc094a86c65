"""Pins oracle/loss_oracle.py (mssim_loss) to tests/golden/mssim.npz (reference run).  CPU only."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo
from tests.conftest import load_golden, rel_err


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mssim_loss_value_and_gradient(tag):
    fix = load_golden("mssim.npz")
    x = T(fix[f"{tag}/x"]).requires_grad_(True)
    loss = lo.mssim_loss(x, T(fix[f"{tag}/gt"]), loss_weight=float(fix[f"{tag}/loss_weight"]))
    loss.backward()
    assert abs(float(loss) - float(fix[f"{tag}/loss"])) < 1e-6
    assert rel_err(x.grad, T(fix[f"{tag}/gx"])) < 1e-5


@pytest.mark.parametrize("tag", ["near", "far"])
def test_consistency_loss_value_and_gradient(tag):
    fix = load_golden("consistency.npz")
    x = T(fix[f"{tag}/x"]).requires_grad_(True)
    loss = lo.consistency_loss(x, T(fix[f"{tag}/gt"]), saturation=1.1, brightness=0.95, loss_weight=0.8)
    loss.backward()
    assert abs(float(loss) - float(fix[f"{tag}/loss"])) < 1e-6
    assert rel_err(x.grad, T(fix[f"{tag}/gx"])) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b"])
def test_chc_loss_with_cosine_term_oracle_vs_reference(tag):
    """oracle/gan_oracle.chc_loss (basic_loss.py:192-219) for loss_lambda in {0, 5/255, 0.5}, both criteria"""
    from oracle import gan_oracle as gorc

    fix = load_golden("chc_lambda.npz")
    for crit in ("l1", "huber"):
        for lam in (0.0, 5 / 255, 0.5):
            x = T(fix[f"{tag}/x"]).requires_grad_(True)
            v = gorc.chc_loss(x, T(fix[f"{tag}/y"]), 0.8, crit, loss_lambda=lam)
            (v * 1.7).backward()
            key = f"{tag}/chc/{crit}_{lam:.6f}"
            assert abs(float(v) - float(fix[key])) < 1e-6, key
            assert rel_err(x.grad, T(fix[key + "/g"])) < 1e-5, key
