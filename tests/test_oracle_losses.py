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
