"""Pins oracle/aug_oracle.py to tests/golden/aug.npz (the reference's augmentations with every random
draw recorded) by replaying the draws.  CPU only."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from neosr_amd.data.draws import ReplayDraws
from oracle import aug_oracle as ao
from tests.conftest import load_draws, load_golden, rel_err

AUGS = ["none", "mixup", "cutmix", "resizemix", "cutblur"]
PROB = [0.5, 0.1, 0.1, 0.1, 0.5]


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture(scope="module")
def fix():
    return load_golden("aug.npz")


@pytest.mark.parametrize("name", ["mixup", "cutmix", "resizemix", "cutblur"])
def test_single_augmentations(fix, name):
    d = ReplayDraws(load_draws(fix, f"fn/{name}/draws"))
    gt, lq = getattr(ao, name)(T(fix[f"fn/{name}/gt"]), T(fix[f"fn/{name}/lq"]), d)
    assert d.exhausted()
    assert rel_err(gt, T(fix[f"fn/{name}/gt_out"])) < 1e-6
    assert rel_err(lq, T(fix[f"fn/{name}/lq_out"])) < 1e-6


@pytest.mark.parametrize("k", range(16))
def test_apply_augment_replay(fix, k):
    d = ReplayDraws(load_draws(fix, f"run/{k}/draws"))
    gt, lq = ao.apply_augment(T(fix[f"run/{k}/gt"]), T(fix[f"run/{k}/lq"]), d, scale=4, augs=AUGS, prob=PROB)
    assert d.exhausted()
    assert rel_err(gt, T(fix[f"run/{k}/gt_out"])) < 1e-6
    assert rel_err(lq, T(fix[f"run/{k}/lq_out"])) < 1e-5
