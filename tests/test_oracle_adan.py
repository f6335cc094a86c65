"""Pins the oracle's Schedule-Free Adan (oracle/neosr_oracle.py:AdanSF, AdanImageTrainer) to fixtures
produced by the reference's `adan_sf` optimizer and its `image` model trajectory.  CPU only."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import neosr_oracle as orc
from tests.conftest import group, load_golden, rel_err


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("tag", ["sf", "plain"])
def test_adan_sf_steps_and_mode_switch(tag):
    fix = load_golden("adan_sf.npz")
    ps = [T(fix[f"{tag}/p0/{i}"]).clone() for i in range(2)]
    opt = orc.AdanSF(ps, lr=2e-3, betas=(0.98, 0.92, 0.987), weight_decay=0.02, warmup_steps=3,
                     schedule_free=tag == "sf")
    for step in range(1, 6):
        opt.step([T(fix[f"{tag}/g{step}/{i}"]) for i in range(2)])
        for i in range(2):
            assert rel_err(ps[i], T(fix[f"{tag}/p{step}/{i}"])) < 1e-6, (step, i)
        if tag == "sf" and step == 3:
            opt.eval()
            assert all(rel_err(ps[i], T(fix[f"{tag}/p_eval/{i}"])) < 1e-6 for i in range(2))
            opt.train()
            assert all(rel_err(ps[i], T(fix[f"{tag}/p_train/{i}"])) < 1e-6 for i in range(2))
    for i in range(2):
        for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"):
            assert rel_err(opt.state[i][k], T(fix[f"{tag}/state/{k}/{i}"])) < 1e-5, k
    stp, wsum, lrmax = fix[f"{tag}/group"]
    assert opt.step_n == int(stp)
    if tag == "sf":
        assert abs(opt.weight_sum - wsum) < 1e-12 and abs(opt.lr_max - lrmax) < 1e-12


def test_image_trajectory_with_adan_sf():
    fix = load_golden("step_adan.npz")
    fwd = lambda P, x: orc.rrdbnet_forward(P, x, 4)  # noqa: E731
    tr = orc.AdanImageTrainer(fwd, group(fix, "init"), lr=8e-4, betas=(0.98, 0.92, 0.987), weight_decay=0.02,
                              warmup_steps=3)
    for it in range(1, 5):
        tr.feed_data(T(fix[f"lq{it}"]), T(fix[f"gt{it}"]))
        tr.optimize_parameters()
        assert abs(tr.log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-5
        assert rel_err(tr.output, T(fix[f"out{it}"])) < 1e-4
        if it == 2:
            tr.opt.eval()
            assert max(rel_err(tr.P[k], v) for k, v in group(fix, "eval2").items()) < 1e-5
            tr.opt.train()
    assert max(rel_err(tr.P[k], v) for k, v in group(fix, "final").items()) < 1e-4
    ema = dict(zip(tr.names, tr.ema))
    assert max(rel_err(ema[k.removeprefix("module.")], v) for k, v in group(fix, "ema").items()
               if k != "n_averaged") < 1e-4
