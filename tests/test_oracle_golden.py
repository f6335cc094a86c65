"""Pins the CPU oracle (oracle/neosr_oracle.py) to fixtures produced by the reference itself
(tests/golden/gen_golden.py).  CPU-only; this is the "oracle is trustworthy" gate."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import neosr_oracle as orc
from tests.conftest import group, load_golden, rel_err

TOL = 1e-5  # same ATen CPU kernels under a re-stated graph: differences are pure fp32 re-association


def test_pixel_shuffle_index_bit_exact():
    fix = load_golden("index.npz")
    for r in (2, 4):
        assert np.array_equal(orc.pixel_shuffle_np(fix[f"ps{r}_in"], r), fix[f"ps{r}_out"])
        assert np.array_equal(orc.pixel_unshuffle_np(fix[f"pu{r}_in"], r), fix[f"pu{r}_out"])


def test_l1_loss_matches_reference():
    fix = load_golden("l1loss.npz")
    pred = torch.from_numpy(fix["pred"]).requires_grad_(True)
    out = orc.l1_loss(pred, torch.from_numpy(fix["target"]), 0.7)
    out.backward()
    assert abs(float(out) - float(fix["loss"])) <= 1e-6 * abs(float(fix["loss"]))
    assert rel_err(pred.grad, torch.from_numpy(fix["grad"])) < 1e-6


def _check_arch(fixname, fwd):
    fix = load_golden(fixname)
    P = group(fix, "param")
    for v in P.values():
        v.requires_grad_(True)
    x, gt = torch.from_numpy(fix["x"]), torch.from_numpy(fix["gt"])
    y = fwd(P, x)
    assert rel_err(y, torch.from_numpy(fix["y"])) < TOL
    loss = orc.l1_loss(y, gt)
    assert abs(float(loss) - float(fix["loss"])) < 1e-5 * abs(float(fix["loss"]))
    loss.backward()
    G = group(fix, "grad")
    for k, g in G.items():
        assert rel_err(P[k].grad, g) < 1e-4, k


def test_rrdbnet_matches_reference():
    _check_arch("esrgan_small.npz", lambda P, x: orc.rrdbnet_forward(P, x, scale=4))


@pytest.mark.parametrize("act", ["prelu", "leakyrelu", "relu"])
def test_compact_matches_reference(act):
    _check_arch(f"compact_small_{act}.npz", lambda P, x: orc.compact_forward(P, x, 4, act))


@pytest.mark.parametrize("arch", ["compact", "esrgan"])
def test_train_step_trajectory_matches_reference(arch):
    """3 iterations of image.optimize_parameters (L1 + AdamW + clip + EMA): losses, weights, EMA, Adam state."""
    fix = load_golden(f"step_{arch}.npz")
    init = group(fix, "init")
    fwd = (lambda P, x: orc.compact_forward(P, x, 4, "prelu")) if arch == "compact" else (
        lambda P, x: orc.rrdbnet_forward(P, x, 4))
    tr = orc.ImageTrainer(fwd, init, lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01, ema=0.999,
                          grad_clip=True)
    for it in range(1, 4):
        tr.feed_data(torch.from_numpy(fix[f"lq{it}"]), torch.from_numpy(fix[f"gt{it}"]))
        tr.optimize_parameters()
        assert abs(tr.log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-5 * abs(fix["log"][it - 1, 0])
        assert rel_err(tr.output, torch.from_numpy(fix[f"out{it}"])) < 1e-4
    final = group(fix, "final")
    ema = group(fix, "ema")
    for i, k in enumerate(tr.names):
        assert rel_err(tr.P[k], final[k]) < 1e-4, k
        assert rel_err(tr.ema[i], ema[f"module.{k}"]) < 1e-4, k
    for k, v in group(fix, "adam_exp_avg").items():
        assert rel_err(tr.m[tr.names.index(k)], v) < 1e-3, k
    for k, v in group(fix, "adam_exp_avg_sq").items():
        assert rel_err(tr.v[tr.names.index(k)], v) < 1e-3, k
