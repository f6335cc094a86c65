"""Pins oracle/gan_oracle.py to fixtures produced by the reference's unet / spectral_norm / VGG /
perceptual / chc / gan_loss code and its 2-iteration GAN training step.  CPU only."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import gan_oracle as gorc
from tests.conftest import group, load_golden, rel_err


@pytest.fixture(scope="module")
def prims():
    return load_golden("gan_prims.npz")


def T(a):
    return torch.from_numpy(np.array(a))


def test_unet_sn_forward_backward_and_uv_evolution(prims):
    P = group(prims, "unet_sd0")
    train = [k for k in P if not k.endswith(("_u", "_v"))]
    for k in train:
        P[k].requires_grad_(True)
    x = T(prims["unet_x"]).requires_grad_(True)
    y = gorc.unet_forward(P, x, training=True)
    assert rel_err(y, T(prims["unet_y"])) < 1e-5
    (y * T(prims["unet_r"])).sum().backward()
    assert rel_err(x.grad, T(prims["unet_gx"])) < 1e-4
    for k, g in group(prims, "unet_grad").items():
        assert rel_err(P[k].grad, g) < 1e-4, k
    for k, v in group(prims, "unet_sd1").items():     # u / v after ONE train-mode forward
        assert rel_err(P[k], v) < 1e-5, k
    y2 = gorc.unet_forward(P, x.detach(), training=True)
    assert rel_err(y2, T(prims["unet_y2"])) < 1e-5
    y3 = gorc.unet_forward(P, x.detach(), training=False)  # eval: no power iteration
    assert rel_err(y3, T(prims["unet_y_eval"])) < 1e-5


def test_gan_and_chc_losses(prims):
    logits = T(prims["gan_logits"])
    for real in (True, False):
        for disc in (True, False):
            t = logits.clone().requires_grad_(True)
            v = gorc.gan_loss(t, real, disc, loss_weight=0.3)
            v.backward()
            tag = f"gan_{int(real)}{int(disc)}"
            assert abs(float(v) - float(prims[tag])) < 1e-6
            assert rel_err(t.grad, T(prims[tag + "_g"])) < 1e-6
    a, b = T(prims["chc_a"]), T(prims["chc_b"])
    for crit in ("huber", "l1"):
        t = a.clone().requires_grad_(True)
        v = gorc.chc_loss(t, b, 0.8, crit)
        v.backward()
        assert abs(float(v) - float(prims[f"chc_{crit}"])) < 1e-6
        assert rel_err(t.grad, T(prims[f"chc_{crit}_g"])) < 1e-6


def test_vgg_taps_and_perceptual_loss(prims):
    P = gorc.vgg_seeded_weights()
    x = T(prims["vgg_x"]).requires_grad_(True)
    feats = gorc.vgg_features(P, x)
    for k, f in group(prims, "vgg_feat").items():
        assert rel_err(feats[k], f) < 1e-5, k
    v = gorc.perceptual_loss(P, x, T(prims["vgg_gt"]), 0.5)
    v.backward()
    assert abs(float(v) - float(prims["percep"])) < 1e-5 * float(prims["percep"])
    assert rel_err(x.grad, T(prims["percep_gx"])) < 1e-4


def test_gan_training_step_trajectory():
    fix = load_golden("step_gan.npz")
    keys = [str(k) for k in fix["log_keys"]]
    tr = gorc.GanTrainer(group(fix, "init_g"), group(fix, "init_d"), gorc.vgg_seeded_weights(),
                         lr_g=1e-3, lr_d=5e-4, betas=(0.9, 0.99), percep_w=0.5, gan_w=0.3)
    for it in (1, 2):
        tr.optimize_parameters(T(fix[f"lq{it}"]), T(fix[f"gt{it}"]))
        for j, k in enumerate(keys):
            ref = fix["log"][it - 1, j]
            assert abs(tr.log[k] - ref) < 2e-4 * max(abs(ref), 1e-3), (it, k, tr.log[k], ref)
        assert rel_err(tr.output, T(fix[f"out{it}"])) < 1e-4
    for k, v in group(fix, "final_g").items():
        assert rel_err(tr.G[k], v) < 1e-3, k
    for k, v in group(fix, "final_d").items():
        assert rel_err(tr.D[k], v) < 1e-3, k
