"""GPU parity of the GAN / perceptual branch (HIP kernels through the C ABI) against fixtures produced
by the reference: U-Net-SN discriminator incl. spectral-norm state evolution, VGG19 taps, perceptual /
chc / BCE-GAN losses, and the 2-iteration GAN training trajectory of the `image` model."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, ROOT, group, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def prims():
    return load_golden("gan_prims.npz")


def G(a):
    return torch.from_numpy(np.array(a)).to(DEV)


def _load_vgg(vgg_module):
    from oracle import gan_oracle as gorc

    sd = {f"vgg_net.{k}": v for k, v in gorc.vgg_seeded_weights().items()}
    missing = vgg_module.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"mean", "std"}


def test_layer_kernels_vs_torch_cpu():
    """space-to-depth conv4x4s2, bilinear x2 (+adjoint), max-pool (+grad routing) vs ATen on CPU."""
    import torch.nn.functional as F

    from neosr_amd.hip import layers as L

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 12, 20, generator=g).requires_grad_(True)
    w = (torch.randn(16, 8, 4, 4, generator=g) * 0.1).requires_grad_(True)
    y = F.leaky_relu(F.conv2d(x, w, None, stride=2, padding=1), 0.2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    yd = L.conv4x4s2(xd, wd, None, L.ACT_LRELU, 0.2)
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert rel_err(yd.permute(0, 3, 1, 2), y) < 1e-5
    assert rel_err(xd.grad.permute(0, 3, 1, 2), x.grad) < 1e-5
    assert rel_err(wd.grad, w.grad) < 1e-5
    for fn, ref_fn in ((L.BilinearUp2.apply, lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)),
                       (L.MaxPool2.apply, lambda t: F.max_pool2d(t, 2, 2))):
        a = torch.randn(2, 8, 6, 10, generator=g).requires_grad_(True)
        r = ref_fn(a)
        gr = torch.randn(r.shape, generator=g)
        r.backward(gr)
        ad = a.detach().permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
        rd = fn(ad)
        rd.backward(gr.permute(0, 2, 3, 1).contiguous().to(DEV))
        assert rel_err(rd.permute(0, 3, 1, 2), r) < 1e-6
        assert rel_err(ad.grad.permute(0, 3, 1, 2), a.grad) < 1e-6


def test_unet_sn_vs_reference_fixture(prims):
    from neosr_amd.archs import build_network

    d = build_network({"type": "unet", "num_in_ch": 3, "num_feat": 8})
    d.load_state_dict(group(prims, "unet_sd0"))
    d = d.to(DEV).train()
    x = G(prims["unet_x"]).requires_grad_(True)
    y = d(x)
    assert rel_err(y, torch.from_numpy(prims["unet_y"])) < 1e-4
    (y * G(prims["unet_r"])).sum().backward()
    assert rel_err(x.grad, torch.from_numpy(prims["unet_gx"])) < 1e-3
    named = dict(d.named_parameters())
    worst = max(rel_err(named[k].grad, g) for k, g in group(prims, "unet_grad").items())
    assert worst < 1e-3, worst
    sd = d.state_dict()
    for k, v in group(prims, "unet_sd1").items():
        assert rel_err(sd[k], v) < 1e-5, k        # u/v advanced by exactly one power iteration
    y2 = d(x.detach())
    assert rel_err(y2, torch.from_numpy(prims["unet_y2"])) < 1e-4
    d.eval()
    with torch.no_grad():
        assert rel_err(d(x.detach()), torch.from_numpy(prims["unet_y_eval"])) < 1e-4


def test_gan_and_chc_losses_vs_reference_fixture(prims):
    from neosr_amd.losses import build_loss

    logits = G(prims["gan_logits"])
    gl = build_loss({"type": "gan_loss", "gan_type": "bce", "loss_weight": 0.3})
    for real in (True, False):
        for disc in (True, False):
            t = logits.clone().requires_grad_(True)
            v = gl(t, target_is_real=real, is_disc=disc)
            v.backward()
            tag = f"gan_{int(real)}{int(disc)}"
            assert abs(v.item() - float(prims[tag])) < 1e-5 * abs(float(prims[tag]))
            assert rel_err(t.grad, torch.from_numpy(prims[tag + "_g"])) < 1e-5
            assert abs(gl.last_mean.item() - float(prims["gan_logits"].mean())) < 1e-5
    a, b = G(prims["chc_a"]), G(prims["chc_b"])
    for crit in ("huber", "l1"):
        t = a.clone().requires_grad_(True)
        v = build_loss({"type": "chc_loss", "loss_weight": 0.8, "criterion": crit})(t, b)
        v.backward()
        assert abs(v.item() - float(prims[f"chc_{crit}"])) < 1e-5 * float(prims[f"chc_{crit}"])
        assert rel_err(t.grad, torch.from_numpy(prims[f"chc_{crit}_g"])) < 1e-5


def test_vgg_taps_and_perceptual_loss_vs_reference_fixture(prims):
    from neosr_amd.losses import build_loss

    pl = build_loss({"type": "vgg_perceptual_loss", "loss_weight": 0.5, "criterion": "chc"})
    _load_vgg(pl.vgg)
    pl = pl.to(DEV)
    x = G(prims["vgg_x"]).requires_grad_(True)
    feats = pl.vgg(x)
    for k, f in group(prims, "vgg_feat").items():
        assert tuple(feats[k].shape) == tuple(f.shape)
        assert rel_err(feats[k], f) < 1e-4, k
    v = pl(x, G(prims["vgg_gt"]))
    v.backward()
    assert abs(v.item() - float(prims["percep"])) < 1e-4 * float(prims["percep"])
    assert rel_err(x.grad, torch.from_numpy(prims["percep_gx"])) < 1e-3


def test_image_model_gan_trajectory_vs_reference_fixture():
    """OUR `image` model (esrgan G + unet D + L1 + perceptual + GAN) from the reference's initial
    weights / batches: every log_dict entry, the outputs and the final G, D weights and SN buffers."""
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options

    fix = load_golden("step_gan.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_gan.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init_g"))
    model.net_d.load_state_dict(group(fix, "init_d"))
    _load_vgg(model.cri_perceptual.vgg)
    keys = [str(k) for k in fix["log_keys"]]
    for it in (1, 2):
        model.feed_data({"lq": torch.from_numpy(fix[f"lq{it}"]), "gt": torch.from_numpy(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert list(log.keys()) == keys
        for j, k in enumerate(keys):
            ref = fix["log"][it - 1, j]
            assert abs(log[k] - ref) < 1e-3 * max(abs(ref), 1e-3), (it, k, log[k], ref)
        assert rel_err(model.output, torch.from_numpy(fix[f"out{it}"])) < 1e-3
    gsd, dsd = model.net_g.state_dict(), model.net_d.state_dict()
    assert max(rel_err(gsd[k], v) for k, v in group(fix, "final_g").items()) < 1e-3
    worst = {k: rel_err(dsd[k], v) for k, v in group(fix, "final_d").items()}
    assert max(worst.values()) < 1e-3, max(worst, key=worst.get)


@pytest.mark.parametrize("criterion", ["l1", "l2", "huber"])
def test_perceptual_loss_other_criteria_vs_oracle(prims, criterion):
    """criterion(f(x)/10, f(gt)/10) with nn.L1Loss / nn.MSELoss / nn.HuberLoss (vgg_perceptual_loss.py:135-141,232-236)"""
    import torch.nn.functional as F

    from neosr_amd.losses import build_loss
    from oracle import gan_oracle as gorc

    pl = build_loss({"type": "vgg_perceptual_loss", "loss_weight": 0.5, "criterion": criterion})
    _load_vgg(pl.vgg)
    pl = pl.to(DEV)
    x = G(prims["vgg_x"]).requires_grad_(True)
    v = pl(x, G(prims["vgg_gt"]))
    v.backward()
    vggP = gorc.vgg_seeded_weights()
    xr = torch.from_numpy(prims["vgg_x"]).requires_grad_(True)
    fx = gorc.vgg_features(vggP, xr)
    with torch.no_grad():
        fg = gorc.vgg_features(vggP, torch.from_numpy(prims["vgg_gt"]))
    fn = {"l1": F.l1_loss, "l2": F.mse_loss, "huber": F.huber_loss}[criterion]
    ref = sum(fn(fx[k] / 10, fg[k] / 10) * w for k, w in gorc.DEFAULT_LAYER_WEIGHTS.items()) * 0.5
    ref.backward()
    assert abs(v.item() - float(ref)) < 1e-4 * float(ref)
    assert rel_err(x.grad, xr.grad) < 1e-3


def test_unet_fused_skip_backward_is_bit_identical(monkeypatch):
    """x0 / x1 / x2 of the U-Net: depth-to-space + skip gradient + LeakyReLU derivative in one backward pass
    (layers.SpaceToDepth2Skip, neosr_depth_to_space2_fused) against the separate passes (autograd's sum, then the
    producer's neosr_leaky_relu pass): the same expressions, so every gradient must agree bit for bit."""
    from neosr_amd.archs import build_network
    from neosr_amd.hip import layers as L

    torch.manual_seed(3)
    d = build_network({"type": "unet", "num_in_ch": 3, "num_feat": 16}).to(DEV).train()
    x = torch.rand(2, 3, 64, 48, device=DEV, generator=None)
    r = torch.randn(2, 1, 64, 48, device=DEV)

    def run():
        d.zero_grad(set_to_none=True)
        for m in d.modules():   # same power-iteration state for both runs
            if hasattr(m, "weight_u"):
                m.weight_u.data.copy_(state[id(m)][0])
                m.weight_v.data.copy_(state[id(m)][1])
        xi = x.clone().requires_grad_(True)
        y = d(xi)
        (y * r).sum().backward()
        return y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in d.parameters()]

    state = {id(m): (m.weight_u.data.clone(), m.weight_v.data.clone()) for m in d.modules() if hasattr(m, "weight_u")}
    y1, gx1, g1 = run()
    monkeypatch.setattr(L, "conv4x4s2_skip", lambda x, xs, w, b=None, act=L.ACT_NONE, slope=0.0: (L.conv4x4s2(x, w, b, act, slope), x))
    y0, gx0, g0 = run()
    assert torch.equal(y0, y1)
    assert torch.equal(gx0, gx1)
    assert all(torch.equal(a, b) for a, b in zip(g0, g1))
