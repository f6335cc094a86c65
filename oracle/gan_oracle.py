"""CPU oracle for the GAN / perceptual branch of the neosr hot path — TEST INFRASTRUCTURE ONLY.

PyTorch-CPU fp32 restatement of: the U-Net-SN discriminator incl. the spectral-norm power
iteration (neosr/archs/unet_arch.py:9-67 + torch.nn.utils.spectral_norm), the VGG19 tap extractor
(neosr/archs/vgg_arch.py:159-199), vgg_perceptual_loss (losses/vgg_perceptual_loss.py:204-242),
chc_loss (losses/basic_loss.py:192-219), gan_loss (losses/gan_loss.py:45-82) and the GAN branch of
`image.closure` (models/image.py:427-625).

Parity status: PINNED against tests/golden/gan_prims.npz and step_gan.npz (reference run on CPU,
tests/golden/gen_golden_gan.py) — except that VGG19 uses seeded random weights (ImageNet weights
are not available offline): "parity unpinned" for the real perceptual weights.
"""

from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle import neosr_oracle as orc

# ------------------------------------------------------------------------------------ spectral norm


def spectral_normalize(P, name: str, training: bool, eps: float = 1e-12) -> torch.Tensor:
    """weight = weight_orig / sigma; one power iteration updating weight_u / weight_v IN PLACE in
    train mode (torch/nn/utils/spectral_norm.py compute_weight)."""
    w = P[f"{name}.weight_orig"]
    u, v = P[f"{name}.weight_u"], P[f"{name}.weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    uu, vv = u.clone(), v.clone()
    sigma = torch.dot(uu, torch.mv(wm, vv))
    return w / sigma


def unet_forward(P, x, training: bool = True, skip: bool = True):
    """unet.forward (unet_arch.py:36-67)."""
    lr = lambda t: F.leaky_relu(t, 0.2)  # noqa: E731
    sn = lambda n: spectral_normalize(P, n, training)  # noqa: E731
    x0 = lr(F.conv2d(x, P["conv0.weight"], P["conv0.bias"], padding=1))
    x1 = lr(F.conv2d(x0, sn("conv1"), None, stride=2, padding=1))
    x2 = lr(F.conv2d(x1, sn("conv2"), None, stride=2, padding=1))
    x3 = lr(F.conv2d(x2, sn("conv3"), None, stride=2, padding=1))
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)  # noqa: E731
    x4 = lr(F.conv2d(up(x3), sn("conv4"), None, padding=1))
    if skip:
        x4 = x4 + x2
    x5 = lr(F.conv2d(up(x4), sn("conv5"), None, padding=1))
    if skip:
        x5 = x5 + x1
    x6 = lr(F.conv2d(up(x5), sn("conv6"), None, padding=1))
    if skip:
        x6 = x6 + x0
    out = lr(F.conv2d(x6, sn("conv7"), None, padding=1))
    out = lr(F.conv2d(out, sn("conv8"), None, padding=1))
    return F.conv2d(out, P["conv9.weight"], P["conv9.bias"], padding=1)


# ------------------------------------------------------------------------------------ VGG / losses

_VGG_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]
DEFAULT_LAYER_WEIGHTS = {"conv1_2": 0.1, "conv2_2": 0.1, "conv3_4": 1.0, "conv4_4": 1.0, "conv5_4": 1.0}


def vgg_seeded_weights(seed: int = 77) -> "OrderedDict[str, torch.Tensor]":
    """the fixture's VGG19 weights: seeded He-normal draw in layer order (gen_golden_gan.seed_vgg_)."""
    g = torch.Generator().manual_seed(seed)
    P, c, blk, j = OrderedDict(), 3, 1, 1
    for v in _VGG_CFG:
        if v == "M":
            blk, j = blk + 1, 1
            continue
        P[f"conv{blk}_{j}.weight"] = torch.randn((v, c, 3, 3), generator=g) * (2.0 / (c * 9)) ** 0.5
        P[f"conv{blk}_{j}.bias"] = torch.randn((v,), generator=g) * 0.01
        c, j = v, j + 1
    return P


def vgg_features(P, x, taps=tuple(DEFAULT_LAYER_WEIGHTS)):
    """VGGFeatureExtractor.forward: (x-0.5)/0.25, conv/ReLU/pool stack, taps BEFORE the ReLU."""
    x = (x - 0.5) / 0.25
    out, blk, j = {}, 1, 1
    for v in _VGG_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            blk, j = blk + 1, 1
            continue
        name = f"conv{blk}_{j}"
        x = F.conv2d(x, P[f"{name}.weight"], P[f"{name}.bias"], padding=1)
        if name in taps:
            out[name] = x.clone()
        x = F.relu(x)
        j += 1
    return out


def chc_loss(pred, target, loss_weight=1.0, criterion="huber", clip_min=0.003921, clip_max=0.996078,
             loss_lambda=0.0):
    """basic_loss.py:192-219."""
    cos = (1 - F.cosine_similarity(pred, target, dim=1, eps=1e-20)).mean()
    t = torch.abs(pred - target) if criterion == "l1" else torch.sqrt((pred - target) ** 2 + 1e-12)
    return loss_weight * torch.mean(torch.clamp(t + loss_lambda * cos, clip_min, clip_max))


def perceptual_loss(vggP, x, gt, loss_weight=1.0, layer_weights=DEFAULT_LAYER_WEIGHTS):
    """vgg_perceptual_loss.forward, non-patch path (:218-242)."""
    fx = vgg_features(vggP, x, tuple(layer_weights))
    with torch.no_grad():
        fg = vgg_features(vggP, gt.detach(), tuple(layer_weights))
    total = 0.0
    for k in fx:
        total = total + chc_loss(fx[k] / 10, fg[k] / 10, 1.0, "huber", 0, 1, 0) * layer_weights[k]
    return total * loss_weight


def gan_loss(net_output, target_is_real: bool, is_disc: bool = False, loss_weight: float = 0.1,
             real_label_val: float = 1.0, fake_label_val: float = 0.0):
    """gan_loss('bce').forward (gan_loss.py:59-82)."""
    t = torch.full_like(net_output, real_label_val if target_is_real else fake_label_val)
    loss = F.binary_cross_entropy_with_logits(net_output, t)
    return loss if is_disc else loss * loss_weight


# ------------------------------------------------------------------------------------ GAN training step


class GanTrainer:
    """`image.optimize_parameters` with a discriminator (image.py:427-662): G phase with D frozen
    (pixel + perceptual + GAN), clip, D phase (real and fake forwards, then both backwards), clip,
    AdamW steps, EMA.  Generator = RRDBNet, discriminator = U-Net-SN, as in BASELINE configs[2]."""

    def __init__(self, g_params, d_params, vggP, *, lr_g, lr_d, betas=(0.9, 0.99), weight_decay=0.01,
                 pix_w=1.0, percep_w=0.5, gan_w=0.3, ema=0.999) -> None:
        self.G = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in g_params.items())
        self.D = OrderedDict((k, v.clone()) for k, v in d_params.items())
        self.d_train = [k for k in self.D if not k.endswith(("_u", "_v"))]
        for k in self.d_train:
            self.D[k].requires_grad_(True)
        self.vggP = vggP
        self.cfg = dict(lr_g=lr_g, lr_d=lr_d, betas=betas, wd=weight_decay, pix=pix_w, per=percep_w,
                        gan=gan_w, ema=ema)
        self.mg = [torch.zeros_like(v) for v in self.G.values()]
        self.vg = [torch.zeros_like(v) for v in self.G.values()]
        self.md = [torch.zeros_like(self.D[k]) for k in self.d_train]
        self.vd = [torch.zeros_like(self.D[k]) for k in self.d_train]
        self.ema = [v.detach().clone() for v in self.G.values()]
        self.step = 0
        self.log: dict[str, float] = {}

    def optimize_parameters(self, lq, gt) -> None:
        c = self.cfg
        out = orc.rrdbnet_forward(self.G, lq, 4)
        # ---- generator phase (D parameters frozen, but u/v still advance: SURVEY App. B-17a)
        Dfrozen = OrderedDict((k, v.detach()) if k in self.d_train else (k, v) for k, v in self.D.items())
        l_pix = orc.l1_loss(out, gt, c["pix"])
        l_per = perceptual_loss(self.vggP, out, gt, c["per"])
        l_gan = gan_loss(unet_forward(Dfrozen, out, True), True, False, c["gan"])
        l_g_total = torch.zeros(1) + l_pix + l_per + l_gan
        g_grads = [g.clone() for g in torch.autograd.grad(l_g_total.sum(), list(self.G.values()))]
        orc.clip_grad_norm_(g_grads, 1.0)
        # ---- discriminator phase: both forwards, then both backwards
        real_pred = unet_forward(self.D, gt, True)
        l_d_real = gan_loss(real_pred, True, True)
        fake_pred = unet_forward(self.D, out.detach(), True)
        l_d_fake = gan_loss(fake_pred, False, True)
        dpar = [self.D[k] for k in self.d_train]
        d_grads = [a + b for a, b in zip(torch.autograd.grad(l_d_real, dpar, retain_graph=True),
                                         torch.autograd.grad(l_d_fake, dpar))]
        orc.clip_grad_norm_(d_grads, 1.0)
        self.step += 1
        with torch.no_grad():
            gp = list(self.G.values())
            orc.adamw_step(gp, g_grads, self.mg, self.vg, self.step, c["lr_g"], c["betas"], 1e-8, c["wd"])
            orc.adamw_step(dpar, d_grads, self.md, self.vd, self.step, c["lr_d"], c["betas"], 1e-8, c["wd"])
            orc.ema_update(self.ema, gp, c["ema"], first=self.step == 1)
        self.output = out.detach()
        self.log = {"l_g_pix": float(l_pix), "l_g_percep": float(l_per), "l_g_gan": float(l_gan),
                    "l_g_total": float(l_g_total), "l_d_real": float(l_d_real),
                    "out_d_real": float(real_pred.mean()), "l_d_fake": float(l_d_fake),
                    "out_d_fake": float(fake_pred.mean()), "l_d_total": float((l_d_real + l_d_fake) / 2)}
