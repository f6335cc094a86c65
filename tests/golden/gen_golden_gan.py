#!/usr/bin/env python
"""Golden fixtures for the GAN / perceptual branch, produced by RUNNING THE REFERENCE on CPU
(build container only):  python tests/golden/gen_golden_gan.py  ->  gan_prims.npz, step_gan.npz

  gan_prims.npz  unet (num_feat 8) forward/backward incl. spectral-norm u/v before/after two forwards;
                 VGGFeatureExtractor taps + vgg_perceptual_loss value/grad (VGG19 weights = the seeded
                 He draw of `seed_vgg_` below, because ImageNet weights cannot be downloaded);
                 gan_loss (bce) and chc_loss values/grads
  step_gan.npz   2 iterations of the reference `image` model: esrgan G + unet D + L1 + perceptual + GAN,
                 AdamW on both, clip, EMA: log_dict, outputs, final G / D weights and SN buffers
"""

from __future__ import annotations

import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import REF, install_reference, np_state, save  # noqa: E402

TOML = """
name = "golden_gan"
model_type = "image"
scale = 4
manual_seed = 1024

[datasets.train]
type = "paired"
dataroot_gt = "/tmp/none_gt"
dataroot_lq = "/tmp/none_lq"
patch_size = 16
batch_size = 2

[path]

[network_g]
type = "esrgan"
num_feat = 16
num_block = 1
num_grow_ch = 8

[network_d]
type = "unet"
num_feat = 8

[train]
ema = 0.999
grad_clip = true

[train.optim_g]
type = "adamw"
lr = 1e-3
betas = [ 0.9, 0.99 ]

[train.optim_d]
type = "adamw"
lr = 5e-4
betas = [ 0.9, 0.99 ]

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[train.perceptual_opt]
type = "vgg_perceptual_loss"
loss_weight = 0.5
criterion = "chc"

[train.gan_opt]
type = "gan_loss"
gan_type = "bce"
loss_weight = 0.3

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def seed_vgg_(vgg_module: torch.nn.Module, seed: int = 77) -> None:
    """Deterministic He-normal VGG weights shared by the reference run and the tests (CPU generator)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in vgg_module.modules():
            if isinstance(m, torch.nn.Conv2d):
                fan_in = m.weight.shape[1] * 9
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.01)


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_gan.toml"
    tmp.write_text(TOML)
    (HERE / "golden_gan.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.archs.unet_arch import unet
    from neosr.losses.basic_loss import chc_loss
    from neosr.losses.gan_loss import gan_loss
    from neosr.losses.vgg_perceptual_loss import vgg_perceptual_loss
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    gen = torch.Generator().manual_seed(11)
    A = {}
    # ---- unet with spectral norm
    torch.manual_seed(21)
    d = unet(num_in_ch=3, num_feat=8)
    d.train()
    for k, v in d.state_dict().items():
        A[f"unet_sd0/{k}"] = v.detach().numpy().copy()
    x = torch.rand(2, 3, 32, 48, generator=gen).requires_grad_(True)
    r = torch.randn(2, 1, 32, 48, generator=gen)
    y = d(x)
    (y * r).sum().backward()
    A["unet_x"], A["unet_r"], A["unet_y"] = x.detach().numpy(), r.numpy(), y.detach().numpy()
    A["unet_gx"] = x.grad.numpy().copy()
    for k, p in d.named_parameters():
        A[f"unet_grad/{k}"] = p.grad.numpy().copy()
    for k, v in d.state_dict().items():
        if k.endswith(("_u", "_v")):
            A[f"unet_sd1/{k}"] = v.detach().numpy().copy()
    y2 = d(x.detach())  # second train-mode forward: u/v advance again
    A["unet_y2"] = y2.detach().numpy()
    d.eval()
    A["unet_y_eval"] = d(x.detach()).detach().numpy()
    # ---- gan_loss / chc_loss
    logits = torch.randn(2, 1, 9, 13, generator=gen)
    gl = gan_loss(gan_type="bce", loss_weight=0.3)
    for real in (True, False):
        for disc in (True, False):
            t = logits.clone().requires_grad_(True)
            v = gl(t, target_is_real=real, is_disc=disc)
            v.backward()
            A[f"gan_{int(real)}{int(disc)}"] = np.float32(v.item())
            A[f"gan_{int(real)}{int(disc)}_g"] = t.grad.numpy().copy()
    A["gan_logits"] = logits.numpy()
    a = torch.rand(2, 3, 10, 12, generator=gen)
    b = (a + 0.3 * torch.randn(2, 3, 10, 12, generator=gen)).clamp(0, 1)
    A["chc_a"], A["chc_b"] = a.numpy(), b.numpy()
    for crit in ("huber", "l1"):
        t = a.clone().requires_grad_(True)
        v = chc_loss(loss_weight=0.8, criterion=crit)(t, b)
        v.backward()
        A[f"chc_{crit}"] = np.float32(v.item())
        A[f"chc_{crit}_g"] = t.grad.numpy().copy()
    # ---- VGG taps + perceptual loss (seeded random VGG19)
    pl = vgg_perceptual_loss(loss_weight=0.5, criterion="chc")
    seed_vgg_(pl.vgg)
    xv = torch.rand(2, 3, 32, 48, generator=gen).requires_grad_(True)
    gv = torch.rand(2, 3, 32, 48, generator=gen)
    feats = pl.vgg(xv)
    for k, f in feats.items():
        A[f"vgg_feat/{k}"] = f.detach().numpy().copy()
    v = pl(xv, gv)
    v.backward()
    A["vgg_x"], A["vgg_gt"] = xv.detach().numpy(), gv.numpy()
    A["percep"] = np.float32(float(v))
    A["percep_gx"] = xv.grad.numpy().copy()
    save("gan_prims.npz", **A)

    # ---- full GAN training step
    opt, _ = parse_options(str(REF), is_train=True)
    torch.manual_seed(1024)
    random.seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    seed_vgg_(model.cri_perceptual.vgg)
    S = {}
    for k, v in np_state(model.net_g.state_dict()).items():
        S[f"init_g/{k}"] = v
    for k, v in np_state(model.net_d.state_dict()).items():
        S[f"init_d/{k}"] = v
    logs, keys = [], None
    for it in (1, 2):
        lq = torch.rand(2, 3, 16, 16, generator=gen)
        gt = torch.rand(2, 3, 64, 64, generator=gen)
        S[f"lq{it}"], S[f"gt{it}"] = lq.numpy(), gt.numpy()
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        keys = list(log.keys())
        logs.append([log[k] for k in keys])
        S[f"out{it}"] = model.output.detach().numpy().copy()
    S["log"] = np.asarray(logs, dtype=np.float64)
    S["log_keys"] = np.array(keys)
    for k, v in np_state(model.net_g.state_dict()).items():
        S[f"final_g/{k}"] = v
    for k, v in np_state(model.net_d.state_dict()).items():
        S[f"final_d/{k}"] = v
    save("step_gan.npz", **S)


if __name__ == "__main__":
    main()
