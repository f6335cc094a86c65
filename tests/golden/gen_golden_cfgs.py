#!/usr/bin/env python
"""Config-shaped golden fixtures, produced by RUNNING THE REFERENCE on CPU (build container only):

    python tests/golden/gen_golden_cfgs.py swinir_medium   -> cfg3_swinir_medium.npz
    python tests/golden/gen_golden_cfgs.py hat_l           -> cfg4_hat_l.npz
    python tests/golden/gen_golden_cfgs.py cfg3            -> step_cfg3.npz   (+ golden_cfg3.toml)
    python tests/golden/gen_golden_cfgs.py cfg2            -> step_cfg2.npz   (+ golden_cfg2.toml)
    python tests/golden/gen_golden_cfgs.py cfg4            -> step_cfg4.npz   (+ golden_cfg4.toml)

  cfg3_swinir_medium.npz  BASELINE configs[3]'s generator AS NAMED: `swinir_medium()` (seeded init + a seeded
                          perturbation so that biases / LN affines are non-trivial), one 64x64 LR patch, forward
                          + backward of sum(y * r): y, dL/dx, per-parameter gradient checksums and a handful of
                          full gradient tensors.  drop_path_rate = 0 (DropPath draws from the global RNG).
  cfg4_hat_l.npz          BASELINE configs[4]'s generator AS NAMED: `hat_l()` in train mode, the same recipe (seeded init +
                          perturbation, one 64x64 LR patch, forward + backward of sum(y * r)): y, dL/dx, gradient
                          checksums of all 1710 parameters and a dozen full gradient tensors (HAB / OCAB / CAB / convs).
  step_cfg3.npz           configs[3]'s COMBINATION at reduced width: `image` model, swinir_small, L1 + VGG19
                          perceptual (seeded weights), adan_sf, clip, EMA — 2 iterations.
  step_cfg2.npz           configs[2]'s combination: `otf` model, feed_data with EVERY random draw recorded ->
                          esrgan G + U-Net-SN D + L1 + perceptual + GAN, adan_sf x 2 — 3 iterations (the pair
                          pool fills and shuffles).
  step_cfg4.npz           configs[4]'s combination: the same with network_g = hat_s.
Same shims / cuda->cpu redirect as gen_golden.py.  Data only; no reference source is copied.
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
from gen_golden_gan import seed_vgg_  # noqa: E402
from gen_golden_otf import DEG, Recorder, pack_draws, smooth_images  # noqa: E402

ADAN = """
[train.optim_{w}]
type = "adan_sf"
lr = {lr}
betas = [ 0.98, 0.92, {b3} ]
weight_decay = 0.02
schedule_free = true
warmup_steps = 4
"""

HEAD = """
name = "golden_{name}"
model_type = "{model}"
scale = 4
manual_seed = 1024
"""

LOSSES_GAN = """
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
"""

LOGGER = """
[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def toml_otf(name: str, net_g: str) -> str:
    return (HEAD.format(name=name, model="otf") + """
[datasets.train]
type = "otf"
dataroot_gt = "/tmp/none_gt"
patch_size = 16
batch_size = 2
queue_size = 4

[degradations]
""" + DEG + """
[path]

[network_g]
""" + net_g + """
[network_d]
type = "unet"
num_feat = 8

[train]
ema = 0.999
grad_clip = true
""" + ADAN.format(w="g", lr="8e-4", b3="0.987") + ADAN.format(w="d", lr="5e-4", b3="0.99") + LOSSES_GAN + LOGGER)


TOMLS = {
    "cfg2": toml_otf("cfg2", 'type = "esrgan"\nnum_feat = 16\nnum_block = 1\nnum_grow_ch = 8\n'),
    "cfg4": toml_otf("cfg4", 'type = "hat_s"\ndrop_path_rate = 0.0\n'),
    "cfg3": HEAD.format(name="cfg3", model="image") + """
[datasets.train]
type = "paired"
dataroot_gt = "/tmp/none_gt"
dataroot_lq = "/tmp/none_lq"
patch_size = 16
batch_size = 2

[path]

[network_g]
type = "swinir_small"
drop_path_rate = 0.0

[train]
ema = 0.999
grad_clip = true
""" + ADAN.format(w="g", lr="1e-3", b3="0.987") + """
[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[train.perceptual_opt]
type = "vgg_perceptual_loss"
loss_weight = 1.0
criterion = "chc"
""" + LOGGER,
}
TOMLS["swinir_medium"] = TOMLS["cfg3"]
TOMLS["hat_l"] = TOMLS["cfg4"]

SAMPLE_G = {  # a few full tensors of the big generators (the rest is pinned by per-tensor checksums)
    "swinir_small": ["conv_first.weight", "layers.0.residual_group.blocks.1.attn.relative_position_bias_table",
                     "layers.3.residual_group.blocks.5.mlp.fc2.weight", "upsample.0.weight", "norm.weight"],
    "hat_s": ["conv_first.weight", "layers.0.residual_group.blocks.1.attn.relative_position_bias_table",
              "layers.2.residual_group.overlap_attn.qkv.weight", "layers.5.residual_group.blocks.5.mlp.fc2.weight",
              "layers.1.residual_group.blocks.0.conv_block.cab.0.weight", "conv_last.weight", "norm.weight"],
}


def checksums(sd):
    keys = [k for k, v in sd.items() if v.is_floating_point()]
    s = np.array([float(sd[k].double().sum()) for k in keys])
    a = np.array([float(sd[k].double().abs().sum()) for k in keys])
    return np.array(keys), s, a


def gen_swinir_medium() -> None:
    from neosr.archs import swinir_arch as S

    for seed in range(1024, 1124):
        torch.manual_seed(seed)
        net = S.swinir_medium(drop_path_rate=0.0)
        sgen = torch.Generator().manual_seed(7000 + seed)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
        x = torch.rand(1, 3, 64, 64, generator=sgen).requires_grad_(True)
        closest = [float("inf")]
        hooks = [m.register_forward_pre_hook(lambda _m, a: closest.__setitem__(0, min(closest[0], float(a[0].abs().min()))))
                 for m in net.modules() if isinstance(m, torch.nn.LeakyReLU)]
        net.train()
        y = net(x)
        for h in hooks:
            h.remove()
        print(f"swinir_medium: seed {seed} closest LeakyReLU input to zero {closest[0]:.2e}")
        if closest[0] > 2e-7:  # see gen_golden_swinir.py: LeakyReLU' jumps at 0
            break
    else:
        raise RuntimeError("no well-conditioned draw found")
    r = torch.randn(y.shape, generator=sgen)
    (y * r).sum().backward()
    A = {"seed": np.int64(seed), "x": x.detach().numpy(), "r": r.numpy(), "y": y.detach().numpy(),
         "gx": x.grad.numpy().copy()}
    keys, s, a = checksums(dict(net.named_parameters()))
    A["p/keys"], A["p/sum"], A["p/abs"] = keys, s, a
    grads = {k: v.grad for k, v in net.named_parameters()}
    A["g/sum"] = np.array([float(grads[str(k)].double().sum()) for k in keys])
    A["g/abs"] = np.array([float(grads[str(k)].double().abs().sum()) for k in keys])
    A["g/l2"] = np.array([float(grads[str(k)].double().norm()) for k in keys])
    for k in ("conv_first.weight", "conv_first.bias", "layers.0.residual_group.blocks.0.attn.qkv.weight",
              "layers.0.residual_group.blocks.1.attn.relative_position_bias_table",
              "layers.2.residual_group.blocks.3.norm1.weight", "layers.3.residual_group.blocks.5.mlp.fc1.weight",
              "layers.5.residual_group.blocks.5.attn.proj.bias", "layers.5.conv.weight",
              "conv_before_upsample.0.weight", "upsample.2.bias", "conv_last.weight"):
        A[f"gfull/{k}"] = grads[k].numpy().copy()
    save("cfg3_swinir_medium.npz", **A)


def gen_hat_l() -> None:
    from neosr.archs import hat_arch as HA

    for seed in range(1024, 1124):
        torch.manual_seed(seed)
        net = HA.hat_l(drop_path_rate=0.0)
        sgen = torch.Generator().manual_seed(9000 + seed)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
        x = torch.rand(1, 3, 64, 64, generator=sgen).requires_grad_(True)
        # Two kinds of derivative jumps.  A (Leaky)ReLU over a feature MAP: one element near zero moves one element of
        # the gradients (2e-7 as in gen_golden_swinir.py).  The ReLU of a channel-attention bottleneck (input 1 x 1:
        # 6 hidden units per CAB, 72 CABs) gates a whole 180-channel map: a unit within rounding of zero on the
        # other side changes every gradient upstream by percents, so the draw must keep all 432 of them well away
        # from zero (1e-3 against values of ~0.13; fp32 summation-order noise there is ~1e-6).
        closest = [float("inf")]
        ca_pre = []

        def hook(_m, a):
            v = a[0]
            if v.shape[-2:] == (1, 1):
                ca_pre.append(v.detach().flatten().clone())
            else:
                closest[0] = min(closest[0], float(v.abs().min()))

        hooks = [m.register_forward_pre_hook(hook) for m in net.modules()
                 if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]
        net.train()
        y = net(x)
        for h in hooks:
            h.remove()
        ca_pre = torch.cat(ca_pre)
        ca_min = float(ca_pre.abs().min())
        print(f"hat_l: seed {seed} closest LeakyReLU / ReLU map input to zero {closest[0]:.2e}, "
              f"closest channel-attention ReLU input {ca_min:.2e} of {ca_pre.numel()}")
        if closest[0] > 2e-7 and ca_min > 1e-3:
            break
    else:
        raise RuntimeError("no well-conditioned draw found")
    r = torch.randn(y.shape, generator=sgen)
    (y * r).sum().backward()
    A = {"seed": np.int64(seed), "x": x.detach().numpy(), "r": r.numpy(), "y": y.detach().numpy(),
         "gx": x.grad.numpy().copy(),
         "ca/pre": ca_pre.numpy()}   # ReLU inputs of the 72 channel-attention bottlenecks, in forward order
    keys, s, a = checksums(dict(net.named_parameters()))
    A["p/keys"], A["p/sum"], A["p/abs"] = keys, s, a
    grads = {k: v.grad for k, v in net.named_parameters()}
    A["g/sum"] = np.array([float(grads[str(k)].double().sum()) for k in keys])
    A["g/abs"] = np.array([float(grads[str(k)].double().abs().sum()) for k in keys])
    A["g/l2"] = np.array([float(grads[str(k)].double().norm()) for k in keys])
    for k in ("conv_first.weight", "layers.0.residual_group.blocks.0.attn.qkv.weight",
              "layers.0.residual_group.blocks.1.attn.relative_position_bias_table",
              "layers.3.residual_group.blocks.2.conv_block.cab.0.weight",
              "layers.3.residual_group.blocks.2.conv_block.cab.3.attention.1.weight",
              "layers.5.residual_group.overlap_attn.relative_position_bias_table",
              "layers.5.residual_group.overlap_attn.qkv.weight", "layers.7.residual_group.blocks.5.mlp.fc1.bias",
              "layers.11.residual_group.blocks.5.attn.proj.weight", "layers.11.conv.weight",
              "layers.6.residual_group.blocks.3.norm2.weight", "upsample.2.bias", "conv_last.weight"):
        A[f"gfull/{k}"] = grads[k].numpy().copy()
    save("cfg4_hat_l.npz", **A)


def gen_step(name: str, opt) -> None:
    from neosr.models import build_model

    otf = opt["model_type"] == "otf"
    R = None
    if otf:
        import neosr.data.transforms as transforms_mod
        import neosr.models.otf as otf_mod
        import neosr.utils.diffjpeg as dj

        dj.device = torch.device("cpu")
        orig_filter2d = dj.filter2D
        otf_mod.filter2D = lambda img, k: orig_filter2d(img.contiguous(), k)
        R = Recorder()
        R.install(otf_mod, transforms_mod)
        opt["datasets"]["train"].update(opt["degradations"])  # train.py:69-70
    torch.manual_seed(1024)
    random.seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    seed_vgg_(model.cri_perceptual.vgg)
    gname = opt["network_g"]["type"]
    S = {}
    big = gname in SAMPLE_G
    if big:
        S["init_g/keys"], S["init_g/sum"], S["init_g/abs"] = checksums(model.net_g.state_dict())
    else:
        for k, v in np_state(model.net_g.state_dict()).items():
            S[f"init_g/{k}"] = v
    if getattr(model, "net_d", None) is not None:
        for k, v in np_state(model.net_d.state_dict()).items():
            S[f"init_d/{k}"] = v
    gen = torch.Generator().manual_seed(4242)
    rng = np.random.default_rng(11)
    if otf:
        import neosr.data.degradations as degr

        def kernels(b):
            ks = []
            for _ in range(b):
                size = int(rng.choice([7, 9, 11, 13, 15, 17, 19, 21]))
                k = degr.random_mixed_kernels(
                    ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"],
                    [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], size, [0.2, 3], [0.2, 3], [-np.pi, np.pi],
                    [0.5, 4], [1, 2], noise_range=None)
                p = (21 - size) // 2
                ks.append(np.pad(k, ((p, p), (p, p))))
            return torch.from_numpy(np.stack(ks)).float()

        def sinc(b):
            ks = []
            for _ in range(b):
                size = int(rng.choice([7, 9, 11, 13, 15, 17, 19, 21]))
                ks.append(degr.circular_lowpass_kernel(rng.uniform(np.pi / 3, np.pi), size, pad_to=21))
            return torch.from_numpy(np.stack(ks)).float()

    logs, keys = [], None
    niter = 3 if otf else 2
    for it in range(1, niter + 1):
        if otf:
            batch = {"gt": smooth_images(gen, 2, 128, 128), "kernel1": kernels(2), "kernel2": kernels(2),
                     "sinc_kernel": sinc(2)}
            for k, v in batch.items():
                S[f"it{it}/{k}"] = v.numpy()
            R.on = True
            model.feed_data(batch)
            R.on = False
            pack_draws(S, f"it{it}/draws", R.drain())
            S[f"it{it}/lq"] = model.lq.numpy().copy()
            S[f"it{it}/gt_out"] = model.gt.numpy().copy()
        else:
            lq = torch.rand(2, 3, 16, 16, generator=gen)
            gt = torch.rand(2, 3, 64, 64, generator=gen)
            S[f"it{it}/lq"], S[f"it{it}/gt"] = lq.numpy(), gt.numpy()
            model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        keys = list(log.keys())
        logs.append([float(log[k]) for k in keys])
        S[f"it{it}/out"] = model.output.detach().numpy().copy()
        print(name, it, dict(zip(keys, logs[-1])))
    S["log"] = np.asarray(logs, dtype=np.float64)
    S["log_keys"] = np.array(keys)
    gsd = model.net_g.state_dict()
    if big:
        _, S["final_g/sum"], S["final_g/abs"] = checksums(gsd)
        for k in SAMPLE_G[gname]:
            S[f"final_g/w/{k}"] = gsd[k].numpy().copy()
    else:
        for k, v in np_state(gsd).items():
            S[f"final_g/{k}"] = v
    if getattr(model, "net_d", None) is not None:
        for k, v in np_state(model.net_d.state_dict()).items():
            S[f"final_d/{k}"] = v
    save(f"step_{name}.npz", **S)


def main() -> None:
    name = sys.argv[1]
    tmp = Path(tempfile.mkdtemp()) / f"golden_{name}.toml"
    tmp.write_text(TOMLS[name])
    if name not in ("swinir_medium", "hat_l"):
        (HERE / f"golden_{name}.toml").write_text(TOMLS[name])
    install_reference(str(tmp))
    if name == "swinir_medium":
        gen_swinir_medium()
        return
    if name == "hat_l":
        gen_hat_l()
        return
    from neosr.utils.options import parse_options

    opt, _ = parse_options(str(REF), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    gen_step(name, opt)


if __name__ == "__main__":
    main()
