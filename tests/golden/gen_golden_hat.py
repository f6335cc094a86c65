#!/usr/bin/env python
"""Golden fixtures for the HAT path, produced by RUNNING THE REFERENCE on CPU (build container only):
    python tests/golden/gen_golden_hat.py

  hat_prims.npz   relative_position_index_SA / _OCA (window 16, overlap 0.5) and calculate_mask as the
                  reference builds them; CAB, HAB (shift 0 / 8) and OCAB blocks (dim 24, 2 heads, 32x48
                  tokens) forward + all gradients
  hat_net.npz     a tiny `hat` (embed 24, depths (2,), window 16) forward + all gradients on 2x3x32x32
  hat_init.npz    per-tensor sums of the seeded `hat_s` init + state-dict key order
  hat_l_fwd.npz   `hat_l` (seeded init, drop_path irrelevant in eval) forward at B=1, 64x64 LR: input + output
  step_hat.npz    2 iterations of the reference `image` model with network_g = hat_s (drop_path_rate 0),
                  L1, AdamW, clip, EMA on 16x16 LR patches: log, outputs, weight sums, a few final tensors
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
from gen_golden import install_reference, save  # noqa: E402
from gen_golden_swinir import tensor_sums  # noqa: E402

TOML = """
name = "golden_hat"
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
type = "hat_s"
drop_path_rate = 0.0

[train]
ema = 0.999
grad_clip = true

[train.optim_g]
type = "adamw"
lr = 2e-4
betas = [ 0.9, 0.99 ]

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def fwd_bwd(A, pre, mod, call, x, gen, noise=0.05):
    with torch.no_grad():  # biases / LayerNorm affine start at 0 / 1: make everything non-trivial
        for p in mod.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * noise)
    x = x.requires_grad_(True)
    y = call(x)
    r = torch.randn(y.shape, generator=gen)
    (y * r).sum().backward()
    A[f"{pre}/x"], A[f"{pre}/r"], A[f"{pre}/y"] = x.detach().numpy(), r.numpy(), y.detach().numpy()
    A[f"{pre}/gx"] = x.grad.numpy().copy()
    for k, v in mod.named_parameters():
        A[f"{pre}/p/{k}"] = v.detach().numpy().copy()
        A[f"{pre}/g/{k}"] = v.grad.numpy().copy()


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_hat.toml"
    tmp.write_text(TOML)
    (HERE / "golden_hat.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.archs import hat_arch as HA
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    gen = torch.Generator().manual_seed(7)
    A = {}
    torch.manual_seed(51)
    tiny = HA.hat(img_size=32, embed_dim=24, depths=(2,), num_heads=(2,), window_size=16, compress_ratio=3,
                  squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle")
    A["rpi_sa_16"] = tiny.relative_position_index_SA.numpy().copy()
    A["rpi_oca_16"] = tiny.relative_position_index_OCA.numpy().copy()
    A["mask_32x48_s8"] = tiny.calculate_mask((32, 48)).numpy().copy()
    rpi_sa, rpi_oca = tiny.relative_position_index_SA, tiny.relative_position_index_OCA

    torch.manual_seed(52)
    cab = HA.CAB(24, compress_ratio=3, squeeze_factor=6)
    fwd_bwd(A, "cab", cab, lambda t: cab(t), torch.randn(2, 24, 20, 28, generator=gen), gen)
    for shift in (0, 8):
        torch.manual_seed(53 + shift)
        blk = HA.HAB(24, (32, 48), 2, window_size=16, shift_size=shift, compress_ratio=3, squeeze_factor=6,
                     conv_scale=0.01, mlp_ratio=2, drop_path=0.0)
        mask = tiny.calculate_mask((32, 48))
        fwd_bwd(A, f"hab_s{shift}", blk, lambda t, blk=blk, mask=mask: blk(t, (32, 48), rpi_sa, mask),
                torch.randn(2, 32 * 48, 24, generator=gen), gen)
    torch.manual_seed(55)
    oc = HA.OCAB(24, (32, 48), 16, 0.5, 2, mlp_ratio=2)
    fwd_bwd(A, "ocab", oc, lambda t: oc(t, (32, 48), rpi_oca), torch.randn(2, 32 * 48, 24, generator=gen), gen)
    save("hat_prims.npz", **A)

    # ---- tiny net (re-drawn until no LeakyReLU input sits within 2e-7 of zero, see gen_golden_swinir.py)
    A = {}
    for seed in range(61, 161):
        torch.manual_seed(seed)
        net = HA.hat(img_size=32, embed_dim=24, depths=(2,), num_heads=(2,), window_size=16, compress_ratio=3,
                     squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle")
        sgen = torch.Generator().manual_seed(2000 + seed)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
        x = torch.rand(2, 3, 32, 32, generator=sgen).requires_grad_(True)
        closest = [float("inf")]
        hooks = [m.register_forward_pre_hook(lambda _m, a: closest.__setitem__(0, min(closest[0], float(a[0].abs().min()))))
                 for m in net.modules() if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]
        y = net(x)
        for hk in hooks:
            hk.remove()
        print(f"tiny hat: seed {seed} closest (Leaky)ReLU input to zero {closest[0]:.2e}")
        if closest[0] > 2e-7:
            break
    else:
        raise RuntimeError("no well-conditioned draw found")
    r = torch.randn(y.shape, generator=gen)
    (y * r).sum().backward()
    A["x"], A["r"], A["y"], A["gx"] = x.detach().numpy(), r.numpy(), y.detach().numpy(), x.grad.numpy().copy()
    for k, v in net.named_parameters():
        A[f"p/{k}"] = v.detach().numpy().copy()
        A[f"g/{k}"] = v.grad.numpy().copy()
    A["keys"] = np.array(list(net.state_dict().keys()))
    save("hat_net.npz", **A)

    # ---- seeded init
    A = {}
    torch.manual_seed(1024)
    net = HA.hat_s()
    keys, s, a = tensor_sums(net.state_dict())
    A["hat_s/keys"], A["hat_s/sum"], A["hat_s/abs"] = np.array(keys), s, a
    save("hat_init.npz", **A)

    # ---- hat_l forward, B = 1
    torch.manual_seed(1024)
    net = HA.hat_l().eval()
    x = torch.rand(1, 3, 64, 64, generator=gen)
    with torch.no_grad():
        y = net(x)
    keys, s, a = tensor_sums(net.state_dict())
    save("hat_l_fwd.npz", x=x.numpy(), y=y.numpy(), init_sum=s, keys=np.array(keys))

    # ---- 2-iteration trajectory with hat_s
    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    random.seed(1024)
    np.random.seed(1024)
    torch.manual_seed(1024)
    model = build_model(opt)
    A = {}
    keys, s, a = tensor_sums(model.net_g.state_dict())
    A["init/keys"], A["init/sum"], A["init/abs"] = np.array(keys), s, a
    dgen = torch.Generator().manual_seed(77)
    for it in range(1, 3):
        lq = torch.rand(2, 3, 16, 16, generator=dgen)
        gt = torch.rand(2, 3, 64, 64, generator=dgen)
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        A[f"it{it}/lq"], A[f"it{it}/gt"] = lq.numpy(), gt.numpy()
        A[f"it{it}/output"] = model.output.detach().numpy().copy()
        for k, v in model.log_dict.items():
            A[f"it{it}/log/{k}"] = np.float64(v)
    keys, s, a = tensor_sums(model.net_g.state_dict())
    A["final/sum"], A["final/abs"] = s, a
    sd = model.net_g.state_dict()
    for k in ("conv_first.weight", "layers.0.residual_group.blocks.1.attn.relative_position_bias_table",
              "layers.2.residual_group.overlap_attn.relative_position_bias_table",
              "layers.5.residual_group.blocks.5.mlp.fc2.weight",
              "layers.3.residual_group.blocks.0.conv_block.cab.3.attention.1.weight", "norm.weight"):
        A[f"final/w/{k}"] = sd[k].numpy().copy()
    save("step_hat.npz", **A)


if __name__ == "__main__":
    main()
