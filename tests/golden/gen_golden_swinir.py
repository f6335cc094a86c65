#!/usr/bin/env python
"""Golden fixtures for the SwinIR window-attention path, produced by RUNNING THE REFERENCE on CPU
(build container only):  python tests/golden/gen_golden_swinir.py

  swinir_prims.npz   relative_position_index, calculate_mask; a SwinTransformerBlock (dim 24,
                     2 heads, shift 0 and 4, 16x24 tokens) forward + all gradients
  swinir_nets.npz    three tiny `swinir` nets (embed 24 / 32, depths (2,2)) covering the three upsamplers /
                     both resi_connections: forward + all gradients; x_size != img_size so the
                     reference takes its calculate_mask() branch
  swinir_init.npz    per-tensor sum / abs-sum of `swinir_small`/`swinir_medium` seeded inits
                     (init-draw parity) and the state-dict key order
  step_swinir.npz    2 iterations of the reference `image` model with network_g = swinir_small
                     (drop_path_rate 0), L1, AdamW, clip, EMA: log_dict, outputs, weight sums

drop_path_rate is 0 in every fixture: DropPath draws from the global torch RNG.
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

TOML = """
name = "golden_swinir"
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
type = "swinir_small"
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


def tensor_sums(sd):
    keys = list(sd.keys())
    s = np.array([float(v.double().sum()) for v in sd.values()])
    a = np.array([float(v.double().abs().sum()) for v in sd.values()])
    return keys, s, a


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_swinir.toml"
    tmp.write_text(TOML)
    (HERE / "golden_swinir.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.archs import swinir_arch as S
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    gen = torch.Generator().manual_seed(5)
    A = {}
    # ---- index / mask
    wa = S.WindowAttention(24, (8, 8), 2, flash_attn=False)
    A["rel_index_8"] = wa.relative_position_index.numpy().copy()
    for shift in (0, 4):
        torch.manual_seed(31 + shift)
        blk = S.SwinTransformerBlock(24, (16, 24), 2, flash_attn=False, window_size=8, shift_size=shift,
                                     mlp_ratio=2.0, drop_path=0.0)
        with torch.no_grad():  # make every parameter non-trivial (biases / LN affine start at 0 / 1)
            for p in blk.parameters():
                p.add_(torch.randn(p.shape, generator=gen) * 0.05)
        if shift:
            A["mask_16x24_s4"] = blk.attn_mask.numpy().copy()
            A["mask_32x16_s4"] = blk.calculate_mask((32, 16)).numpy().copy()
        x = torch.randn(2, 16 * 24, 24, generator=gen).requires_grad_(True)
        r = torch.randn(2, 16 * 24, 24, generator=gen)
        y = blk(x, (16, 24))
        (y * r).sum().backward()
        pre = f"blk_s{shift}"
        A[f"{pre}/x"], A[f"{pre}/r"], A[f"{pre}/y"] = x.detach().numpy(), r.numpy(), y.detach().numpy()
        A[f"{pre}/gx"] = x.grad.numpy().copy()
        for k, v in blk.named_parameters():
            A[f"{pre}/p/{k}"] = v.detach().numpy().copy()
            A[f"{pre}/g/{k}"] = v.grad.numpy().copy()
    save("swinir_prims.npz", **A)

    # ---- tiny full nets
    A = {}
    cfgs = {
        "ps": dict(upsampler="pixelshuffle", resi_connection="1conv"),
        "psd": dict(upsampler="pixelshuffledirect", resi_connection="1conv"),
        "nc": dict(upsampler="nearest+conv", resi_connection="3conv"),
    }
    for tag, kw in cfgs.items():
        # LeakyReLU'(v) jumps at v = 0: an activation input within fp32 rounding of zero makes the
        # gradient depend on the summation order of the conv that produced it.  Re-draw until every
        # LeakyReLU input of the forward pass is at least 2e-7 away from zero (activations are O(0.05),
        # so fp32 rounding of a conv output is O(1e-8)), so the fixture pins the
        # arithmetic and not one implementation's rounding of an ill-conditioned element.
        for seed in range(41, 141):
            torch.manual_seed(seed)
            # 3conv has a (dim/4 -> dim/4) 1x1 conv: embed 32 keeps its GEMM K a multiple of 4
            net = S.swinir(img_size=16, embed_dim=32 if tag == "nc" else 24, depths=(2, 2), num_heads=(2, 2),
                           window_size=8, mlp_ratio=2.0, drop_path_rate=0.0, **kw)
            sgen = torch.Generator().manual_seed(1000 + seed)
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(torch.randn(p.shape, generator=sgen) * 0.02)
            # nearest+conv runs three LeakyReLUs at 16x the LR area: a smaller input keeps the number of
            # activation inputs (hence the chance of one landing on zero) manageable
            hw = (8, 16) if tag == "nc" else (16, 24)
            x = torch.rand(2, 3, *hw, generator=sgen).requires_grad_(True)
            closest = [float("inf")]
            hooks = [m.register_forward_pre_hook(lambda _m, a: closest.__setitem__(0, min(closest[0], float(a[0].abs().min()))))
                     for m in net.modules() if isinstance(m, torch.nn.LeakyReLU)]
            y = net(x)
            for h in hooks:
                h.remove()
            print(f"{tag}: seed {seed} closest LeakyReLU input to zero {closest[0]:.2e}")
            if closest[0] > 2e-7:
                break
        else:
            raise RuntimeError("no well-conditioned draw found")
        r = torch.randn(y.shape, generator=gen)
        (y * r).sum().backward()
        A[f"{tag}/x"], A[f"{tag}/r"], A[f"{tag}/y"] = x.detach().numpy(), r.numpy(), y.detach().numpy()
        A[f"{tag}/gx"] = x.grad.numpy().copy()
        for k, v in net.named_parameters():
            A[f"{tag}/p/{k}"] = v.detach().numpy().copy()
            A[f"{tag}/g/{k}"] = v.grad.numpy().copy()
        A[f"{tag}/keys"] = np.array(list(net.state_dict().keys()))
    save("swinir_nets.npz", **A)

    # ---- seeded-init parity
    A = {}
    for name in ("swinir_small", "swinir_medium"):
        torch.manual_seed(1024)
        net = getattr(S, name)()
        keys, s, a = tensor_sums(net.state_dict())
        A[f"{name}/keys"], A[f"{name}/sum"], A[f"{name}/abs"] = np.array(keys), s, a
    save("swinir_init.npz", **A)

    # ---- 2-iteration training trajectory
    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    seed = opt["manual_seed"]
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
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
              "layers.3.residual_group.blocks.5.mlp.fc2.weight", "upsample.0.weight", "norm.weight"):
        A[f"final/w/{k}"] = sd[k].numpy().copy()
    save("step_swinir.npz", **A)


if __name__ == "__main__":
    main()
