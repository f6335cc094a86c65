#!/usr/bin/env python
"""Golden fixtures for Friendly-SAM, produced by RUNNING THE REFERENCE on CPU (build container only):
    python tests/golden/gen_golden_fsam.py  ->  fsam.npz, step_fsam.npz

  fsam.npz       reference `fsam` (base torch.optim.AdamW) on two tensors, 4 steps: the gradient at w is
                 given, the closure installs a given gradient at w + e(w); parameters at the perturbed
                 point and after every step, momentum / old_p state
  step_fsam.npz  5 iterations of the reference `image` model: esrgan (reduced) + L1 + adamw + clip + EMA
                 with `sam = "fsam"`, `sam_init = 3` (iterations 1-2 plain, 3-5 sharpness-aware: closure
                 twice, no clipping, the SAM optimizer's own AdamW state)
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
from gen_golden import install_reference, np_state, save  # noqa: E402

TOML = """
name = "golden_fsam"
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
num_block = 2
num_grow_ch = 8

[train]
ema = 0.999
grad_clip = true
sam = "fsam"
sam_init = 3

[train.optim_g]
type = "adamw"
lr = 2e-4
betas = [ 0.9, 0.99 ]
weight_decay = 0.01

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_fsam.toml"
    tmp.write_text(TOML)
    (HERE / "golden_fsam.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.models import build_model
    from neosr.optimizers.fsam import fsam
    from neosr.utils.options import parse_options

    gen = torch.Generator().manual_seed(5)
    A = {}
    ps = [torch.randn(6, 5, generator=gen).requires_grad_(True), torch.randn(13, generator=gen).requires_grad_(True)]
    for i, p in enumerate(ps):
        A[f"p0/{i}"] = p.detach().numpy().copy()
    opt = fsam(ps, torch.optim.AdamW, rho=0.5, sigma=1, lmbda=0.9, adaptive=True, lr=1e-2, betas=(0.9, 0.99),
               weight_decay=0.01)
    for step in range(1, 5):
        g2 = []
        for i, p in enumerate(ps):
            g = torch.randn(p.shape, generator=gen)
            A[f"g{step}/{i}"] = g.numpy().copy()
            p.grad = g.clone()
            g2.append(torch.randn(p.shape, generator=gen))
            A[f"h{step}/{i}"] = g2[-1].numpy().copy()

        def closure(_it, g2=g2, step=step):
            for i, p in enumerate(ps):
                A[f"pert{step}/{i}"] = p.detach().numpy().copy()
                p.grad = g2[i].clone()

        opt.step(closure, step)
        for i, p in enumerate(ps):
            A[f"p{step}/{i}"] = p.detach().numpy().copy()
            A[f"mom{step}/{i}"] = opt.state[p]["momentum"].numpy().copy()
    save("fsam.npz", **A)

    # ---- model trajectory
    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    random.seed(1024)
    np.random.seed(1024)
    torch.manual_seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    A = {f"init/{k}": v for k, v in np_state(model.net_g.state_dict()).items()}
    dgen = torch.Generator().manual_seed(77)
    logs = []
    for it in range(1, 6):
        lq = torch.rand(2, 3, 16, 16, generator=dgen)
        gt = torch.rand(2, 3, 64, 64, generator=dgen)
        A[f"lq{it}"], A[f"gt{it}"] = lq.numpy(), gt.numpy()
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        logs.append([log["l_g_pix"], log["l_g_total"]])
        A[f"out{it}"] = model.output.detach().numpy().copy()
        for k, v in np_state(model.net_g.state_dict()).items():
            if k.endswith("conv_first.weight") or k.endswith("conv_last.bias"):
                A[f"w{it}/{k}"] = v
    A["log"] = np.asarray(logs, dtype=np.float64)
    for k, v in np_state(model.net_g.state_dict()).items():
        A[f"final/{k}"] = v
    for k, v in np_state(model.net_g_ema.state_dict()).items():
        A[f"ema/{k}"] = v
    names = [n for n, _ in model.net_g.named_parameters()]
    params = list(model.net_g.parameters())
    for i in (0, len(names) - 1):
        A[f"momentum/{names[i]}"] = model.sam_optimizer_g.state[params[i]]["momentum"].numpy().copy()
        bst = model.sam_optimizer_g.base_optimizer.state[params[i]]
        A[f"base_exp_avg/{names[i]}"] = bst["exp_avg"].numpy().copy()
        A[f"base_step/{names[i]}"] = np.asarray(float(bst["step"]))
    save("step_fsam.npz", **A)


if __name__ == "__main__":
    main()
