#!/usr/bin/env python
"""Golden fixtures for the Schedule-Free Adan optimizer, produced by RUNNING THE REFERENCE on CPU
(build container only):  python tests/golden/gen_golden_adan.py  ->  adan_sf.npz, step_adan.npz

  adan_sf.npz    reference `adan_sf` on two tensors, 5 steps with given gradients (warmup_steps 3,
                 schedule_free on and off): parameters after every step, after `.eval()` and after
                 `.train()`, final state tensors and group scalars
  step_adan.npz  4 iterations of the reference `image` model: esrgan (reduced) + L1 + adan_sf
                 (schedule_free, warmup 3) + clip + EMA, incl. a save-style eval()/train() round trip
                 after iteration 2: log, outputs, final weights / EMA / optimizer state
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
name = "golden_adan"
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

[train.optim_g]
type = "adan_sf"
lr = 8e-4
betas = [ 0.98, 0.92, 0.987 ]
weight_decay = 0.02
schedule_free = true
warmup_steps = 3

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_adan.toml"
    tmp.write_text(TOML)
    (HERE / "golden_adan.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.models import build_model
    from neosr.optimizers.adan_sf import adan_sf
    from neosr.utils.options import parse_options

    gen = torch.Generator().manual_seed(3)
    A = {}
    for sf in (True, False):
        tag = "sf" if sf else "plain"
        ps = [torch.randn(5, 7, generator=gen).requires_grad_(True), torch.randn(11, generator=gen).requires_grad_(True)]
        for i, p in enumerate(ps):
            A[f"{tag}/p0/{i}"] = p.detach().numpy().copy()
        opt = adan_sf(ps, lr=2e-3, betas=(0.98, 0.92, 0.987), weight_decay=0.02, warmup_steps=3, schedule_free=sf)
        if sf:
            opt.train()
        for step in range(1, 6):
            for i, p in enumerate(ps):
                g = torch.randn(p.shape, generator=gen) * (0.5 + 0.1 * step)
                A[f"{tag}/g{step}/{i}"] = g.numpy().copy()
                p.grad = g.clone()
            opt.step()
            for i, p in enumerate(ps):
                A[f"{tag}/p{step}/{i}"] = p.detach().numpy().copy()
            if sf and step == 3:
                opt.eval()
                for i, p in enumerate(ps):
                    A[f"{tag}/p_eval/{i}"] = p.detach().numpy().copy()
                opt.train()
                for i, p in enumerate(ps):
                    A[f"{tag}/p_train/{i}"] = p.detach().numpy().copy()
        for i, p in enumerate(ps):
            for k, v in opt.state[p].items():
                A[f"{tag}/state/{k}/{i}"] = v.numpy().copy()
        g0 = opt.param_groups[0]
        A[f"{tag}/group"] = np.array([g0["step"], g0["weight_sum"], g0["lr_max"]], dtype=np.float64)
    save("adan_sf.npz", **A)

    # ---- model trajectory
    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    random.seed(1024)
    np.random.seed(1024)
    torch.manual_seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    A = {f"init/{k}": v for k, v in np_state(model.net_g.state_dict()).items()}
    dgen = torch.Generator().manual_seed(99)
    logs = []
    for it in range(1, 5):
        lq = torch.rand(2, 3, 16, 16, generator=dgen)
        gt = torch.rand(2, 3, 64, 64, generator=dgen)
        A[f"lq{it}"], A[f"gt{it}"] = lq.numpy(), gt.numpy()
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it)
        log = model.get_current_log()
        logs.append([log["l_g_pix"], log["l_g_total"]])
        A[f"out{it}"] = model.output.detach().numpy().copy()
        if it == 2:  # what save_network does around torch.save (base.py:325-354)
            model.optimizer_g.eval()
            for k, v in np_state(model.net_g.state_dict()).items():
                A[f"eval2/{k}"] = v
            model.optimizer_g.train()
    A["log"] = np.asarray(logs, dtype=np.float64)
    for k, v in np_state(model.net_g.state_dict()).items():
        A[f"final/{k}"] = v
    for k, v in np_state(model.net_g_ema.state_dict()).items():
        A[f"ema/{k}"] = v
    st = model.optimizer_g.state_dict()["state"]
    names = [n for n, _ in model.net_g.named_parameters()]
    for i in (0, len(names) - 1):
        for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "z", "neg_pre_grad"):
            A[f"optstate/{k}/{names[i]}"] = st[i][k].numpy().copy()
    g0 = model.optimizer_g.param_groups[0]
    A["group"] = np.array([g0["step"], g0["weight_sum"], g0["lr_max"]], dtype=np.float64)
    save("step_adan.npz", **A)


if __name__ == "__main__":
    main()
