#!/usr/bin/env python
"""Validation-time inference fixtures, produced by RUNNING THE REFERENCE on CPU (build container only):
    python tests/golden/gen_golden_val.py  ->  val.npz

`image.test()` (neosr/models/image.py:664-783) of the reference `image` model (esrgan reduced, EMA on):
whole-image branch (`val.tile = -1`) and the partitioned branch (`val.tile = 24` on a 40x52 input:
flip-padding to a multiple of the split count, 16-pixel overlaps, merge, crop), from the EMA weights.
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
name = "golden_val"
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
type = "adamw"
lr = 1e-3
betas = [ 0.9, 0.99 ]
weight_decay = 0.01

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[val]
val_freq = 1000
tile = 24

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_val.toml"
    tmp.write_text(TOML)
    (HERE / "golden_val.toml").write_text(TOML)
    install_reference(str(tmp))
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    random.seed(1024)
    np.random.seed(1024)
    torch.manual_seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    A = {f"init/{k}": v for k, v in np_state(model.net_g.state_dict()).items()}
    gen = torch.Generator().manual_seed(7)
    # one training step so that EMA != net_g is not guaranteed (first EMA update copies); perturb the EMA instead
    with torch.no_grad():
        for p in model.net_g_ema.parameters():
            p.add_(torch.randn(p.shape, generator=gen) * 0.01)
    for k, v in np_state(model.net_g_ema.state_dict()).items():
        A[f"ema/{k}"] = v
    for name, (h, w), tile in (("whole", (20, 28), -1), ("tiled", (40, 52), 24), ("tiled_small", (17, 23), 24)):
        lq = torch.rand(1, 3, h, w, generator=gen)
        A[f"{name}/lq"] = lq.numpy()
        model.opt["val"]["tile"] = tile
        model.feed_data({"lq": lq})
        model.test()
        A[f"{name}/out"] = model.output.detach().numpy().copy()
        assert model.net_g.training
    save("val.npz", **A)


if __name__ == "__main__":
    main()
