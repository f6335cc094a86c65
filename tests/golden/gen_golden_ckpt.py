#!/usr/bin/env python
"""Checkpoint / resume fixtures, produced by RUNNING THE REFERENCE on CPU (build container only):
    python tests/golden/gen_golden_ckpt.py  ->  ckpt/net_g_2.pth, ckpt/2.state, ckpt.npz

The reference `image` model (esrgan reduced, L1, adan_sf schedule-free, EMA, MultiStepLR) trains 2
iterations and `save()`s (image.py:932-942, base.py:281-470): `net_g_2.pth` (the EMA weights under
"params") and `2.state` are kept verbatim as data fixtures.  A second reference model is then built
the way train.py resumes (misc.check_resume -> pretrain_network_g = net_g_2.pth; resume_training) and
trains iterations 3-4; ckpt.npz holds the inputs, logs, outputs and final weights / EMA of that run.
"""

from __future__ import annotations

import random
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import install_reference, np_state, save  # noqa: E402

TOML = """
name = "golden_ckpt"
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
ema = 0.9
grad_clip = true

[train.optim_g]
type = "adan_sf"
lr = 8e-4
betas = [ 0.98, 0.92, 0.987 ]
weight_decay = 0.02
schedule_free = true
warmup_steps = 3

[train.scheduler]
type = "multisteplr"
milestones = [ 1, 3 ]
gamma = 0.5

[train.pixel_opt]
type = "L1Loss"
loss_weight = 1.0

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


def build(parse_options, build_model, extra_path=None):
    opt, _ = parse_options(str(HERE), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 0
    if extra_path:
        opt["path"].update(extra_path)
    random.seed(1024)
    np.random.seed(1024)
    torch.manual_seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    return opt, model


def main():
    tmpd = Path(tempfile.mkdtemp())
    tmp = tmpd / "golden_ckpt.toml"
    tmp.write_text(TOML)
    (HERE / "golden_ckpt.toml").write_text(TOML)
    install_reference(str(tmp))
    _load = torch.load

    def load_cpu(f, *a, **kw):  # base.load_network maps to "cuda" (base.py:381-383)
        kw["map_location"] = "cpu"
        return _load(f, *a, **kw)

    torch.load = load_cpu
    from neosr.models import build_model
    from neosr.utils.misc import check_resume
    from neosr.utils.options import parse_options

    dirs = {"models": str(tmpd / "models"), "training_states": str(tmpd / "training_states")}
    for d in dirs.values():
        Path(d).mkdir(parents=True)
    opt, model = build(parse_options, build_model, dirs)
    A = {f"init/{k}": v for k, v in np_state(model.net_g.state_dict()).items()}
    dgen = torch.Generator().manual_seed(4242)
    data = []
    for it in range(1, 5):
        data.append((torch.rand(2, 3, 16, 16, generator=dgen), torch.rand(2, 3, 64, 64, generator=dgen)))
        A[f"lq{it}"], A[f"gt{it}"] = data[-1][0].numpy(), data[-1][1].numpy()
    for it in (1, 2):
        model.feed_data({"lq": data[it - 1][0], "gt": data[it - 1][1]})
        model.optimize_parameters(it)
        model.update_learning_rate(it, warmup_iter=-1)
    model.save(0, 2)
    for k, v in np_state(model.net_g.state_dict()).items():
        A[f"after_save/{k}"] = v  # schedule-free eval()/train() round trips around both writes
    out = HERE / "ckpt"
    out.mkdir(exist_ok=True)
    shutil.copy(Path(dirs["models"]) / "net_g_2.pth", out / "net_g_2.pth")
    shutil.copy(Path(dirs["training_states"]) / "2.state", out / "2.state")

    # ---- resume the way train.py does (train.py:124-147, misc.py:131-165)
    state = torch.load(out / "2.state", map_location="cpu", weights_only=True)
    opt2, _ = parse_options(str(HERE), is_train=True)
    opt2["path"].update(dirs)
    opt2["path"]["resume_state"] = str(out / "2.state")
    check_resume(opt2, state["iter"])
    A["resume_pretrain_name"] = np.array(Path(opt2["path"]["pretrain_network_g"]).name)
    _, model2 = build(parse_options, build_model, {**dirs, "pretrain_network_g": opt2["path"]["pretrain_network_g"],
                                                     "resume_state": opt2["path"]["resume_state"]})
    model2.resume_training(state)
    for k, v in np_state(model2.net_g.state_dict()).items():
        A[f"resumed/{k}"] = v
    A["resumed_lr"] = np.asarray(model2.get_current_learning_rate(), dtype=np.float64)
    logs = []
    for it in (3, 4):
        model2.feed_data({"lq": data[it - 1][0], "gt": data[it - 1][1]})
        model2.optimize_parameters(it)
        model2.update_learning_rate(it, warmup_iter=-1)
        log = model2.get_current_log()
        logs.append([log["l_g_pix"], log["l_g_total"]])
        A[f"out{it}"] = model2.output.detach().numpy().copy()
        A[f"lr{it}"] = np.asarray(model2.get_current_learning_rate(), dtype=np.float64)
    A["log"] = np.asarray(logs, dtype=np.float64)
    for k, v in np_state(model2.net_g.state_dict()).items():
        A[f"final/{k}"] = v
    for k, v in np_state(model2.net_g_ema.state_dict()).items():
        A[f"ema/{k}"] = v
    g0 = model2.optimizer_g.param_groups[0]
    A["group"] = np.array([g0["step"], g0["weight_sum"], g0["lr_max"], g0["lr"]], dtype=np.float64)
    save("ckpt.npz", **A)


if __name__ == "__main__":
    main()
