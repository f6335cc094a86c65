"""Small host utilities (seeding, logging) used on the training path."""

from __future__ import annotations

import logging
import random
from pathlib import Path

import torch

from neosr_amd.utils.dist_util import get_dist_info

_initialized: set[str] = set()


def set_random_seed(seed: int) -> None:
    """neosr/utils/misc.py:43-46: python `random` + torch global generator."""
    random.seed(seed)
    torch.manual_seed(seed)


def get_root_logger(name: str = "neosr", level: int = logging.INFO) -> logging.Logger:
    """Rank-0 stdout logger (neosr/utils/logger.py:158-207; other ranks only log errors)."""
    logger = logging.getLogger(name)
    if name in _initialized:
        return logger
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s"))
    logger.addHandler(handler)
    logger.propagate = False
    logger.setLevel(level if get_dist_info()[0] == 0 else logging.ERROR)
    _initialized.add(name)
    return logger


class tc:
    """terminal colours used in the reference's messages (neosr/utils/misc.py)."""

    red = "\033[31m"
    light_blue = "\033[94m"
    light_green = "\033[92m"
    end = "\033[0m"


def check_resume(opt: dict, resume_iter: int) -> None:
    """neosr/utils/misc.py:131-165: when resuming, every `network_*` is reloaded from
    `<path.models>/net_<x>_<iter>.pth` (overriding pretrain paths, unless listed in
    `path.ignore_resume_networks`) and `param_key_* = "params_ema"` falls back to "params"."""
    if opt["path"].get("resume_state", None):
        networks = [key for key in opt if key.startswith("network_")]
        if any(opt["path"].get(f"pretrain_{n}") is not None for n in networks):
            print("NOTICE: pretrain_network_* is ignored during resuming.")
        for network in networks:
            basename = network.replace("network_", "")
            ignore = opt["path"].get("ignore_resume_networks")
            if ignore is None or network not in ignore:
                opt["path"][f"pretrain_{network}"] = Path(opt["path"]["models"]) / f"net_{basename}_{resume_iter}.pth"
        for key in [k for k in opt["path"] if k.startswith("param_key")]:
            if opt["path"][key] == "params_ema":
                opt["path"][key] = "params"


def load_resume_state(opt: dict, device="cpu"):
    """train.py:124-147: newest `<experiments>/<name>/training_states/*.state` under auto_resume, else
    `path.resume_state`; returns the loaded state (or None) after `check_resume`."""
    resume_state_path = None
    if opt.get("auto_resume"):
        state_path = Path("experiments") / opt["name"] / "training_states"
        if state_path.is_dir():
            states = [float(p.name.split(".state")[0]) for p in state_path.iterdir() if p.name.endswith("state")]
            if states:
                resume_state_path = state_path / f"{max(states):.0f}.state"
                opt["path"]["resume_state"] = resume_state_path
    elif opt["path"].get("resume_state"):
        resume_state_path = opt["path"]["resume_state"]
    if resume_state_path is None:
        return None
    resume_state = torch.load(resume_state_path, map_location=device, weights_only=True)
    check_resume(opt, resume_state["iter"])
    return resume_state
