"""CLI + TOML -> the nested ``opt`` dict every plugin reads.

Drop-in for neosr/utils/options.py:39-275: same flags (``-opt``, ``--launcher``, ``--auto_resume``,
``--debug``, ``--local_rank``), same TOML schema, same derived keys (``dist, rank, world_size,
deterministic, auto_resume, is_train, num_gpu, datasets.*.phase/scale, path.*``) and the same
seeding rule (``manual_seed + rank`` into python ``random`` and torch).  Differences, on purpose:
the parse is cached (the reference re-parses argv and re-enters ``init_dist`` from every
import-time ``net_opt()`` / ``rng()`` call) and ``argv`` may be passed explicitly.
"""

from __future__ import annotations

import argparse
import os
import random
import sys
from pathlib import Path
from typing import Any

import torch

try:  # python >= 3.11
    import tomllib
except ModuleNotFoundError:  # python 3.10: same parser under its pre-stdlib name
    import tomli as tomllib

from neosr_amd.utils.dist_util import get_dist_info, init_dist
from neosr_amd.utils.misc import set_random_seed, tc

_CACHE: dict[str, Any] = {"opt": None, "args": None}


def toml_load(f) -> dict[str, Any]:
    """Load a TOML file (neosr/utils/options.py:15-36)."""
    try:
        with Path(f).open("rb") as fh:
            return tomllib.load(fh)
    except Exception as exc:
        msg = f"{tc.red}Error decoding TOML file {f}: {exc}{tc.end}"
        raise tomllib.TOMLDecodeError(msg) from exc


def _build_parser(root_path) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="neosr", usage=argparse.SUPPRESS,
                                description="-------- neosr command-line options --------")
    p.add_argument("-opt", type=str, required=False, help="Path to option TOML file.")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none",
                   help="job launcher")
    p.add_argument("--auto_resume", action="store_true", default=False)
    p.add_argument("--debug", action="store_true")
    p.add_argument("--local_rank", type=int, default=0)
    # conversion flags are accepted (and ignored) so reference command lines keep parsing
    g = p.add_argument_group("model conversion")
    g.add_argument("--input", type=str, required=False)
    g.add_argument("-onnx", "--onnx", action="store_true", default=False)
    g.add_argument("-safetensor", "--safetensor", action="store_true", default=False)
    g.add_argument("-net", "--network", type=str, required=False)
    g.add_argument("-s", "--scale", type=int, default=4)
    g.add_argument("-window", "--window", type=int, default=None)
    g.add_argument("-opset", "--opset", type=int, default=17)
    g.add_argument("-static", "--static", type=int, nargs=3, default=None)
    g.add_argument("-nocheck", "--nocheck", action="store_true", default=False)
    g.add_argument("-fp16", "--fp16", action="store_true", default=False)
    g.add_argument("-optimize", "--optimize", action="store_true", default=False)
    g.add_argument("-fulloptimization", "--fulloptimization", action="store_true", default=False)
    g.add_argument("--output", type=str, required=False, default=root_path)
    return p


def parse_options(root_path, is_train: bool = True, argv: list[str] | None = None,
                  use_cache: bool = False) -> tuple[dict[str, Any], argparse.Namespace]:
    if use_cache and _CACHE["opt"] is not None:
        return _CACHE["opt"], _CACHE["args"]
    parser = _build_parser(root_path)
    args = parser.parse_args(argv) if argv is not None else parser.parse_known_args()[0]

    if args.input is None and args.opt is None:
        msg = f"{tc.red}Didn't get a config! Please link the config file using -opt /path/to/config.toml{tc.end}"
        raise ValueError(msg)
    if args.input is not None:
        return {}, args
    if not args.opt.endswith(".toml"):
        msg = f"{tc.light_blue}neosr has switched to TOML configuration files! See options/.{tc.end}"
        raise ValueError(msg)

    opt = toml_load(args.opt)

    # distributed settings
    if args.launcher == "none":
        opt["dist"] = False
    else:
        opt["dist"] = True
        if args.launcher == "slurm" and "dist_params" in opt:
            init_dist(args.launcher, **opt["dist_params"])
        else:
            init_dist(args.launcher)
    opt["rank"], opt["world_size"] = get_dist_info()

    # random seed / determinism
    seed = opt.get("manual_seed")
    if seed is None:
        opt["deterministic"] = False
        seed = random.randint(1024, 10000)
        opt["manual_seed"] = seed
    else:
        opt["deterministic"] = True
        os.environ["PYTHONHASHSEED"] = str(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        # our kernels use fixed-order reductions: deterministic with or without this flag
    set_random_seed(seed + opt["rank"])

    opt["auto_resume"] = args.auto_resume
    opt["is_train"] = is_train

    if args.debug and not opt["name"].startswith("debug"):
        opt["name"] = "debug_" + opt["name"]

    if opt.get("num_gpu", "auto") == "auto":
        opt["num_gpu"] = torch.cuda.device_count()

    for phase, dataset in opt.get("datasets", {}).items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        for key in ("dataroot_gt", "dataroot_lq"):
            if dataset.get(key) is not None:
                dataset[key] = str(Path(dataset[key]).expanduser())

    if opt.get("path") is not None:
        for key, val in opt["path"].items():
            if val is not None and ("resume_state" in key or "pretrain_network" in key):
                opt["path"][key] = str(Path(val).expanduser())

    if is_train:
        exp_root = (opt.get("path") or {}).get("experiments_root")
        if exp_root is None:
            exp_root = Path(root_path) / "experiments"
        exp_root = Path(exp_root) / opt["name"]
        if opt.get("path") is None:
            opt["path"] = {}
        opt["path"]["experiments_root"] = exp_root
        opt["path"]["models"] = exp_root / "models"
        opt["path"]["training_states"] = exp_root / "training_states"
        opt["path"]["log"] = exp_root
        opt["path"]["visualization"] = exp_root / "visualization"
        if "debug" in opt["name"]:
            if "val" in opt:
                opt["val"]["val_freq"] = 8
            opt.setdefault("logger", {})
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        opt.setdefault("path", {})
        results_root = opt["path"].get("results_root")
        if results_root is None:
            results_root = Path(root_path) / "experiments" / "results"
        results_root = Path(results_root) / opt["name"]
        opt["path"]["results_root"] = results_root
        opt["path"]["log"] = results_root
        opt["path"]["visualization"] = results_root

    _CACHE["opt"], _CACHE["args"] = opt, args
    return opt, args


def set_global_opt(opt: dict[str, Any] | None) -> None:
    """Install an already-built opt dict as the process-wide default (tests, bench, embedding)."""
    _CACHE["opt"] = opt


def global_opt() -> dict[str, Any] | None:
    """The cached opt; parsed lazily from ``sys.argv`` when it names a TOML via ``-opt``."""
    if _CACHE["opt"] is None and "-opt" in sys.argv:
        try:
            parse_options(str(Path(__file__).resolve().parents[2]), is_train=True)
        except (SystemExit, ValueError):
            return None
    return _CACHE["opt"]


def net_opt() -> tuple[int, bool]:
    """(scale, training) defaults for arch constructors (neosr/archs/arch_util.py:12-27).

    The reference bakes the TOML ``scale`` in as a default argument at import time; we read it
    lazily at construction time instead (same value, no import-order coupling).
    """
    opt = global_opt()
    if opt is None:
        return 4, True
    return int(opt.get("scale", 4)), "train" in opt.get("datasets", {})
