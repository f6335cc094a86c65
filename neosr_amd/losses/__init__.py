"""LOSS_REGISTRY population + ``build_loss`` (neosr/losses/__init__.py:25-39)."""

from __future__ import annotations

import importlib
from copy import deepcopy
from pathlib import Path
from typing import Any

from neosr_amd.utils.misc import get_root_logger
from neosr_amd.utils.registry import LOSS_REGISTRY

__all__ = ["build_loss"]

for _f in sorted(Path(__file__).resolve().parent.glob("*_loss.py")):
    importlib.import_module(f"neosr_amd.losses.{_f.stem}")


def build_loss(opt: dict[str, Any]):
    opt = deepcopy(opt)
    loss_type = opt.pop("type")
    loss = LOSS_REGISTRY.get(loss_type)(**opt)
    get_root_logger().info(f"Loss [{loss.__class__.__name__}] enabled.")
    return loss
