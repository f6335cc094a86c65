"""ARCH_REGISTRY population + ``build_network`` (neosr/archs/__init__.py:14-34)."""

from __future__ import annotations

import importlib
from copy import deepcopy
from pathlib import Path
from typing import Any

from neosr_amd.utils.misc import get_root_logger
from neosr_amd.utils.registry import ARCH_REGISTRY

__all__ = ["build_network"]

_loaded = False


def _import_archs() -> None:
    global _loaded
    if _loaded:
        return
    for f in sorted(Path(__file__).resolve().parent.glob("*_arch.py")):
        importlib.import_module(f"neosr_amd.archs.{f.stem}")
    _loaded = True


def build_network(opt: dict[str, Any]):
    """``opt`` is a ``[network_g]`` / ``[network_d]`` table: pops ``type`` and splats the rest."""
    _import_archs()
    opt = deepcopy(opt)
    network_type = opt.pop("type")
    net = ARCH_REGISTRY.get(network_type)(**opt)
    get_root_logger().info(f"Using network [{net.__class__.__name__}].")
    return net
