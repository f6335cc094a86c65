"""MODEL_REGISTRY population + ``build_model`` (neosr/models/__init__.py:23-37)."""

from __future__ import annotations

import importlib
from copy import deepcopy
from pathlib import Path
from typing import Any

from neosr_amd.utils.misc import get_root_logger
from neosr_amd.utils.registry import MODEL_REGISTRY

__all__ = ["build_model"]

for _f in sorted(Path(__file__).resolve().parent.glob("*.py")):
    if _f.stem not in {"__init__"}:
        importlib.import_module(f"neosr_amd.models.{_f.stem}")


def build_model(opt: dict[str, Any]):
    opt = deepcopy(opt)
    model = MODEL_REGISTRY.get(opt["model_type"])(opt)
    get_root_logger().info(f"Using model [{model.__class__.__name__}].")
    return model
