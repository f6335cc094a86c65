"""pytest configuration: the `gpu` marker, repo-root imports and golden-fixture helpers."""

from __future__ import annotations

import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name: str) -> dict[str, np.ndarray]:
    with np.load(GOLDEN / name) as z:
        return {k: z[k] for k in z.files}


def group(fix: dict[str, np.ndarray], prefix: str, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    """sub-dict `prefix/...` of a fixture as tensors (insertion order = reference state_dict order)"""
    out = OrderedDict()
    for k, v in fix.items():
        if k.startswith(prefix + "/"):
            out[k[len(prefix) + 1 :]] = torch.from_numpy(np.array(v)).to(device)
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b||  (tolerance on per-tensor relative L2, not elementwise)"""
    a = a.detach().double().cpu().flatten()
    b = b.detach().double().cpu().flatten()
    denom = float(b.norm())
    return float((a - b).norm()) / (denom if denom > 0 else 1.0)


def load_draws(fix: dict[str, np.ndarray], prefix: str):
    """[(kind, value), ...] recorded by tests/golden/gen_golden_otf.py under `prefix/NNN`."""
    kinds = [str(k) for k in fix[f"{prefix}/kinds"]]
    out = []
    for i, kind in enumerate(kinds):
        v = fix[f"{prefix}/{i:03d}"]
        out.append((kind, str(v) if v.dtype.kind in "US" else v))
    return out


DEG_OPT = {  # tests/golden/golden_otf.toml [degradations]
    "resize_prob": [0.3, 0.4, 0.3], "resize_range": [0.5, 1.5], "gaussian_noise_prob": 0.5,
    "noise_range": [0, 2], "poisson_scale_range": [0.05, 0.25], "gray_noise_prob": 0.4,
    "jpeg_range": [40, 95], "second_blur_prob": 0.5, "resize_prob2": [0.3, 0.4, 0.3],
    "resize_range2": [0.3, 1.5], "gaussian_noise_prob2": 0.5, "noise_range2": [0, 2],
    "poisson_scale_range2": [0.05, 0.1], "gray_noise_prob2": 0.4, "jpeg_range2": [35, 95],
}
