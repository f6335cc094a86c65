"""CPU: the band arithmetic of partitioned inference (neosr_amd/models/tiling.py) with the oracle's
esrgan forward as the network, against the reference's `image.test()` outputs (tests/golden/val.npz)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import neosr_oracle as orc
from tests.conftest import group, load_golden, rel_err


@pytest.mark.parametrize("name", ["tiled", "tiled_small"])
def test_tiled_inference_vs_reference_fixture(name):
    from neosr_amd.models.tiling import tiled_inference

    fix = load_golden("val.npz")
    P = {k.removeprefix("module."): v for k, v in group(fix, "ema").items() if k != "n_averaged"}
    lq = torch.from_numpy(np.array(fix[f"{name}/lq"]))
    with torch.no_grad():
        out = tiled_inference(lambda x: orc.rrdbnet_forward(P, x, 4), lq, 24, 4)
    assert rel_err(out, torch.from_numpy(np.array(fix[f"{name}/out"]))) < 1e-5


def test_whole_image_oracle_vs_reference_fixture():
    fix = load_golden("val.npz")
    P = {k.removeprefix("module."): v for k, v in group(fix, "ema").items() if k != "n_averaged"}
    with torch.no_grad():
        out = orc.rrdbnet_forward(P, torch.from_numpy(np.array(fix["whole/lq"])), 4)
    assert rel_err(out, torch.from_numpy(np.array(fix["whole/out"]))) < 1e-5
