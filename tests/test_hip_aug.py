"""GPU parity of the batch augmentations through the C ABI (`neosr_resize_aa`, `neosr_box_blend`): the
antialiased resizes against ATen's outputs, every augmentation and 16 full `apply_augment` runs against
the reference with its random draws replayed.  Tolerance 1e-3 relative (observed ~1e-6)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from neosr_amd.data.draws import ReplayDraws
from tests.conftest import load_draws, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
AUGS = ["none", "mixup", "cutmix", "resizemix", "cutblur"]
PROB = [0.5, 0.1, 0.1, 0.1, 0.5]


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture(scope="module")
def fix():
    return load_golden("aug.npz")


@pytest.mark.parametrize("key,src,size,mode", [
    ("bilinear_up4", "in16", (64, 64), "bilinear"), ("bicubic_up4", "in16", (64, 64), "bicubic"),
    ("bicubic_down4", "in64", (16, 16), "bicubic"), ("bicubic_23x37", "in64", (23, 37), "bicubic"),
    ("bicubic_50x9", "in64", (50, 9), "bicubic")])
def test_resize_aa_vs_aten(fix, key, src, size, mode):
    from neosr_amd.data import augmentations as A

    y = A.resize_aa(T(fix[f"resize/{src}"]).to(DEV), size[0], size[1], mode, clamp=False)
    ref = T(fix[f"resize/{key}"])
    assert rel_err(y, ref) < 1e-5
    assert float((y.cpu() - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("name", ["mixup", "cutmix", "resizemix", "cutblur"])
def test_single_augmentations_vs_reference(fix, name):
    from neosr_amd.data import augmentations as A

    d = ReplayDraws(load_draws(fix, f"fn/{name}/draws"), DEV)
    gt, lq = getattr(A, name)(T(fix[f"fn/{name}/gt"]).to(DEV), T(fix[f"fn/{name}/lq"]).to(DEV), d)
    assert d.exhausted()
    assert rel_err(gt, T(fix[f"fn/{name}/gt_out"])) < 1e-5
    assert rel_err(lq, T(fix[f"fn/{name}/lq_out"])) < 1e-5


@pytest.mark.parametrize("k", range(16))
def test_apply_augment_replay_vs_reference(fix, k):
    from neosr_amd.data import augmentations as A

    d = ReplayDraws(load_draws(fix, f"run/{k}/draws"), DEV)
    gt, lq = A.apply_augment(T(fix[f"run/{k}/gt"]).to(DEV), T(fix[f"run/{k}/lq"]).to(DEV), d, scale=4, augs=AUGS,
                             prob=PROB)
    assert d.exhausted()
    assert rel_err(gt, T(fix[f"run/{k}/gt_out"])) < 1e-4
    assert rel_err(lq, T(fix[f"run/{k}/lq_out"])) < 1e-3
