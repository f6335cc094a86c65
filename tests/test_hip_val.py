"""GPU: `image.test()` (validation-time inference, whole image and partitioned) against
the reference's outputs on the same inputs and EMA weights (tests/golden/val.npz)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, ROOT, group, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,tile", [("whole", -1), ("tiled", 24), ("tiled_small", 24)])
def test_image_test_vs_reference_fixture(name, tile):
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options

    fix = load_golden("val.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_val.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    model.net_g_ema.load_state_dict(group(fix, "ema"))
    model.opt["val"]["tile"] = tile
    model.feed_data({"lq": torch.from_numpy(np.array(fix[f"{name}/lq"]))})
    model.test()
    ref = torch.from_numpy(np.array(fix[f"{name}/out"]))
    assert model.output.shape == ref.shape
    assert rel_err(model.output, ref) < 1e-3
    assert model.net_g.training
    vis = model.get_current_visuals()
    assert set(vis) == {"lq", "result"} and vis["result"].device.type == "cpu"


def test_device_prefetcher_stages_batches_in_order():
    from neosr_amd.data.prefetch_dataloader import CUDAPrefetcher, DevicePrefetcher

    assert CUDAPrefetcher is DevicePrefetcher
    data = [{"lq": torch.full((3, 8, 8), float(i)).pin_memory(), "path": f"im{i}"} for i in range(5)]
    pf = DevicePrefetcher(data, {})
    for rnd in range(2):
        seen = []
        while (b := pf.next()) is not None:
            assert b["lq"].is_cuda and isinstance(b["path"], str)
            seen.append(int(b["lq"].mean().item()))
        assert seen == list(range(5))
        pf.reset()


def test_validation_loop_metrics_images_and_best_record(tmp_path):
    """`image.validation` (image.py:785-922): feed_data -> test() per image, PSNR / SSIM means, best-so-far record, PNGs"""
    from neosr_amd import metrics as M
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options

    fix = load_golden("val.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_val.toml")])
    opt["val"]["tile"] = -1
    opt["val"]["metrics"] = {"psnr": {"type": "calculate_psnr", "crop_border": 4},
                             "ssim": {"type": "calculate_ssim", "crop_border": 4, "better": "higher"}}
    opt["path"]["visualization"] = tmp_path
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    model.net_g_ema.load_state_dict(group(fix, "ema"))
    lq = torch.from_numpy(np.array(fix["whole/lq"]))
    ref_out = torch.from_numpy(np.array(fix["whole/out"]))
    g = torch.Generator().manual_seed(4)
    gts = [(ref_out + 0.05 * torch.randn(ref_out.shape, generator=g)).clamp(0, 1) for _ in range(2)]

    class DS:
        opt = {"name": "valset", "type": "paired"}

    class Loader(list):
        dataset = DS()

    loader = Loader({"lq": lq, "gt": gt, "lq_path": [f"/x/im{i}.png"]} for i, gt in enumerate(gts))
    model.validation(loader, 7, None)
    exp = np.mean([M.calculate_psnr(M.tensor2img(ref_out), M.tensor2img(gt), crop_border=4) for gt in gts])
    assert abs(model.metric_results["psnr"] - exp) < 0.05          # the HIP output vs the reference output: 1e-3 rel
    assert 0.0 < model.metric_results["ssim"] <= 1.0
    assert model.best_metric_results["valset"]["psnr"] == {"better": "higher", "val": model.metric_results["psnr"], "iter": 7}
    assert (tmp_path / "im0" / "im0_7.png").exists() and (tmp_path / "im1" / "im1_7.png").exists()
    assert model.is_train and model.net_g.training and not hasattr(model, "lq")
