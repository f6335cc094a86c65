"""GPU parity of the fused Schedule-Free Adan step (`neosr_adan_sf_step`, `neosr_lerp`) through the C ABI:
against the reference optimizer fixture (tests/golden/adan_sf.npz) and, inside OUR `image` model,
against the reference 4-iteration trajectory (tests/golden/step_adan.npz).  Tolerance 1e-3 relative
(observed ~1e-6)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import group, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("tag", ["sf", "plain"])
def test_adan_sf_kernel_vs_reference_fixture(tag):
    from neosr_amd.optimizers import adan_sf

    fix = load_golden("adan_sf.npz")
    shapes = [fix[f"{tag}/p0/{i}"].shape for i in range(2)]
    from neosr_amd.hip.nets import arena_layout

    init = [T(fix[f"{tag}/p0/{i}"]) for i in range(2)]
    offs, total = arena_layout(init)  # 16-byte aligned starts: 35 elements -> the second tensor starts at 36
    arena = torch.zeros(total, device=DEV)
    ps = []
    for t, off in zip(init, offs):
        arena[off: off + t.numel()].copy_(t.reshape(-1))
        ps.append(torch.nn.Parameter(arena[off: off + t.numel()].view(t.shape)))
    opt = adan_sf(ps, lr=2e-3, betas=(0.98, 0.92, 0.987), weight_decay=0.02, warmup_steps=3,
                  schedule_free=tag == "sf")
    for step in range(1, 6):
        for i, p in enumerate(ps):
            p.grad = T(fix[f"{tag}/g{step}/{i}"]).to(DEV)
        opt.step()
        for i, p in enumerate(ps):
            assert rel_err(p, T(fix[f"{tag}/p{step}/{i}"])) < 1e-5, (step, i)
        if tag == "sf" and step == 3:
            opt.eval()
            assert all(rel_err(ps[i], T(fix[f"{tag}/p_eval/{i}"])) < 1e-5 for i in range(2))
            opt.train()
            assert all(rel_err(ps[i], T(fix[f"{tag}/p_train/{i}"])) < 1e-5 for i in range(2))
    for i, p in enumerate(ps):
        for k in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "neg_pre_grad"):
            assert rel_err(opt.state[p][k], T(fix[f"{tag}/state/{k}/{i}"])) < 1e-4, k
    g0 = opt.param_groups[0]
    assert g0["step"] == int(fix[f"{tag}/group"][0])
    if tag == "sf":
        assert abs(g0["weight_sum"] - fix[f"{tag}/group"][1]) < 1e-12


def test_image_model_trajectory_adan_sf_vs_reference_fixture():
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden("step_adan.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_adan.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    for it in range(1, 5):
        model.feed_data({"lq": T(fix[f"lq{it}"]), "gt": T(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert abs(log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-4 * fix["log"][it - 1, 0]
        assert rel_err(model.output, T(fix[f"out{it}"])) < 1e-3
        if it == 2:
            model.optimizer_g.eval()
            sd = model.net_g.state_dict()
            assert max(rel_err(sd[k], v) for k, v in group(fix, "eval2").items()) < 1e-3
            model.optimizer_g.train()
    sd, esd = model.net_g.state_dict(), model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in group(fix, "final").items()) < 1e-3
    assert max(rel_err(esd[k], v) for k, v in group(fix, "ema").items() if k != "n_averaged") < 1e-3
    st = model.optimizer_g.state
    named = dict(model.net_g.named_parameters())
    for kind in ("exp_avg", "exp_avg_sq", "exp_avg_diff", "z", "neg_pre_grad"):
        for name, v in group(fix, f"optstate/{kind}").items():
            assert rel_err(st[named[name]][kind], v) < 1e-3, (kind, name)
