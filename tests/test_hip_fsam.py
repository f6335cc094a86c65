"""GPU parity of Friendly-SAM (`neosr_fsam_first_step` + the fused base step) through the C ABI: against
the reference optimizer fixture (tests/golden/fsam.npz) and, inside OUR `image` model with
`train.sam = "fsam"`, against the reference 5-iteration trajectory (tests/golden/step_fsam.npz).
Tolerance 1e-3 relative (observed ~1e-6)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import group, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.array(a))


def test_fsam_steps_vs_reference_fixture():
    from neosr_amd.hip.nets import arena_layout
    from neosr_amd.optimizers import AdamW, fsam

    fix = load_golden("fsam.npz")
    init = [T(fix[f"p0/{i}"]) for i in range(2)]
    offs, total = arena_layout(init)
    arena = torch.zeros(total, device=DEV)
    ps = []
    for t, off in zip(init, offs):
        arena[off: off + t.numel()].copy_(t.reshape(-1))
        ps.append(torch.nn.Parameter(arena[off: off + t.numel()].view(t.shape)))
    opt = fsam(ps, AdamW, rho=0.5, sigma=1, lmbda=0.9, adaptive=True, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.01)
    for step in range(1, 5):
        for i, p in enumerate(ps):
            p.grad = T(fix[f"g{step}/{i}"]).to(DEV)

        def closure(_it, step=step):
            for i, p in enumerate(ps):
                assert rel_err(p.detach(), T(fix[f"pert{step}/{i}"])) < 1e-5, ("perturbed", step, i)
                p.grad = T(fix[f"h{step}/{i}"]).to(DEV)

        opt.step(closure, step)
        for i, p in enumerate(ps):
            assert rel_err(p.detach(), T(fix[f"p{step}/{i}"])) < 1e-5, (step, i)
            assert rel_err(opt.state[p]["momentum"], T(fix[f"mom{step}/{i}"])) < 1e-5, (step, i)


def test_image_model_trajectory_fsam_vs_reference_fixture():
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden("step_fsam.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_fsam.toml")])
    model = build_model(opt)
    model.net_g.load_state_dict(group(fix, "init"))
    for it in range(1, 6):
        model.feed_data({"lq": T(fix[f"lq{it}"]), "gt": T(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert abs(log["l_g_pix"] - fix["log"][it - 1, 0]) < 1e-4 * fix["log"][it - 1, 0], it
        assert rel_err(model.output, T(fix[f"out{it}"])) < 1e-3, it
        sd = model.net_g.state_dict()
        for k, v in group(fix, f"w{it}").items():
            assert rel_err(sd[k], v) < 1e-3, (it, k)
    sd, esd = model.net_g.state_dict(), model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in group(fix, "final").items()) < 1e-3
    assert max(rel_err(esd[k], v) for k, v in group(fix, "ema").items() if k != "n_averaged") < 1e-3
    params = dict(model.net_g.named_parameters())
    for k, v in group(fix, "momentum").items():
        assert rel_err(model.sam_optimizer_g.state[params[k]]["momentum"], v) < 1e-3, k
    for k, v in group(fix, "base_exp_avg").items():
        assert rel_err(model.sam_optimizer_g.base_optimizer.state[params[k]]["exp_avg"], v) < 1e-3, k
