"""GPU: checkpoint / resume wire format against files WRITTEN BY THE REFERENCE
(tests/golden/ckpt/net_g_2.pth, 2.state; tests/golden/gen_golden_ckpt.py): our `save()` produces the same
keys / values, and a model resumed from the reference's files continues like the reference did."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import GOLDEN, ROOT, group, load_golden, rel_err

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.array(a))


def _opt(tmp_path, extra=None):
    from neosr_amd.utils.options import parse_options

    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_ckpt.toml")])
    opt["path"]["models"] = str(tmp_path / "models")
    opt["path"]["training_states"] = str(tmp_path / "training_states")
    opt["path"].update(extra or {})
    return opt


def test_save_matches_reference_files(tmp_path):
    from neosr_amd.models import build_model

    fix = load_golden("ckpt.npz")
    model = build_model(_opt(tmp_path))
    model.net_g.load_state_dict(group(fix, "init"))
    for it in (1, 2):
        model.feed_data({"lq": T(fix[f"lq{it}"]), "gt": T(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        model.update_learning_rate(it, warmup_iter=-1)
    model.save(0, 2)
    sd = model.net_g.state_dict()
    assert max(rel_err(sd[k], v) for k, v in group(fix, "after_save").items()) < 1e-3
    ours = torch.load(tmp_path / "models" / "net_g_2.pth", map_location="cpu", weights_only=True)
    ref = torch.load(GOLDEN / "ckpt" / "net_g_2.pth", map_location="cpu", weights_only=True)
    assert list(ours.keys()) == list(ref.keys()) == ["params"]
    assert list(ours["params"].keys()) == list(ref["params"].keys())  # EMA weights, no module. / n_averaged
    for k, v in ref["params"].items():
        assert ours["params"][k].dtype == v.dtype and rel_err(ours["params"][k], v) < 1e-3, k
    so = torch.load(tmp_path / "training_states" / "2.state", map_location="cpu", weights_only=True)
    sr = torch.load(GOLDEN / "ckpt" / "2.state", map_location="cpu", weights_only=True)
    assert so["epoch"] == sr["epoch"] and so["iter"] == sr["iter"] == 2
    assert len(so["optimizers"]) == len(sr["optimizers"]) == 1 and so["schedulers"] == sr["schedulers"]
    go, gr = so["optimizers"][0]["param_groups"][0], sr["optimizers"][0]["param_groups"][0]
    assert set(gr) <= set(go)
    for k, v in gr.items():
        if isinstance(v, float):
            assert abs(go[k] - v) <= 1e-6 * abs(v) + 1e-12, k
        else:
            assert go[k] == v, k
    sto, str_ = so["optimizers"][0]["state"], sr["optimizers"][0]["state"]
    assert list(sto.keys()) == list(str_.keys())
    for i in (0, len(str_) - 1):
        assert set(str_[i]) <= set(sto[i])
        for k, v in str_[i].items():
            if torch.is_tensor(v) and v.numel() > 1:
                assert rel_err(sto[i][k].float(), v.float()) < 1e-3, (i, k)


def test_resume_from_reference_files(tmp_path):
    from neosr_amd.models import build_model
    from neosr_amd.utils.misc import load_resume_state

    fix = load_golden("ckpt.npz")
    opt = _opt(tmp_path, {"models": str(GOLDEN / "ckpt"), "resume_state": str(GOLDEN / "ckpt" / "2.state"),
                          "pretrain_network_g": "/nonexistent/ignored.pth"})
    state = load_resume_state(opt)
    assert state["iter"] == 2 and opt["path"]["pretrain_network_g"].name == str(fix["resume_pretrain_name"])
    model = build_model(opt)
    model.resume_training(state)
    sd = model.net_g.state_dict()
    for k, v in group(fix, "resumed").items():  # net_g restarts from the saved EMA weights
        assert torch.equal(sd[k].cpu(), v), k
    assert np.allclose(model.get_current_learning_rate(), fix["resumed_lr"], rtol=1e-12)
    for j, it in enumerate((3, 4)):
        model.feed_data({"lq": T(fix[f"lq{it}"]), "gt": T(fix[f"gt{it}"])})
        model.optimize_parameters(it)
        model.update_learning_rate(it, warmup_iter=-1)
        log = model.get_current_log()
        assert abs(log["l_g_pix"] - fix["log"][j, 0]) < 1e-4 * fix["log"][j, 0]
        assert rel_err(model.output, T(fix[f"out{it}"])) < 1e-3
        assert np.allclose(model.get_current_learning_rate(), fix[f"lr{it}"], rtol=1e-12)
    sd, esd = model.net_g.state_dict(), model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in group(fix, "final").items()) < 1e-3
    assert max(rel_err(esd[k], v) for k, v in group(fix, "ema").items() if k != "n_averaged") < 1e-3
    g0 = model.optimizer_g.param_groups[0]
    assert g0["step"] == int(fix["group"][0])
    assert abs(g0["weight_sum"] - fix["group"][1]) < 1e-9 * abs(fix["group"][1])
    assert abs(g0["lr"] - fix["group"][3]) < 1e-12
