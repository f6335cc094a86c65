"""CPU tests of the host-side mirror of the reference's plugin surface: registries, TOML option
parsing (vs the reference's own parse of the same file), state-dict key/shape/initialisation parity."""

from __future__ import annotations

import json
from pathlib import Path

import pytest
import torch

from tests.conftest import GOLDEN, ROOT, group, load_golden


def test_registry_contract():
    from neosr_amd.utils.registry import Registry

    r = Registry("thing")

    @r.register()
    def foo():
        return 1

    class Bar:
        pass

    r.register(Bar, suffix="neosr")
    assert r.get("foo") is foo and r.get("Bar") is Bar and "foo" in r and "Bar_neosr" in r
    with pytest.raises(AssertionError):
        r.register(foo)
    with pytest.raises(KeyError, match="No object named 'nope'"):
        r.get("nope")
    assert set(r.keys()) == {"foo", "Bar_neosr"}


def test_plugins_registered_under_reference_names():
    from neosr_amd import ARCH_REGISTRY, LOSS_REGISTRY, MODEL_REGISTRY
    from neosr_amd.archs import build_network  # noqa: F401  (triggers the scan)
    from neosr_amd.archs import _import_archs
    from neosr_amd.losses import build_loss  # noqa: F401
    from neosr_amd.models import build_model  # noqa: F401

    _import_archs()
    assert {"esrgan", "compact"} <= set(ARCH_REGISTRY.keys())
    assert "L1Loss" in LOSS_REGISTRY and "image" in MODEL_REGISTRY


_OPT_CASES = [("compact", GOLDEN / "golden_compact.toml"), ("esrgan", GOLDEN / "golden_esrgan.toml")] + [
    (f"bench_{n}", ROOT / "options" / f"bench_{n}.toml")
    for n in ("compact", "esrgan", "esrgan_otf_gan", "swinir_medium", "hat_l_otf_gan")]


@pytest.mark.parametrize(("arch", "toml"), _OPT_CASES, ids=[c[0] for c in _OPT_CASES])
def test_parse_options_matches_reference_dump(arch, toml):
    """Our parse of a TOML == the reference's parse of the same file (tests/golden/opt_<name>.json, written by
    gen_golden.py / gen_golden_opts.py from neosr/utils/options.py:39-275): the two reduced golden configs and
    the five shipped option files (options/bench_*.toml = BASELINE configs[0..4])."""
    from neosr_amd.utils.options import parse_options

    opt, args = parse_options(str(ROOT), True, argv=["-opt", str(toml)])
    ref = json.loads((GOLDEN / f"opt_{arch}.json").read_text())

    def norm(o):
        if isinstance(o, dict):
            return {k: norm(v) for k, v in o.items()}
        if isinstance(o, Path):
            return "<path>/" + o.name
        if isinstance(o, (list, tuple)):
            return [norm(v) for v in o]
        return o

    got = norm(opt)
    got["num_gpu"] = ref["num_gpu"]  # device count of the machine that ran it
    assert got == ref
    assert args.launcher == "none" and opt["dist"] is False and opt["rank"] == 0


def test_parse_options_errors():
    from neosr_amd.utils.options import parse_options

    with pytest.raises(ValueError):
        parse_options(str(ROOT), True, argv=[])
    with pytest.raises(ValueError):
        parse_options(str(ROOT), True, argv=["-opt", "x.yml"])


def test_esrgan_state_dict_keys_shapes_and_seeded_init_match_reference():
    """Same constructor call order => same RNG consumption => identical seeded weights."""
    from neosr_amd.archs import build_network

    fix = load_golden("step_esrgan.npz")
    init = group(fix, "init")
    torch.manual_seed(1024)
    net = build_network({"type": "esrgan", "num_feat": 16, "num_block": 2, "num_grow_ch": 8, "scale": 4})
    sd = net.state_dict()
    assert list(sd.keys()) == list(init.keys())
    for k, v in init.items():
        assert sd[k].shape == v.shape, k
        assert torch.equal(sd[k], v), k


def test_compact_state_dict_keys_shapes_and_seeded_init_match_reference():
    from neosr_amd.archs import build_network

    fix = load_golden("step_compact.npz")
    init = group(fix, "init")
    torch.manual_seed(1024)
    net = build_network({"type": "compact", "num_feat": 16, "num_conv": 3, "upscale": 4})
    sd = net.state_dict()
    assert list(sd.keys()) == list(init.keys())
    for k, v in init.items():
        assert torch.equal(sd[k], v), k


def test_default_esrgan_param_count():
    from neosr_amd.archs import build_network

    net = build_network({"type": "esrgan", "scale": 4})
    assert sum(p.numel() for p in net.parameters()) == 16_697_987   # SURVEY §3.4 [probe]
    assert len(net.state_dict()) == 702
    net = build_network({"type": "compact", "upscale": 4})
    assert sum(p.numel() for p in net.parameters()) == 621_424


def test_flatten_parameters_preserves_values_and_order():
    from neosr_amd.archs import build_network
    from neosr_amd.hip.nets import flatten_parameters_

    torch.manual_seed(0)
    net = build_network({"type": "compact", "num_feat": 8, "num_conv": 2, "upscale": 4})
    before = {k: v.clone() for k, v in net.state_dict().items()}
    arena = flatten_parameters_(net)
    assert arena.numel() == sum(p.numel() for p in net.parameters())
    off = 0
    for k, p in net.named_parameters():
        assert torch.equal(p, before[k])
        assert p.data_ptr() == arena.data_ptr() + off * 4
        off += p.numel()
    assert flatten_parameters_(net) is arena  # idempotent


def test_unsupported_options_fail_loudly():
    from neosr_amd.models.image import image

    class Dummy(image):
        def __init__(self, opt):  # skip device work: only exercise the option screening
            self.opt = opt
            self.is_train = True

    opt = {"train": {"wavelet_guided": True, "optim_g": {"type": "adamw", "lr": 1e-4}}, "datasets": {"train": {}},
           "scale": 4}
    with pytest.raises(NotImplementedError, match="wavelet_guided"):
        Dummy(opt).init_training_settings()
    # SAM is supported, but not together with gradient accumulation (image.py:251-257)
    opt = {"train": {"sam": "fsam", "optim_g": {"type": "adamw", "lr": 1e-4}},
           "datasets": {"train": {"accumulate": 2}}, "scale": 4}
    with pytest.raises(NotImplementedError, match="accumulation"):
        Dummy(opt).init_training_settings()


def test_check_resume_rewrites_pretrain_paths_like_the_reference():
    """misc.check_resume (reference neosr/utils/misc.py:131-165) + train.load_resume_state on the
    reference-written state file (tests/golden/ckpt/2.state)."""
    from neosr_amd.utils.misc import check_resume, load_resume_state

    opt = {"name": "x", "network_g": {}, "network_d": {}, "auto_resume": False,
           "path": {"models": "/m", "resume_state": str(GOLDEN / "ckpt" / "2.state"),
                    "pretrain_network_g": "/old.pth", "param_key_g": "params_ema", "param_key_d": "params",
                    "ignore_resume_networks": ["network_d"]}}
    state = load_resume_state(opt)
    assert state["iter"] == 2 and state["epoch"] == 0 and len(state["optimizers"]) == 1
    assert opt["path"]["pretrain_network_g"] == Path("/m/net_g_2.pth")
    assert "pretrain_network_d" not in opt["path"]          # listed in ignore_resume_networks
    assert opt["path"]["param_key_g"] == "params" and opt["path"]["param_key_d"] == "params"
    opt2 = {"network_g": {}, "path": {"models": "/m"}}
    check_resume(opt2, 5)                                     # no resume_state: untouched
    assert opt2["path"] == {"models": "/m"}
    assert load_resume_state({"auto_resume": False, "path": {}}) is None
    # the state file holds what torch.optim / the reference's adan_sf put there
    g = state["optimizers"][0]["param_groups"][0]
    assert g["train_mode"] is True and g["betas"] == [0.98, 0.92, 0.987] and g["step"] == 2
    assert state["schedulers"][0]["milestones"] == {1: 1, 3: 1} and state["schedulers"][0]["last_epoch"] == 2


def test_master_only_log_read_leaves_chain_health_to_the_common_read(monkeypatch):
    """the @master_only checkpoint writers read the pending scalars on rank 0 alone.  That read must neither
    acknowledge the chain launches' slow-wait mark nor switch the launches off nor raise on the abort word — the next read
    every rank makes does (reference: neosr/models/base.py:281-475 writes checkpoints under @master_only)."""
    from collections import OrderedDict

    import neosr_amd.models.base as mb

    acted = []
    monkeypatch.setattr(mb.base, "_act_on_chain_health", lambda self, slow, st: acted.append((slow, st)))
    m = mb.base.__new__(mb.base)
    m.opt = {"dist": True, "rank": 0, "world_size": 2}
    m.log_dict = OrderedDict()
    m._log_work = None
    m._log_health = True
    m._iters_seen, m._log_iters, m.chain_slow_grace_iters = 30, 30, 20
    # scalars as the all-reduce left them (sums over 2 ranks; the division by world_size happens after `_log_work.wait()`,
    # skipped here): l_g_pix, l_g_total, then the two health words — slow mark on one rank, no abort
    m._log_dev = (["l_g_pix", "l_g_total"], torch.tensor([0.25, 0.5, 0.5, 0.0]))
    log = m.get_current_log(act_on_health=False)
    assert log["l_g_pix"] == 0.25 and acted == [] and m.chain_health_ok
    # the common read acts (with the words scaled back to rank counts)
    m._log_dev = (["l_g_pix", "l_g_total"], torch.tensor([0.25, 0.5, 0.5, 0.0]))
    m.get_current_log()
    assert acted == [(1.0, 0.0)]
    # an abort word seen by the lone read: nothing raised, but the caller is told not to write
    m._log_dev = (["l_g_pix", "l_g_total"], torch.tensor([0.25, 0.5, 0.0, 1.5]))
    m.get_current_log(act_on_health=False)
    assert not m.chain_health_ok and len(acted) == 1
    # NaN still raises on the lone read (the reference's per-iteration check, image.py:611-619)
    m._log_dev = (["l_g_pix", "l_g_total"], torch.tensor([0.25, float("nan"), 0.0, 0.0]))
    with pytest.raises(ValueError):
        m.get_current_log(act_on_health=False)
