"""GPU parity of the remaining optimizers (`neosr_optim_step`): `adan`, `adamw_sf`, `adamw_win` against
fixtures produced by the reference classes (tests/golden/optim.npz), `Adam` / `NAdam` against torch.optim on
CPU (the reference instantiates those torch classes directly, base.py:152-157).  Tolerance 1e-3 relative
(observed ~1e-6)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from tests.conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.array(a))


def _arena_params(init):
    from neosr_amd.hip.nets import arena_layout

    offs, total = arena_layout(init)
    arena = torch.zeros(total, device=DEV)
    ps = []
    for t, off in zip(init, offs):
        arena[off: off + t.numel()].copy_(t.reshape(-1))
        ps.append(torch.nn.Parameter(arena[off: off + t.numel()].view(t.shape)))
    return ps


CASES = {
    "adan_prox": ("adan", dict(lr=2e-3, betas=(0.98, 0.92, 0.99), weight_decay=0.02)),
    "adan_noprox": ("adan", dict(lr=2e-3, betas=(0.98, 0.92, 0.99), weight_decay=0.02, no_prox=True)),
    "adamw_sf": ("adamw_sf", dict(lr=2.5e-3, betas=(0.9, 0.999), weight_decay=0.01, warmup_steps=3)),
    "adamw_win": ("adamw_win", dict(lr=5e-4, weight_decay=0.02, acceleration_mode="win")),
    "adamw_win2": ("adamw_win", dict(lr=5e-4, weight_decay=0.02, acceleration_mode="win2")),
    "adamw_plain": ("adamw_win", dict(lr=5e-4, weight_decay=0.02, acceleration_mode="none")),
}


@pytest.mark.parametrize("tag", list(CASES))
def test_reference_optimizers_vs_fixture(tag):
    from neosr_amd import optimizers

    fix = load_golden("optim.npz")
    name, kw = CASES[tag]
    ps = _arena_params([T(fix[f"{tag}/p0/{i}"]) for i in range(2)])
    opt = getattr(optimizers, name)(ps, **kw)
    if hasattr(opt, "train") and name == "adamw_sf":
        opt.train()
    for step in range(1, 6):
        for i, p in enumerate(ps):
            p.grad = T(fix[f"{tag}/g{step}/{i}"]).to(DEV)
        opt.step()
        for i, p in enumerate(ps):
            assert rel_err(p, T(fix[f"{tag}/p{step}/{i}"])) < 1e-5, (step, i)
        if tag == "adamw_sf" and step == 3:
            opt.eval()
            assert all(rel_err(ps[i], T(fix[f"{tag}/p_eval/{i}"])) < 1e-5 for i in range(2))
            opt.train()


@pytest.mark.parametrize("name", ["Adam", "NAdam"])
def test_torch_optimizers_vs_torch_cpu(name):
    from neosr_amd import optimizers

    g = torch.Generator().manual_seed(23)
    init = [torch.randn(6, 5, generator=g), torch.randn(9, generator=g)]
    grads = [[torch.randn(t.shape, generator=g) for t in init] for _ in range(5)]
    kw = dict(lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)
    ref_p = [t.clone().requires_grad_(True) for t in init]
    ref = getattr(torch.optim, name)(ref_p, **kw)
    ps = _arena_params(init)
    opt = getattr(optimizers, name)(ps, **kw)
    for gs in grads:
        for p, rp, gg in zip(ps, ref_p, gs):
            p.grad, rp.grad = gg.to(DEV), gg.clone()
        opt.step()
        ref.step()
        for p, rp in zip(ps, ref_p):
            assert rel_err(p, rp) < 1e-5


@pytest.mark.parametrize("name", ["Adam", "NAdam"])
def test_torch_optimizer_state_resumes_bias_correction(name):
    """a `.state` written by torch.optim.Adam / NAdam (per-parameter `step`, NAdam's `mu_product`) is adopted:
    the resumed run continues the bias correction / momentum schedule instead of restarting at step 0, and the
    state written back carries `step` per parameter again (extra.py)."""
    from neosr_amd import optimizers

    g = torch.Generator().manual_seed(29)
    init = [torch.randn(6, 5, generator=g), torch.randn(9, generator=g)]
    grads = [[torch.randn(t.shape, generator=g) for t in init] for _ in range(6)]
    kw = dict(lr=1e-3, betas=(0.9, 0.99), weight_decay=0.01)
    ref_p = [t.clone().requires_grad_(True) for t in init]
    ref = getattr(torch.optim, name)(ref_p, **kw)
    for gs in grads[:3]:
        for rp, gg in zip(ref_p, gs):
            rp.grad = gg.clone()
        ref.step()
    ps = _arena_params([rp.detach() for rp in ref_p])
    opt = getattr(optimizers, name)(ps, **kw)
    opt.load_state_dict(ref.state_dict())
    for gs in grads[3:]:
        for p, rp, gg in zip(ps, ref_p, gs):
            p.grad, rp.grad = gg.to(DEV), gg.clone()
        opt.step()
        ref.step()
        for p, rp in zip(ps, ref_p):
            assert rel_err(p, rp) < 1e-5
    sd, rsd = opt.state_dict(), ref.state_dict()
    for i in sd["state"]:
        assert float(sd["state"][i]["step"]) == float(rsd["state"][i]["step"]) == 6.0
        if name == "NAdam":
            assert abs(float(sd["state"][i]["mu_product"]) - float(rsd["state"][i]["mu_product"])) < 1e-6


def test_fused_clip_and_ema_in_generic_step():
    """the model-level clip + EMA hooks work for the generic kernel exactly as for adamw"""
    from neosr_amd import optimizers
    from neosr_amd.hip.nets import arena_layout

    g = torch.Generator().manual_seed(5)
    init = [torch.randn(8, 4, generator=g), torch.randn(12, generator=g)]
    grads = [torch.randn(t.shape, generator=g) * 3 for t in init]
    ps = _arena_params(init)
    opt = optimizers.Adam(ps, lr=1e-2, betas=(0.9, 0.99))
    offs, total = arena_layout(init)
    ema = torch.zeros(total, device=DEV)
    for p, gg in zip(ps, grads):
        p.grad = gg.to(DEV)
    opt.set_clip(1.0)
    opt.set_ema(ema, 0.9, first=True)
    opt.step()
    norm = torch.sqrt(sum((gg.double() ** 2).sum() for gg in grads))
    coef = min(1.0, 1.0 / (float(norm) + 1e-6))
    ref_p = [t.clone().requires_grad_(True) for t in init]
    ref = torch.optim.Adam(ref_p, lr=1e-2, betas=(0.9, 0.99))
    for rp, gg in zip(ref_p, grads):
        rp.grad = gg * coef
    ref.step()
    for p, rp, off in zip(ps, ref_p, offs):
        assert rel_err(p, rp) < 1e-5
        assert rel_err(ema[off: off + rp.numel()], rp.detach().flatten()) < 1e-5
