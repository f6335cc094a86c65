"""The opt-in direct hand-off of parameter gradients (neosr_amd/hip/nets.py: direct_param_grads) against the autograd
path: same bits in `.grad`, in the weights and in the EMA copy after training steps; accumulation; outside the scope
nothing changes."""

from __future__ import annotations

from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _net():
    from neosr_amd.archs.compact_arch import compact

    torch.manual_seed(7)
    net = compact(num_feat=32, num_conv=3, upscale=2).cuda().train()
    net.flat_parameters()
    return net


def test_direct_grads_equal_autograd_grads_and_accumulate():
    from neosr_amd.hip import nets

    net = _net()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 32, 32, generator=g).cuda()
    t = torch.rand(2, 3, 64, 64, generator=g).cuda()
    (net(x) - t).abs().mean().backward()
    ref = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    with nets.direct_param_grads():
        y = net(x)
        assert y.grad_fn is not None and len(y.grad_fn.next_functions) == 2   # (x, anchor): no parameter edges
        (y - t).abs().mean().backward()
        got = [p.grad for p in net.parameters()]
        assert all(torch.equal(a, b) for a, b in zip(got, ref))
        flat = nets.flat_grad_of(list(net.parameters()))
        assert flat is not None and flat.data_ptr() == net.__dict__["_neosr_direct"].flat.data_ptr()
        # a second backward with `.grad` set accumulates (g + g)
        (net(x) - t).abs().mean().backward()
        assert all(torch.equal(p.grad, 2 * b) for p, b in zip(net.parameters(), ref))
        # the next step re-uses the persistent arena
        for p in net.parameters():
            p.grad = None
        (net(x) - t).abs().mean().backward()
        assert all(torch.equal(p.grad, b) for p, b in zip(net.parameters(), ref))
        # a frozen parameter, no_grad and eval mode take the autograd path
        first = next(net.parameters())
        first.requires_grad_(False)
        assert len(net(x).grad_fn.next_functions) > 2
        first.requires_grad_(True)
        with torch.no_grad():
            assert net(x).grad_fn is None
    # outside the scope: one edge per parameter again
    assert len(net(x).grad_fn.next_functions) == 1 + len(list(net.parameters()))


def _train(direct: bool, steps: int = 3):
    from neosr_amd.hip import nets
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options, set_global_opt

    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(ROOT / "options" / "bench_compact.toml")])
    opt["datasets"]["train"]["patch_size"] = 32
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 1
    set_global_opt(opt)
    env, nets._DIRECT_ENV = nets._DIRECT_ENV, (None if direct else "0")  # noqa: SLF001
    try:
        torch.manual_seed(1024)
        model = build_model(opt)
        g = torch.Generator().manual_seed(5)
        logs = []
        for it in range(1, steps + 1):
            lq, gt = torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 3, 128, 128, generator=g)
            model.feed_data({"lq": lq, "gt": gt})
            model.optimize_parameters(it)
            logs.append(model.get_current_log()["l_g_total"])
        used = "_neosr_direct" in model.net_g.__dict__
        return model.net_g.flat_parameters().clone(), model.net_g_ema.arena().clone(), logs, used
    finally:
        nets._DIRECT_ENV = env  # noqa: SLF001


def test_compact_training_steps_bit_identical_with_and_without_direct_grads():
    w1, e1, l1, used1 = _train(True)
    w0, e0, l0, used0 = _train(False)
    assert used1 and not used0
    assert l1 == l0
    assert torch.equal(w1, w0) and torch.equal(e1, e0)
