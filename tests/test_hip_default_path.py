"""GPU: the path `bench.py` actually runs — esrgan at B >= 4 with every switch at its default (chain launches of
conv3x3_wino4_chain_kernel, F(4x4,3x3) forward / backward-data, RRDB-level F(4x4) weight gradients) — held directly
to the CPU oracle (oracle/neosr_oracle.py: rrdbnet_forward follows esrgan_arch.py:196-214, ImageTrainer follows
image.py:427-662), not only transitively through per-layer / per-launch comparisons.  The library's profiler is asked
how the trunk was launched, so a silent fall-back to F(2x2) / per-layer launches fails the test instead of passing it."""

from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Prof:
    """library profiler around a block: which kernels the conv classes ran as"""

    def __enter__(self):
        from neosr_amd import _C

        self.lib = _C.load()
        self.lib.neosr_prof_enable(1)
        return self

    def __exit__(self, *exc):
        from neosr_amd import _C

        lib = self.lib
        nc = lib.neosr_prof_num_classes()
        ms, ln, fl, by = (C.c_double * nc)(), (C.c_longlong * nc)(), (C.c_double * nc)(), (C.c_double * nc)()
        ex, algo = (C.c_double * nc)(), (C.c_longlong * (3 * nc))()
        ch_l, ch_n = (C.c_longlong * nc)(), (C.c_longlong * nc)()
        _C.check(lib.neosr_prof_collect_chain(ch_l, ch_n), "neosr_prof_collect_chain")
        _C.check(lib.neosr_prof_collect_exec(ex, algo), "neosr_prof_collect_exec")
        _C.check(lib.neosr_prof_collect(ms, ln, fl, by), "neosr_prof_collect")
        lib.neosr_prof_enable(0)
        self.chain_launches = [int(v) for v in ch_l]
        self.chain_layers = [int(v) for v in ch_n]
        self.by_algo = [[int(algo[3 * i + a]) for a in range(3)] for i in range(nc)]
        return False


def _assert_default_trunk(prof: _Prof, num_block: int) -> None:
    # class 0 = forward layers, 1 = backward-data layers, 2 = weight gradients (csrc/prof.h)
    assert prof.chain_launches[0] == num_block and prof.chain_launches[1] == num_block, prof.chain_launches
    assert prof.chain_layers[0] == 15 * num_block and prof.chain_layers[1] == 15 * num_block, prof.chain_layers
    for cls in (0, 1):      # every chain layer is an F(4x4,3x3) layer (algo 2)
        assert prof.by_algo[cls][2] >= 15 * num_block, (cls, prof.by_algo[cls])
    assert prof.by_algo[2][2] >= num_block, prof.by_algo[2]   # the RRDB-level weight gradients ran in the F(4x4) form
    assert prof.lib.neosr_conv_chain_status() == 0


def test_esrgan_default_path_vs_oracle():
    """Default RRDBNet (23 blocks, 64 feat) at B = 4, 64x64 LR — 64 workgroups per launch, the smallest batch that takes
    the default path of the bench (chain + F(4x4) + RRDB-level weight gradients): output and all 702 gradients vs the
    CPU oracle."""
    from neosr_amd.archs import build_network
    from oracle import neosr_oracle as orc

    torch.manual_seed(1024)
    net = build_network({"type": "esrgan", "scale": 4})
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.rand(4, 3, 64, 64)
    gt = torch.rand(4, 3, 256, 256)
    y_ref = orc.rrdbnet_forward(P, x, 4)
    orc.l1_loss(y_ref, gt).backward()
    net = net.to(DEV).train()
    with _Prof() as prof:
        y = net(x.to(DEV))
        F.l1_loss(y, gt.to(DEV)).backward()
        torch.cuda.synchronize()
    _assert_default_trunk(prof, 23)
    assert rel_err(y, y_ref) < 1e-4
    named = dict(net.named_parameters())
    errs = {k: rel_err(named[k].grad, P[k].grad) for k in P}
    worst = max(errs, key=errs.get)
    assert errs[worst] < 1e-3, (worst, errs[worst])


def test_image_step_default_path_vs_oracle():
    """Two `feed_data` + `optimize_parameters` iterations of the `image` model on options/bench_esrgan.toml (the
    headline option file; `num_block` 3, B = 4) against the oracle's ImageTrainer: loss, output, weights, EMA."""
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options, set_global_opt
    from oracle import neosr_oracle as orc

    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(ROOT / "options" / "bench_esrgan.toml")])
    opt["network_g"]["num_block"] = 3
    opt["datasets"]["train"]["batch_size"] = 4
    opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 1
    set_global_opt(opt)
    model = build_model(opt)
    init = OrderedDict((k, v.detach().cpu().clone()) for k, v in model.net_g.state_dict().items())
    og = opt["train"]["optim_g"]
    tr = orc.ImageTrainer(lambda P, x: orc.rrdbnet_forward(P, x, 4), init, lr=og["lr"], betas=tuple(og["betas"]),
                          weight_decay=og["weight_decay"], ema=opt["train"]["ema"], grad_clip=True)
    g = torch.Generator().manual_seed(11)
    with _Prof() as prof:
        for it in (1, 2):
            lq, gt = torch.rand(4, 3, 64, 64, generator=g), torch.rand(4, 3, 256, 256, generator=g)
            tr.feed_data(lq, gt)
            tr.optimize_parameters()
            model.feed_data({"lq": lq, "gt": gt})
            model.optimize_parameters(it)
            log = model.get_current_log()
            assert abs(log["l_g_pix"] - tr.log["l_g_pix"]) < 1e-4 * tr.log["l_g_pix"], it
            assert rel_err(model.output, tr.output) < 1e-3, it
        torch.cuda.synchronize()
    assert prof.chain_launches[0] == 6 and prof.chain_launches[1] == 6, prof.chain_launches
    assert prof.by_algo[0][2] >= 90 and prof.by_algo[1][2] >= 90 and prof.by_algo[2][2] >= 6, prof.by_algo[:3]
    assert prof.lib.neosr_conv_chain_status() == 0
    sd, esd = model.net_g.state_dict(), model.net_g_ema.state_dict()
    assert max(rel_err(sd[k], v) for k, v in tr.P.items()) < 1e-3
    assert max(rel_err(esd[k if k in esd else "module." + k], e) for k, e in zip(tr.names, tr.ema)) < 1e-3


def test_fast_matmul_tier_default_path_vs_oracle():
    """The `fast_matmul` tier (neosr_set_fast_matmul: two bf16 pieces per operand of the F(4x4,3x3) products, bf16 MFMA, fp32
    accumulation; reference train.py:168-173 / image.py:117-127) on the SAME launches as the default path — chain launches
    asserted through the profiler — against the CPU oracle at the tier's stated tolerance: 2e-3 on the output of the
    23-block net and 1e-2 on its gradients (fp32 path: 1e-4 / 1e-3; per layer the tier is ~1e-4 of the output scale,
    TF32 — what the reference switches on — ~5e-4).  It must also DIFFER from the fp32 run (the tier was taken), and
    switching back restores the fp32 bits (the weight images are re-packed for the mode)."""
    from neosr_amd import _C
    from neosr_amd.archs import build_network
    from oracle import neosr_oracle as orc

    torch.manual_seed(1024)
    net = build_network({"type": "esrgan", "scale": 4})
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x = torch.rand(4, 3, 64, 64)
    gt = torch.rand(4, 3, 256, 256)
    y_ref = orc.rrdbnet_forward(P, x, 4)
    orc.l1_loss(y_ref, gt).backward()
    net = net.to(DEV).train()
    runs = []
    try:
        for fast in (False, True, False):
            _C.set_fast_matmul(fast)
            net.zero_grad(set_to_none=True)
            with _Prof() as prof:
                y = net(x.to(DEV))
                F.l1_loss(y, gt.to(DEV)).backward()
                torch.cuda.synchronize()
            _assert_default_trunk(prof, 23)
            named = dict(net.named_parameters())
            errs = {k: rel_err(named[k].grad, P[k].grad) for k in P}
            worst = max(errs, key=errs.get)
            runs.append((y.detach().clone(), rel_err(y, y_ref), errs[worst], worst,
                         torch.cat([named[k].grad.flatten() for k in P])))
    finally:
        _C.set_fast_matmul(False)
    print("fast_matmul tier: output rel err %.2e (fp32 %.2e), worst gradient %.2e at %s (fp32 %.2e)"
          % (runs[1][1], runs[0][1], runs[1][2], runs[1][3], runs[0][2]))
    assert runs[0][1] < 1e-4 and runs[0][2] < 1e-3
    assert runs[1][1] < 2e-3 and runs[1][2] < 1e-2, (runs[1][1], runs[1][2], runs[1][3])
    assert not torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][0], runs[2][0]) and torch.equal(runs[0][4], runs[2][4])


def test_fast_matmul_and_amp_option_keys_select_the_tier():
    """`fast_matmul` / `use_amp` / `bfloat16` in an option file switch the tier on at model build; a model built afterwards
    without them is fp32 again (the switch is process-wide and every model build sets it from its own options)."""
    from neosr_amd import _C
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options, set_global_opt

    try:
        for keys in ({}, {"fast_matmul": True}, {"use_amp": True, "bfloat16": True}, {}):
            opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(ROOT / "options" / "bench_esrgan.toml")])
            opt["network_g"]["num_block"] = 1
            opt["datasets"]["train"]["batch_size"] = 4
            opt["dist"], opt["rank"], opt["world_size"], opt["num_gpu"] = False, 0, 1, 1
            opt.update(keys)
            set_global_opt(opt)
            model = build_model(opt)
            assert model.use_amp is False and bool(_C.FAST_MATMUL) == bool(keys), keys
            assert bool(_C.load().neosr_set_fast_matmul(int(bool(keys)))) == bool(keys)
            g = torch.Generator().manual_seed(3)
            model.feed_data({"lq": torch.rand(4, 3, 64, 64, generator=g), "gt": torch.rand(4, 3, 256, 256, generator=g)})
            model.optimize_parameters(1)
            assert model.get_current_log()["l_g_pix"] > 0
    finally:
        _C.set_fast_matmul(False)
