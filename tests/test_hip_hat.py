"""GPU parity tests of the HAT path through the C ABI: the streaming 16x16-window attention kernels
(self and overlapping, forward and backward), channel attention / CAB, HAB and OCAB blocks and whole
nets against fixtures produced by the reference, `hat_l` forward, and the `image` model trajectory
with network_g = hat_s.  Tolerance 1e-3 relative (per-tensor ||d||/||ref||), observed ~1e-6.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import group, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.array(a))


def _self_attention_oracle(qkv, table, B, H, W, C, heads, shift, scale):
    from oracle import hat_oracle as ho
    from oracle import swinir_oracle as so

    x = torch.roll(qkv, (-shift, -shift), (1, 2)) if shift else qkv
    xw = so.window_partition(x, 16).view(-1, 256, 3 * C)
    q, k, v = xw.reshape(-1, 256, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = (q * scale) @ k.transpose(-2, -1)
    bias = table[ho.rpi_sa(16).view(-1)].view(256, 256, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift:
        m = so.calculate_mask(H, W, 16, shift).to(attn.dtype)
        attn = (attn.view(B, m.shape[0], heads, 256, 256) + m[None, :, None]).view(-1, heads, 256, 256)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, 16, 16, C)
    o = so.window_reverse(o, 16, H, W)
    return torch.roll(o, (shift, shift), (1, 2)) if shift else o


def _overlap_attention_oracle(qkv, table, B, H, W, C, heads, scale):
    from oracle import hat_oracle as ho
    from oracle import swinir_oracle as so

    q = qkv[..., :C]
    kv = qkv[..., C:].permute(0, 3, 1, 2)  # b, 2c, h, w
    qw = so.window_partition(q, 16).view(-1, 256, C)
    kvw = F.unfold(kv, kernel_size=(24, 24), stride=16, padding=4)
    nw = kvw.shape[-1]
    kvw = kvw.view(B, 2, C, 576, nw).permute(1, 0, 4, 3, 2).reshape(2, B * nw, 576, C)
    d = C // heads
    qh = qw.reshape(-1, 256, heads, d).permute(0, 2, 1, 3)
    kh = kvw[0].reshape(-1, 576, heads, d).permute(0, 2, 1, 3)
    vh = kvw[1].reshape(-1, 576, heads, d).permute(0, 2, 1, 3)
    attn = (qh * scale) @ kh.transpose(-2, -1)
    bias = table[ho.rpi_oca(16, 0.5).view(-1)].view(256, 576, -1).permute(2, 0, 1)
    attn = (attn + bias.unsqueeze(0)).softmax(-1)
    o = (attn @ vh).transpose(1, 2).reshape(-1, 16, 16, C)
    return so.window_reverse(o, 16, H, W)


@pytest.mark.parametrize("shift", [0, 8])
# (54 / 6: an odd head size — the token rows then take the 4-byte form of the raw-buffer loads, csrc/attn_rows.h)
@pytest.mark.parametrize("B,H,W,C,heads", [(2, 32, 48, 60, 6), (1, 16, 16, 24, 2), (1, 32, 32, 180, 6), (1, 32, 16, 54, 6)])
def test_flash_self_attention_fwd_bwd_vs_oracle(B, H, W, C, heads, shift):
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(B * H + C + shift)
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    table = torch.randn(31 * 31, heads, generator=g) * 0.5
    r = torch.randn(B, H, W, C, generator=g)
    scale = (C // heads) ** -0.5
    a, t = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref = _self_attention_oracle(a, t, B, H, W, C, heads, shift, scale)
    (ref * r.double()).sum().backward()
    a2, t2 = qkv.to(DEV).requires_grad_(True), table.to(DEV).requires_grad_(True)
    got = tr.flash_window_attention(a2, t2, heads, 16, shift, scale)
    assert rel_err(got, ref) < 1e-5
    (got * r.to(DEV)).sum().backward()
    assert rel_err(a2.grad, a.grad) < 1e-5
    assert rel_err(t2.grad, t.grad) < 1e-5


@pytest.mark.parametrize("B,H,W,C,heads", [(2, 32, 48, 60, 6), (1, 16, 16, 24, 2), (1, 48, 32, 180, 6), (1, 16, 32, 54, 6)])
def test_flash_overlapping_attention_fwd_bwd_vs_oracle(B, H, W, C, heads):
    """zero-padded 24x24 key windows (nn.Unfold), negative-index wrap of rpi_oca, fold of dK / dV"""
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(B * H + C)
    qkv = torch.randn(B, H, W, 3 * C, generator=g)
    table = torch.randn(39 * 39, heads, generator=g) * 0.5
    r = torch.randn(B, H, W, C, generator=g)
    scale = (C // heads) ** -0.5
    a, t = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref = _overlap_attention_oracle(a, t, B, H, W, C, heads, scale)
    (ref * r.double()).sum().backward()
    a2, t2 = qkv.to(DEV).requires_grad_(True), table.to(DEV).requires_grad_(True)
    got = tr.flash_window_attention(a2, t2, heads, 24, 0, scale)
    assert rel_err(got, ref) < 1e-5
    (got * r.to(DEV)).sum().backward()
    assert rel_err(a2.grad, a.grad) < 1e-5
    assert rel_err(t2.grad, t.grad) < 1e-5


def test_flash_attention_bwd_is_deterministic():
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(9)
    qkv, table = torch.randn(2, 32, 32, 540, generator=g).to(DEV), torch.randn(1521, 6, generator=g).to(DEV)
    r = torch.randn(2, 32, 32, 180, generator=g).to(DEV)
    outs = []
    for _ in range(2):
        a, t = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
        (tr.flash_window_attention(a, t, 6, 24, 0, 30 ** -0.5) * r).sum().backward()
        outs.append((a.grad.clone(), t.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,H,W,C,heads,shift,ws", [(2, 32, 48, 180, 6, 8, 16), (1, 64, 64, 60, 6, 0, 16), (3, 16, 24, 24, 2, 4, 8)])
def test_flash_self_attention_one_pass_backward_is_bit_identical_to_two_pass(B, H, W, C, heads, shift, ws):
    """`neosr_set_fattn_fused`: the one-pass backward (S / dP / P / dS of a tile formed once, dQ + dK + dV from them) against
    the two recompute passes it replaces — same accumulation orders, so dqkv and the bias-table gradient must be equal
    bit for bit; the vs-oracle tests above run on the one-pass kernel (the default)."""
    from neosr_amd import _C
    from neosr_amd.hip import transformer as tr

    lib = _C.load()
    g = torch.Generator().manual_seed(B * H + C + shift)
    qkv = torch.randn(B, H, W, 3 * C, generator=g).to(DEV)
    table = (torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5).to(DEV)
    r = torch.randn(B, H, W, C, generator=g).to(DEV)
    outs = []
    prev = lib.neosr_set_fattn_fused(1)
    try:
        for fused in (1, 0, 1):
            lib.neosr_set_fattn_fused(fused)
            a, t = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
            (tr.flash_window_attention(a, t, heads, ws, shift, (C // heads) ** -0.5, ws) * r).sum().backward()
            outs.append((a.grad.clone(), t.grad.clone()))
    finally:
        lib.neosr_set_fattn_fused(prev)
    for ga, gt in outs[1:]:
        assert torch.equal(ga, outs[0][0]) and torch.equal(gt, outs[0][1])


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_cab_vs_reference_fixture():
    from neosr_amd.archs.hat_arch import CAB

    fix = load_golden("hat_prims.npz")
    cab = CAB(24, 3, 6)
    cab.load_state_dict(group(fix, "cab/p"))
    cab = cab.to(DEV)
    x = _nhwc(T(fix["cab/x"])).to(DEV).requires_grad_(True)
    y = cab(x, None, 1.0)
    (y * _nhwc(T(fix["cab/r"])).to(DEV)).sum().backward()
    assert rel_err(y, _nhwc(T(fix["cab/y"]))) < 1e-4
    assert rel_err(x.grad, _nhwc(T(fix["cab/gx"]))) < 1e-3
    named = dict(cab.named_parameters())
    for k, g in group(fix, "cab/g").items():
        assert rel_err(named[k].grad, g) < 1e-3, k


@pytest.mark.parametrize("pre", ["hab_s0", "hab_s8", "ocab"])
def test_hab_ocab_blocks_vs_reference_fixture(pre):
    from neosr_amd.archs.hat_arch import HAB, OCAB

    fix = load_golden("hat_prims.npz")
    if pre == "ocab":
        blk = OCAB(24, (32, 48), 16, 0.5, 2, mlp_ratio=2)
    else:
        blk = HAB(24, (32, 48), 2, 16, int(pre[-1]), 3, 6, 0.01, 2)
    blk.load_state_dict(group(fix, f"{pre}/p"))
    blk = blk.to(DEV).train()
    x = T(fix[f"{pre}/x"]).view(2, 32, 48, 24).to(DEV).requires_grad_(True)
    y = blk(x)
    (y * T(fix[f"{pre}/r"]).view(2, 32, 48, 24).to(DEV)).sum().backward()
    assert rel_err(y.view(2, -1, 24), T(fix[f"{pre}/y"])) < 1e-4
    assert rel_err(x.grad.view(2, -1, 24), T(fix[f"{pre}/gx"])) < 1e-3
    named = dict(blk.named_parameters())
    worst = max((rel_err(named[k].grad, g), k) for k, g in group(fix, f"{pre}/g").items())
    assert worst[0] < 1e-3, worst


def test_tiny_hat_net_vs_reference_fixture():
    from neosr_amd.archs.hat_arch import hat

    fix = load_golden("hat_net.npz")
    net = hat(img_size=32, embed_dim=24, depths=(2,), num_heads=(2,), window_size=16, compress_ratio=3,
              squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle", upscale=4)
    assert list(net.state_dict().keys()) == [str(k) for k in fix["keys"]]
    net.load_state_dict(group(fix, "p"), strict=False)
    net = net.to(DEV).train()
    x = T(fix["x"]).to(DEV).requires_grad_(True)
    y = net(x)
    (y * T(fix["r"]).to(DEV)).sum().backward()
    assert rel_err(y, T(fix["y"])) < 1e-4
    assert rel_err(x.grad, T(fix["gx"])) < 1e-3
    named = dict(net.named_parameters())
    worst = max((rel_err(named[k].grad, g), k) for k, g in group(fix, "g").items())
    assert worst[0] < 1e-3, worst


def test_tiny_hat_window8_vs_reference_fixture():
    """hat with window_size 8: HAB on 8x8 windows (shift 4, all nine mask regions), OCAB with 12x12 zero-padded key windows
    (the <8, 8> / <8, 12> instantiations of the streaming attention kernels), forward + every gradient vs the reference"""
    from neosr_amd.archs.hat_arch import hat

    fix = load_golden("hat_w8.npz")
    net = hat(img_size=16, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=8, compress_ratio=3,
              squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle", upscale=4)
    assert list(net.state_dict().keys()) == [str(k) for k in fix["keys"]]
    net.load_state_dict(group(fix, "p"), strict=False)
    net = net.to(DEV).train()
    x = T(fix["x"]).to(DEV).requires_grad_(True)
    y = net(x)
    (y * T(fix["r"]).to(DEV)).sum().backward()
    assert rel_err(y, T(fix["y"])) < 1e-4
    assert rel_err(x.grad, T(fix["gx"])) < 1e-3
    named = dict(net.named_parameters())
    worst = max((rel_err(named[k].grad, g), k) for k, g in group(fix, "g").items())
    assert worst[0] < 1e-3, worst
    with pytest.raises(Exception, match="window_size"):
        hat(img_size=16, embed_dim=24, depths=(2,), num_heads=(2,), window_size=12, upsampler="pixelshuffle", upscale=4)


def test_hat_l_forward_b1_vs_reference_fixture():
    """BASELINE configs[4] generator at full size: seeded init identical to the reference, forward at 64x64 LR"""
    from neosr_amd.archs import hat_arch as A
    from neosr_amd.utils import options

    fix = load_golden("hat_l_fwd.npz")
    options.set_global_opt({"manual_seed": 1024, "rank": 0, "scale": 4, "datasets": {"train": {}}})
    try:
        torch.manual_seed(1024)
        net = A.hat_l(upscale=4)
    finally:
        options.set_global_opt(None)
    s = np.array([float(v.double().sum()) for v in net.state_dict().values()])
    np.testing.assert_allclose(s, fix["init_sum"], rtol=1e-6, atol=1e-6)
    net = net.to(DEV).eval()
    with torch.no_grad():
        y = net(T(fix["x"]).to(DEV))
    assert rel_err(y, T(fix["y"])) < 1e-4


def test_image_model_trajectory_hat_s_vs_reference_fixture():
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import parse_options
    from tests.conftest import GOLDEN, ROOT

    fix = load_golden("step_hat.npz")
    opt, _ = parse_options(str(ROOT), True, argv=["-opt", str(GOLDEN / "golden_hat.toml")])
    model = build_model(opt)
    s = np.array([float(v.double().sum()) for v in model.net_g.state_dict().values()])
    np.testing.assert_allclose(s, fix["init/sum"], rtol=1e-5, atol=1e-5)
    for it in (1, 2):
        model.feed_data({"lq": T(fix[f"it{it}/lq"]), "gt": T(fix[f"it{it}/gt"])})
        model.optimize_parameters(it)
        log = model.get_current_log()
        ref = float(fix[f"it{it}/log/l_g_pix"])
        assert abs(log["l_g_pix"] - ref) < 1e-4 * ref
        assert rel_err(model.output, T(fix[f"it{it}/output"])) < 1e-3
    sd = model.net_g.state_dict()
    for k in [f for f in fix if f.startswith("final/w/")]:
        assert rel_err(sd[k[len("final/w/"):]], T(fix[k])) < 1e-3, k


@pytest.mark.parametrize("C,Cs,rows,B", [(180, 6, 4096, 3), (60, 4, 1024, 3), (6, 2, 100, 3), (10, 2, 77, 70)])
def test_channel_gate_vector_and_scalar_forms_vs_torch(C, Cs, rows, B):
    """ChannelAttention + HAB's gated residual (hat_arch.py:15-37, 347) through the side kernels of csrc/cab.hip: channel
    counts that are multiples of 4 take the 16-byte forms (whole-row pooling pass, quad gate scale / gradient), the others
    the scalar forms; more than 64 samples (the B = 70 case) take the gate backward's unstaged path.  Reference: the same
    formulas in torch float64."""
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(C + rows)
    alpha = 0.01
    H, W = (rows // 64, 64) if rows % 64 == 0 else (1, rows)
    y = torch.randn(B, H, W, C, generator=g)
    res = torch.randn(B, H, W, C, generator=g)
    w1, b1 = torch.randn(Cs, C, generator=g) * 0.3, torch.randn(Cs, generator=g) * 0.1
    w2, b2 = torch.randn(C, Cs, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1
    go = torch.randn(B, H, W, C, generator=g)

    def ref(y, w1, b1, w2, b2, res):
        pooled = y.mean(dim=(1, 2))
        attn = torch.sigmoid(torch.relu(pooled @ w1.t() + b1) @ w2.t() + b2)
        return res + alpha * y * attn[:, None, None, :]

    leaves = [t.double().requires_grad_(True) for t in (y, w1, b1, w2, b2, res)]
    out_ref = ref(*leaves)
    grads_ref = torch.autograd.grad(out_ref, leaves, go.double())
    dl = [t.to(DEV).requires_grad_(True) for t in (y, w1, b1, w2, b2, res)]
    out = tr.ChannelGate.apply(dl[0], dl[1], dl[2], dl[3], dl[4], dl[5], alpha)
    grads = torch.autograd.grad(out, dl, go.to(DEV))
    assert rel_err(out, out_ref) < 1e-5
    for a, b in zip(grads, grads_ref):
        assert rel_err(a, b) < 1e-4, (a.shape, rel_err(a, b))


@pytest.mark.parametrize("n,off", [(60 * 4096, 0), (1003, 0), (4096, 1)])
def test_gelu_and_leaky_relu_vector_and_scalar_forms(n, off):
    """exact-erf GELU (hat_arch.py:46) and the leaky-ReLU / add passes of hip/layers.py: sizes that are multiples of 4 on
    16-byte aligned storage take the float4 kernels, an odd size or a view that starts one float into its storage the scalar
    ones.  GELU's erf is the A&S 7.1.26 form of the GEMM epilogues (|error| <= 1.5e-7)."""
    from neosr_amd.hip import layers as L
    from neosr_amd.hip import transformer as tr

    g = torch.Generator().manual_seed(n + off)
    x = (torch.randn(n + off, generator=g) * 2.0).to(DEV)[off:].requires_grad_(True)
    go = torch.randn(n + off, generator=g).to(DEV)[off:]
    xd = x.detach().cpu().double().requires_grad_(True)
    y_ref = torch.nn.functional.gelu(xd)
    (gx_ref,) = torch.autograd.grad(y_ref, xd, go.cpu().double())
    y = tr.Gelu.apply(x)
    (gx,) = torch.autograd.grad(y, x, go)
    assert rel_err(y, y_ref) < 1e-6 and rel_err(gx, gx_ref) < 1e-6
    lr = L.LeakyReLU.apply(x.detach(), 0.2)
    assert torch.equal(lr.cpu(), torch.nn.functional.leaky_relu(x.detach().cpu(), 0.2))


@pytest.mark.parametrize("B,H,W,K,N", [(4, 64, 64, 180, 60), (2, 32, 48, 60, 180)])
def test_conv_gelu_epilogues_are_bit_identical_to_the_elementwise_pass(B, H, W, K, N):
    """HAT's CAB (hat_arch.py:62-66: conv -> GELU -> conv): GELU in the F(4x4,3x3) convolution's epilogue with the
    pre-activation as second output (`act = NEOSR_ACT_GELU`, `out2`), and GELU'(pre-activation) in the backward-data epilogue of
    the convolution behind it (`out_mask_gelu`) — the block plans use both (csrc/blocks.hip) — against the convolution followed by
    `neosr_gelu`: the same expressions (csrc/gelu.h), so the results must agree bit for bit."""
    from neosr_amd import _C
    from neosr_amd.hip import ops

    lib = _C.load()
    g = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, H, W, K, generator=g).to(DEV)
    w = (torch.randn(N, K, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    w4 = ops.conv3x3_pack_wino4(w)
    # forward
    u_ref = ops.conv3x3(x, w, b, w_wino4=w4)
    t_ref = torch.empty_like(u_ref)
    _C.check(lib.neosr_gelu(u_ref.data_ptr(), None, t_ref.data_ptr(), u_ref.numel(), _C.stream_ptr()), "neosr_gelu")
    u = torch.empty_like(u_ref)
    t = ops.conv3x3(x, w, b, w_wino4=w4, act=_C.ACT_GELU, out2=u)
    assert torch.equal(u, u_ref) and torch.equal(t, t_ref)
    # backward-data of a convolution whose INPUT was gelu(z): gz = dgrad(gy) * gelu'(z)
    z = torch.randn(B, H, W, K, generator=g).to(DEV)
    gy = torch.randn(B, H, W, N, generator=g).to(DEV)
    w4d = ops.conv3x3_pack_wino4(w, ops.CONV_DGRAD)
    gt = ops.conv3x3(gy, w, None, mode=ops.CONV_DGRAD, w_wino4=w4d)
    gz_ref = torch.empty_like(gt)
    _C.check(lib.neosr_gelu(z.data_ptr(), gt.data_ptr(), gz_ref.data_ptr(), gt.numel(), _C.stream_ptr()), "neosr_gelu")
    gz = ops.conv3x3(gy, w, None, mode=ops.CONV_DGRAD, w_wino4=w4d, out_mask=z, out_mask_gelu=True)
    assert torch.equal(gz, gz_ref)
    # a launch that cannot take the F(4x4,3x3) kernel refuses the GELU epilogue instead of ignoring it
    with pytest.raises(_C.NeosrAmdError):
        ops.conv3x3(x, w, b, w_pack=ops.conv3x3_pack_weights(w), act=_C.ACT_GELU, out2=u)
