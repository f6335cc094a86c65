"""CPU oracle for the HAT generator — TEST INFRASTRUCTURE ONLY.

PyTorch-CPU fp32 restatement of neosr/archs/hat_arch.py: ChannelAttention / CAB (:15-52),
WindowAttention (:168-216), HAB (:299-351), OCAB (:445-516), AttenBlocks / RHAG (:607-720), the
relative-position index builders incl. the negative-index wrap of calculate_rpi_oca (:1015-1068),
calculate_mask (:1070-1100) and hat.forward (:1109-1147).  Functional: parameters come as a dict keyed
exactly like the reference state_dict.

Parity status: PINNED against tests/golden/hat_*.npz (reference imported and run on CPU by
tests/golden/gen_golden_hat.py; drop_path_rate = 0 in the fixtures).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from oracle.swinir_oracle import calculate_mask, drop_path, window_partition, window_reverse

VARIANTS = {
    # neosr/archs/hat_arch.py:1150-1207
    "hat_s": dict(window_size=16, compress_ratio=24, squeeze_factor=24, conv_scale=0.01, overlap_ratio=0.5,
                  depths=[6] * 6, embed_dim=144, num_heads=[6] * 6, mlp_ratio=2),
    "hat_m": dict(window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5,
                  depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, mlp_ratio=2),
    "hat_l": dict(window_size=16, compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5,
                  depths=[6] * 12, embed_dim=180, num_heads=[6] * 12, mlp_ratio=2),
}


def rpi_sa(ws: int) -> torch.Tensor:
    """hat_arch.py:1015-1033."""
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def rpi_oca(ws: int, overlap_ratio: float) -> torch.Tensor:
    """hat_arch.py:1035-1068 (values may be negative: they index the table from its end)."""
    wse = ws + int(overlap_ratio * ws)
    co = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    ce = torch.stack(torch.meshgrid([torch.arange(wse), torch.arange(wse)], indexing="ij")).flatten(1)
    rel = (ce[:, None, :] - co[:, :, None]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - wse + 1
    rel[:, :, 1] += ws - wse + 1
    rel[:, :, 0] *= ws + wse - 1
    return rel.sum(-1)


def _lin(P, name, x):
    return F.linear(x, P[f"{name}.weight"], P.get(f"{name}.bias"))


def _ln(P, name, x):
    return F.layer_norm(x, (x.shape[-1],), P[f"{name}.weight"], P[f"{name}.bias"], 1e-5)


def cab(P, pre: str, x):
    """CAB.forward (hat_arch.py:40-52): conv - GELU - conv - channel attention; x is (B,C,H,W)."""
    y = F.conv2d(x, P[f"{pre}.cab.0.weight"], P[f"{pre}.cab.0.bias"], padding=1)
    y = F.conv2d(F.gelu(y), P[f"{pre}.cab.2.weight"], P[f"{pre}.cab.2.bias"], padding=1)
    a = F.adaptive_avg_pool2d(y, 1)
    a = F.relu(F.conv2d(a, P[f"{pre}.cab.3.attention.1.weight"], P[f"{pre}.cab.3.attention.1.bias"]))
    a = torch.sigmoid(F.conv2d(a, P[f"{pre}.cab.3.attention.3.weight"], P[f"{pre}.cab.3.attention.3.bias"]))
    return y * a


def window_attention(P, pre: str, x, rpi, heads: int, ws: int, mask=None):
    """WindowAttention.forward (hat_arch.py:168-216)."""
    b_, n, c = x.shape
    qkv = _lin(P, f"{pre}.qkv", x).reshape(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * (c // heads) ** -0.5) @ k.transpose(-2, -1)
    bias = P[f"{pre}.relative_position_bias_table"][rpi.view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1)
    attn = attn + bias.contiguous().unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    x = (attn.softmax(-1) @ v).transpose(1, 2).reshape(b_, n, c)
    return _lin(P, f"{pre}.proj", x)


def hab(P, pre: str, x, x_size, heads: int, ws: int, shift: int, conv_scale: float, dp=None):
    """HAB.forward (hat_arch.py:299-351). dp = (mask_attn, mask_mlp, keep_prob) replays DropPath draws."""
    h, w = x_size
    b, _, c = x.shape
    shortcut = x
    x = _ln(P, f"{pre}.norm1", x).view(b, h, w, c)
    conv_x = cab(P, f"{pre}.conv_block", x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous().view(b, h * w, c)
    sx = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2)) if shift > 0 else x
    mask = calculate_mask(h, w, ws, shift) if shift > 0 else None
    xw = window_partition(sx, ws).view(-1, ws * ws, c)
    aw = window_attention(P, f"{pre}.attn", xw, rpi_sa(ws), heads, ws, mask).view(-1, ws, ws, c)
    sx = window_reverse(aw, ws, h, w)
    ax = (torch.roll(sx, shifts=(shift, shift), dims=(1, 2)) if shift > 0 else sx).view(b, h * w, c)
    x = shortcut + (drop_path(ax, dp[0], dp[2]) if dp else ax) + conv_x * conv_scale
    y = _lin(P, f"{pre}.mlp.fc2", F.gelu(_lin(P, f"{pre}.mlp.fc1", _ln(P, f"{pre}.norm2", x))))
    return x + (drop_path(y, dp[1], dp[2]) if dp else y)


def ocab(P, pre: str, x, x_size, heads: int, ws: int, overlap_ratio: float):
    """OCAB.forward (hat_arch.py:445-516): queries from the window, keys / values from the zero-padded
    overlapping window gathered by nn.Unfold; bias rows picked with (possibly negative) rpi_oca."""
    h, w = x_size
    b, _, c = x.shape
    ows = int(ws * overlap_ratio) + ws
    shortcut = x
    x = _ln(P, f"{pre}.norm1", x).view(b, h, w, c)
    qkv = _lin(P, f"{pre}.qkv", x).reshape(b, h, w, 3, c).permute(3, 0, 4, 1, 2)
    q = qkv[0].permute(0, 2, 3, 1)
    kv = torch.cat((qkv[1], qkv[2]), dim=1)
    qw = window_partition(q, ws).view(-1, ws * ws, c)
    kvw = F.unfold(kv, kernel_size=(ows, ows), stride=ws, padding=(ows - ws) // 2)  # b, 2c*ows*ows, nw
    nw = kvw.shape[-1]
    kvw = kvw.view(b, 2, c, ows * ows, nw).permute(1, 0, 4, 3, 2).reshape(2, b * nw, ows * ows, c)
    kw, vw = kvw[0], kvw[1]
    d = c // heads
    qh = qw.reshape(-1, ws * ws, heads, d).permute(0, 2, 1, 3)
    kh = kw.reshape(-1, ows * ows, heads, d).permute(0, 2, 1, 3)
    vh = vw.reshape(-1, ows * ows, heads, d).permute(0, 2, 1, 3)
    attn = (qh * d**-0.5) @ kh.transpose(-2, -1)
    rpi = rpi_oca(ws, overlap_ratio)
    bias = P[f"{pre}.relative_position_bias_table"][rpi.view(-1)].view(ws * ws, ows * ows, -1).permute(2, 0, 1)
    attn = (attn + bias.contiguous().unsqueeze(0)).softmax(-1)
    aw = (attn @ vh).transpose(1, 2).reshape(-1, ws, ws, c)
    x = window_reverse(aw, ws, h, w).view(b, h * w, c)
    x = _lin(P, f"{pre}.proj", x) + shortcut
    return x + _lin(P, f"{pre}.mlp.fc2", F.gelu(_lin(P, f"{pre}.mlp.fc1", _ln(P, f"{pre}.norm2", x))))


def hat_forward(P, x, *, depths, num_heads, embed_dim, window_size=16, conv_scale=0.01, overlap_ratio=0.5,
                upscale=4, img_range=1.0, drop_path_masks=None, drop_path_rate=0.0, **_):
    """hat.forward, upsampler = pixelshuffle, resi_connection = 1conv (hat_arch.py:1109-1147)."""
    c, ws = embed_dim, window_size
    mean = 0.5 if x.shape[1] == 3 else 0.0
    x = (x - mean) * img_range
    x = F.conv2d(x, P["conv_first.weight"], P["conv_first.bias"], padding=1)
    b, _, h, w = x.shape
    t = _ln(P, "patch_embed.norm", x.flatten(2).transpose(1, 2))
    dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
    bi = 0
    for li, depth in enumerate(depths):
        t_in = t
        for j in range(depth):
            dp = None
            if drop_path_masks is not None and dpr[bi] > 0.0:
                dp = (drop_path_masks[bi][0], drop_path_masks[bi][1], 1.0 - dpr[bi])
            t = hab(P, f"layers.{li}.residual_group.blocks.{j}", t, (h, w), num_heads[li], ws,
                    0 if j % 2 == 0 else ws // 2, conv_scale, dp)
            bi += 1
        t = ocab(P, f"layers.{li}.residual_group.overlap_attn", t, (h, w), num_heads[li], ws, overlap_ratio)
        img = t.transpose(1, 2).view(b, c, h, w)
        img = F.conv2d(img, P[f"layers.{li}.conv.weight"], P[f"layers.{li}.conv.bias"], padding=1)
        t = img.flatten(2).transpose(1, 2) + t_in
    t = _ln(P, "norm", t)
    feat = t.transpose(1, 2).view(b, c, h, w)
    x = F.conv2d(feat, P["conv_after_body.weight"], P["conv_after_body.bias"], padding=1) + x
    x = F.leaky_relu(F.conv2d(x, P["conv_before_upsample.0.weight"], P["conv_before_upsample.0.bias"], padding=1), 0.01)
    n_up = {2: 1, 4: 2, 8: 3}.get(upscale)
    if n_up is not None:
        for k in range(n_up):
            x = F.pixel_shuffle(F.conv2d(x, P[f"upsample.{2 * k}.weight"], P[f"upsample.{2 * k}.bias"], padding=1), 2)
    else:
        x = F.pixel_shuffle(F.conv2d(x, P["upsample.0.weight"], P["upsample.0.bias"], padding=1), 3)
    x = F.conv2d(x, P["conv_last.weight"], P["conv_last.bias"], padding=1)
    return x / img_range + mean
