"""CPU oracle for the SwinIR generator (window-attention path) — TEST INFRASTRUCTURE ONLY.

PyTorch-CPU fp32 restatement of neosr/archs/swinir_arch.py: Mlp (:15-38), window_partition /
window_reverse (:41-78), WindowAttention incl. the relative-position index (:116-137, :150-212),
SwinTransformerBlock incl. calculate_mask (:313-392), RSTB (:657-663), Upsample /
UpsampleOneStep (:768-811) and swinir.forward (:1025-1079).  Functional: parameters come as a dict
keyed exactly like the reference state_dict.

Parity status: PINNED against tests/golden/swinir_*.npz (reference imported and run on CPU by
tests/golden/gen_golden_swinir.py; drop_path_rate=0 in the fixtures because DropPath draws from the
global torch RNG — the DropPath arithmetic itself is restated in `drop_path` and unit-tested).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

VARIANTS = {
    # neosr/archs/swinir_arch.py:1093-1130
    "swinir_small": dict(img_size=64, depths=[6] * 4, embed_dim=60, num_heads=[6] * 4,
                         upsampler="pixelshuffledirect", resi_connection="1conv"),
    "swinir_medium": dict(img_size=48, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6,
                          upsampler="pixelshuffle", resi_connection="1conv"),
    "swinir_large": dict(img_size=64, depths=[6] * 9, embed_dim=240, num_heads=[8] * 9,
                         upsampler="nearest+conv", resi_connection="3conv"),
}


def relative_position_index(ws: int) -> torch.Tensor:
    """swinir_arch.py:122-137: pairwise (dy + ws-1) * (2ws-1) + (dx + ws-1)."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x, ws):
    b, h, w, c = x.shape
    x = x.view(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def window_reverse(windows, ws, h, w):
    b = int(windows.shape[0] / (h * w / ws / ws))
    x = windows.view(b, h // ws, w // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def calculate_mask(h: int, w: int, ws: int, shift: int) -> torch.Tensor:
    """swinir_arch.py:313-341: 9 regions of the rolled image, -100 between different regions."""
    img = torch.zeros((1, h, w, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def window_attention(P, pre: str, x, heads: int, ws: int, mask=None):
    """WindowAttention.forward, non-flash branch (swinir_arch.py:150-212)."""
    b_, n, c = x.shape
    qkv = F.linear(x, P[f"{pre}.qkv.weight"], P.get(f"{pre}.qkv.bias"))
    qkv = qkv.reshape(b_, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (c // heads) ** -0.5
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws).view(-1)
    bias = P[f"{pre}.relative_position_bias_table"][idx].view(n, n, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, n, n)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(x, P[f"{pre}.proj.weight"], P[f"{pre}.proj.bias"])


def drop_path(x, keep_mask, keep_prob: float):
    """arch_util.py:118-133 with the Bernoulli draw `keep_mask` (B,) passed in."""
    if keep_mask is None:
        return x
    return x * (keep_mask / keep_prob).view(-1, *([1] * (x.ndim - 1)))


def swin_block(P, pre: str, x, x_size, heads: int, ws: int, shift: int, dp=None):
    """SwinTransformerBlock.forward (swinir_arch.py:343-392). dp = (mask_attn, mask_mlp, keep_prob)."""
    h, w = x_size
    b, _, c = x.shape
    shortcut = x
    x = F.layer_norm(x, (c,), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], 1e-5).view(b, h, w, c)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, c)
    mask = calculate_mask(h, w, ws, shift) if shift > 0 else None
    aw = window_attention(P, f"{pre}.attn", xw, heads, ws, mask).view(-1, ws, ws, c)
    x = window_reverse(aw, ws, h, w)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x.view(b, h * w, c)
    x = shortcut + (drop_path(x, dp[0], dp[2]) if dp else x)
    y = F.layer_norm(x, (c,), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], 1e-5)
    y = F.linear(F.gelu(F.linear(y, P[f"{pre}.mlp.fc1.weight"], P[f"{pre}.mlp.fc1.bias"])),
                 P[f"{pre}.mlp.fc2.weight"], P[f"{pre}.mlp.fc2.bias"])
    return x + (drop_path(y, dp[1], dp[2]) if dp else y)


def _conv(P, name, x, pad=1):
    return F.conv2d(x, P[f"{name}.weight"], P[f"{name}.bias"], padding=pad)


def _resi_conv(P, name, x, resi):
    if resi == "1conv":
        return _conv(P, name, x)
    x = F.leaky_relu(_conv(P, f"{name}.0", x), 0.2)
    x = F.leaky_relu(_conv(P, f"{name}.2", x, 0), 0.2)
    return _conv(P, f"{name}.4", x)


def swinir_forward(P, x, *, depths, num_heads, embed_dim, upsampler, resi_connection="1conv", window_size=8,
                   upscale=4, img_range=1.0, img_size=None, drop_path_masks=None, drop_path_rate=0.0):
    """swinir.forward (swinir_arch.py:1040-1079).  drop_path_masks: optional list (one (2,B) 0/1
    tensor per block) replaying the Bernoulli draws of DropPath in train mode."""
    c = embed_dim
    mean = 0.5 if x.shape[1] == 3 else 0.0
    x = (x - mean) * img_range
    x = _conv(P, "conv_first", x)
    b, _, h, w = x.shape
    t = x.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (c,), P["patch_embed.norm.weight"], P["patch_embed.norm.bias"], 1e-5)
    nblk = sum(depths)
    dpr = [v.item() for v in torch.linspace(0, drop_path_rate, nblk)]
    bi = 0
    for li, depth in enumerate(depths):
        t_in = t
        for j in range(depth):
            dp = None
            if drop_path_masks is not None and dpr[bi] > 0.0:
                dp = (drop_path_masks[bi][0], drop_path_masks[bi][1], 1.0 - dpr[bi])
            shift = 0 if j % 2 == 0 else window_size // 2
            t = swin_block(P, f"layers.{li}.residual_group.blocks.{j}", t, (h, w), num_heads[li], window_size,
                           shift, dp)
            bi += 1
        img = t.transpose(1, 2).view(b, c, h, w)
        img = _resi_conv(P, f"layers.{li}.conv", img, resi_connection)
        t = img.flatten(2).transpose(1, 2) + t_in
    t = F.layer_norm(t, (c,), P["norm.weight"], P["norm.bias"], 1e-5)
    feat = t.transpose(1, 2).view(b, c, h, w)
    x = _resi_conv(P, "conv_after_body", feat, resi_connection) + x
    if upsampler == "pixelshuffle":
        x = F.leaky_relu(_conv(P, "conv_before_upsample.0", x), 0.01)
        n_up = {2: 1, 4: 2, 8: 3}.get(upscale)
        if n_up is not None:
            for k in range(n_up):
                x = F.pixel_shuffle(_conv(P, f"upsample.{2 * k}", x), 2)
        else:
            x = F.pixel_shuffle(_conv(P, "upsample.0", x), 3)
        x = _conv(P, "conv_last", x)
    elif upsampler == "pixelshuffledirect":
        x = F.pixel_shuffle(_conv(P, "upsample.0", x), upscale)
    elif upsampler == "nearest+conv":
        x = F.leaky_relu(_conv(P, "conv_before_upsample.0", x), 0.01)
        x = F.leaky_relu(_conv(P, "conv_up1", F.interpolate(x, scale_factor=2, mode="nearest")), 0.2)
        x = F.leaky_relu(_conv(P, "conv_up2", F.interpolate(x, scale_factor=2, mode="nearest")), 0.2)
        x = _conv(P, "conv_last", F.leaky_relu(_conv(P, "conv_hr", x), 0.2))
    else:
        raise ValueError(upsampler)
    return x / img_range + mean
