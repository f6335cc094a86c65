"""``swinir_small`` / ``swinir_medium`` / ``swinir_large`` — SwinIR generators (drop-in for
neosr/archs/swinir_arch.py:819-1130).

Same constructor arguments, same ``state_dict`` (keys, shapes, buffer order incl.
``relative_position_index`` / ``attn_mask``) and the same initialisation draw order as the reference,
so seeded runs start from identical weights and reference ``.pth`` files load.  The torch modules
below are parameter holders; ``forward`` composes HIP kernels on channels-last HBM buffers:

  * tokens ARE channels-last pixels — PatchEmbed / PatchUnEmbed / flatten / transpose cost nothing
  * LayerNorm                         ``neosr_layernorm_fwd/bwd``
  * qkv / proj / fc1 / fc2            fp32 MFMA GEMM with fused bias, exact GELU, DropPath scale and
                                      the residual shortcut (``neosr_gemm``)
  * roll + window partition + SDPA    ``neosr_window_attention_fwd/bwd`` (addressing only; analytic
                                      relative-position index and shifted-window mask)
  * 3x3 convs / PixelShuffle          the MFMA implicit-GEMM conv and an index kernel
"""

from __future__ import annotations

import math

import torch
from torch import nn
from torch.nn.init import trunc_normal_

from neosr_amd import _C
from neosr_amd.archs.arch_util import DropPathBank, drop_path_bank, drop_scale, droppath_ctor_reseed, net_opt
from neosr_amd.hip import layers as L
from neosr_amd.hip import transformer as T
from neosr_amd.utils.registry import ARCH_REGISTRY


def _relative_position_index(ws: int) -> torch.Tensor:
    """[i, j] -> (yi - yj + ws-1) * (2ws-1) + (xi - xj + ws-1) for tokens i, j of a ws x ws window."""
    y, x = torch.arange(ws * ws) // ws, torch.arange(ws * ws) % ws
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def _shift_mask(h: int, w: int, ws: int, shift: int) -> torch.Tensor:
    """(nW, ws*ws, ws*ws) 0 / -100 mask of the shifted-window blocks: a token of the rolled image
    belongs to one of 3x3 regions (before the last window / last window before the seam / after the
    seam, per axis); pairs from different regions are masked.  The kernel evaluates the same rule."""

    def region(n: int) -> torch.Tensor:
        p = torch.arange(n)
        return (p >= n - ws).long() + (p >= n - shift).long()

    rid = region(h)[:, None] * 3 + region(w)[None, :]
    rid = rid.view(h // ws, ws, w // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    return torch.where(rid[:, :, None] != rid[:, None, :], -100.0, 0.0)


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: int, num_heads: int, qkv_bias: bool = True, qk_scale=None) -> None:
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=0.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=8, shift_size=0, mlp_ratio=2.0,
                 qkv_bias=True, qk_scale=None, drop_path=0.0) -> None:
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.drop_prob = window_size, shift_size, float(drop_path)
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias, qk_scale)
        if self.drop_prob > 0.0:
            droppath_ctor_reseed()  # where the reference constructs DropPath(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        mask = _shift_mask(*input_resolution, self.window_size, self.shift_size) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", mask)

    def _plan_params(self):
        a, m = self.attn, self.mlp
        return (self.norm1.weight, self.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias,
                a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                m.fc2.weight, m.fc2.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # x: (B, H, W, C) channels-last tokens
        b, h, w, _ = x.shape
        a = self.attn
        if T.use_block_plan(x) and a.qkv.bias is not None:   # the whole block as one library call per direction (csrc/blocks.hip)
            meta = getattr(self, "_plan_meta", None)
            if meta is None:
                meta = self._plan_meta = T.PlanMeta({
                    "names": _C.TBLOCK_PARAMS,
                    "ints": {"heads": self.num_heads, "ws": self.window_size, "ks": self.window_size,
                             "shift": self.shift_size, "hidden": self.mlp.fc1.out_features, "attn": 0},
                    "floats": {"scale": float(a.scale), "eps1": self.norm1.eps, "eps2": self.norm2.eps}})
            rs = drop_scale(self.drop_prob, self.training, b, x.device)    # two sites, two draws (swinir_arch.py:387,390)
            rs2 = drop_scale(self.drop_prob, self.training, b, x.device)
            return T.tblock(x, rs, rs2, meta, self._plan_params())
        x, y = T.residual_layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)  # (shortcut, norm)
        qkv = T.linear(y, a.qkv.weight, a.qkv.bias)
        y = T.window_attention(qkv, a.relative_position_bias_table, self.num_heads, self.window_size,
                               self.shift_size, a.scale)
        x = T.linear(y, a.proj.weight, a.proj.bias, x, drop_scale(self.drop_prob, self.training, b, x.device), h * w)
        x, y = T.residual_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return T.mlp(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, x,
                     drop_scale(self.drop_prob, self.training, b, x.device), h * w)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale,
                 drop_path) -> None:
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size,
                                 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias, qk_scale,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)
        ])


def _resi_conv_holder(dim: int, resi_connection: str) -> nn.Module:
    if resi_connection == "1conv":
        return nn.Conv2d(dim, dim, 3, 1, 1)
    if resi_connection == "3conv":
        return nn.Sequential(nn.Conv2d(dim, dim // 4, 3, 1, 1), nn.LeakyReLU(0.2, True),
                             nn.Conv2d(dim // 4, dim // 4, 1, 1, 0), nn.LeakyReLU(0.2, True),
                             nn.Conv2d(dim // 4, dim, 3, 1, 1))
    raise ValueError(f"resi_connection {resi_connection} is not supported")


def _resi_conv(m: nn.Module, x: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """`1conv`: one 3x3; `3conv`: 3x3 -> LeakyReLU(0.2) -> 1x1 (a GEMM on channels-last) -> LeakyReLU -> 3x3;
    the `+ res` shortcut rides in the last conv's epilogue."""
    if isinstance(m, nn.Conv2d):
        return L.conv3x3(x, m.weight, m.bias, res=res)
    x = L.conv3x3(x, m[0].weight, m[0].bias, L.ACT_LRELU, 0.2)
    x = T.linear(x, m[2].weight.view(m[2].weight.shape[0], -1), m[2].bias)
    x = L.LeakyReLU.apply(x, 0.2)
    return L.conv3x3(x, m[4].weight, m[4].bias, res=res)


class RSTB(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale,
                 drop_path, resi_connection) -> None:
        super().__init__()
        self.residual_group = BasicLayer(dim, input_resolution, depth, num_heads, window_size, mlp_ratio,
                                         qkv_bias, qk_scale, drop_path)
        self.conv = _resi_conv_holder(dim, resi_connection)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = x
        for blk in self.residual_group.blocks:
            y = blk(y)
        return _resi_conv(self.conv, y, x)


class _Norm(nn.Module):
    """holder that reproduces the `patch_embed.norm.*` key prefix"""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class swinir(nn.Module):
    def __init__(self, img_size=32, patch_size=1, in_chans=3, embed_dim=60, depths=(6, 6, 6, 6),
                 num_heads=(6, 6, 6, 6), flash_attn=False, window_size=8, mlp_ratio=2.0, qkv_bias=True,
                 qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 ape=False, patch_norm=True, use_checkpoint=False, upscale=None, img_range=1.0,
                 upsampler="pixelshuffle", resi_connection="1conv", **kwargs) -> None:
        super().__init__()
        if flash_attn or ape or drop_rate or attn_drop_rate or patch_size != 1 or norm_layer is not nn.LayerNorm:
            # flash_attn runs SDPA under no_grad in the reference (no gradient reaches qkv); ape /
            # dropout / patch_size > 1 are never enabled by the registered variants
            raise _C.NeosrAmdError("swinir: flash_attn / ape / dropout / patch_size != 1 are not implemented")
        if upsampler not in ("pixelshuffle", "pixelshuffledirect", "nearest+conv"):
            raise _C.NeosrAmdError(f"swinir: upsampler {upsampler!r} is not implemented")
        num_in_ch = num_out_ch = in_chans
        num_feat = 64
        self.img_range = img_range
        self.mean = 0.5 if in_chans == 3 else 0.0
        self.upscale = net_opt()[0] if upscale is None else upscale
        self.upsampler, self.in_chans, self.embed_dim, self.window_size = upsampler, in_chans, embed_dim, window_size
        res = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.patches_resolution = res

        self.conv_first = nn.Conv2d(num_in_ch, embed_dim, 3, 1, 1)
        self.patch_embed = _Norm(embed_dim) if patch_norm else nn.Module()
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i, depth in enumerate(depths):
            self.layers.append(RSTB(embed_dim, res, depth, num_heads[i], window_size, mlp_ratio, qkv_bias,
                                    qk_scale, dpr[sum(depths[:i]): sum(depths[: i + 1])], resi_connection))
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = _resi_conv_holder(embed_dim, resi_connection)
        if upsampler == "pixelshuffle":
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            m: list[nn.Module] = []
            if (self.upscale & (self.upscale - 1)) == 0:
                for _ in range(int(math.log2(self.upscale))):
                    m += [nn.Conv2d(num_feat, 4 * num_feat, 3, 1, 1), nn.PixelShuffle(2)]
            elif self.upscale == 3:
                m += [nn.Conv2d(num_feat, 9 * num_feat, 3, 1, 1), nn.PixelShuffle(3)]
            else:
                raise ValueError(f"scale {self.upscale} is not supported. Supported scales: 2^n and 3.")
            self.upsample = nn.Sequential(*m)
            self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)
        elif upsampler == "pixelshuffledirect":
            self.upsample = nn.Sequential(nn.Conv2d(embed_dim, self.upscale**2 * num_out_ch, 3, 1, 1),
                                          nn.PixelShuffle(self.upscale))
        else:  # nearest+conv
            assert self.upscale == 4, "only support x4 now."
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(embed_dim, num_feat, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
            self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m: nn.Module) -> None:
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _C.require_device(x, "x")
        ws = self.window_size
        if x.shape[2] % ws or x.shape[3] % ws:
            raise _C.NeosrAmdError(f"swinir: input {tuple(x.shape[2:])} must be a multiple of window_size {ws}")
        lr = L.ACT_LRELU
        # (x - mean) * img_range fused with NCHW -> NHWC
        t = L.VGGInput.apply(x, self.mean, 1.0 / self.img_range, (self.in_chans + 3) // 4 * 4)
        x0 = L.conv3x3(t, self.conv_first.weight, self.conv_first.bias)
        tok = x0
        if hasattr(self.patch_embed, "norm"):
            n = self.patch_embed.norm
            tok = T.layer_norm(tok, n.weight, n.bias, n.eps)
        if not hasattr(self, "_dp_bank"):
            self._dp_bank = DropPathBank()
        with drop_path_bank(self._dp_bank, self.training, x.shape[0], x.device):
            for layer in self.layers:
                tok = layer(tok)
        tok = T.layer_norm(tok, self.norm.weight, self.norm.bias, self.norm.eps)
        y = _resi_conv(self.conv_after_body, tok, x0)
        if self.upsampler == "pixelshuffle":
            c = self.conv_before_upsample[0]
            y = L.conv3x3(y, c.weight, c.bias, lr, 0.01)
            for m in self.upsample:
                if isinstance(m, nn.Conv2d):
                    y = L.conv3x3(y, m.weight, m.bias)
                else:
                    y = T.PixelShuffleNHWC.apply(y, m.upscale_factor)
            y = L.conv3x3(y, self.conv_last.weight, self.conv_last.bias)
        elif self.upsampler == "pixelshuffledirect":
            c = self.upsample[0]
            y = T.PixelShuffleNHWC.apply(L.conv3x3(y, c.weight, c.bias), self.upscale)
        else:
            c = self.conv_before_upsample[0]
            y = L.conv3x3(y, c.weight, c.bias, lr, 0.01)
            y = L.conv3x3(y, self.conv_up1.weight, self.conv_up1.bias, lr, 0.2, ups=True)
            y = L.conv3x3(y, self.conv_up2.weight, self.conv_up2.bias, lr, 0.2, ups=True)
            y = L.conv3x3(y, self.conv_hr.weight, self.conv_hr.bias, lr, 0.2)
            y = L.conv3x3(y, self.conv_last.weight, self.conv_last.bias)
        out = L.ToNCHW.apply(y, self.in_chans)
        # x / img_range + mean
        return T.Affine.apply(out, self.mean * self.img_range, 1.0 / self.img_range)


@ARCH_REGISTRY.register()
def swinir_small(**kwargs):
    return swinir(img_size=64, depths=[6, 6, 6, 6], embed_dim=60, num_heads=[6, 6, 6, 6],
                  upsampler="pixelshuffledirect", resi_connection="1conv", **kwargs)


@ARCH_REGISTRY.register()
def swinir_medium(**kwargs):
    return swinir(img_size=48, depths=[6, 6, 6, 6, 6, 6], embed_dim=180, num_heads=[6, 6, 6, 6, 6, 6],
                  upsampler="pixelshuffle", resi_connection="1conv", **kwargs)


@ARCH_REGISTRY.register()
def swinir_large(**kwargs):
    return swinir(img_size=64, embed_dim=240, depths=[6] * 9, num_heads=[8] * 9, upsampler="nearest+conv",
                  resi_connection="3conv", **kwargs)
