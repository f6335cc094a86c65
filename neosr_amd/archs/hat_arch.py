"""``hat_s`` / ``hat_m`` / ``hat_l`` — HAT generators (drop-in for neosr/archs/hat_arch.py:831-1207).

Same constructor arguments, same ``state_dict`` (keys, shapes, the two ``relative_position_index_*``
buffers, ordering) and the same initialisation draw order as the reference.  The torch modules are
parameter holders; ``forward`` composes HIP kernels on channels-last HBM buffers (tokens ARE pixels):

  * HAB   LayerNorm -> { qkv GEMM -> streaming 16x16 (shifted-)window attention -> proj GEMM with the
          DropPath scale and the shortcut fused ;  conv3x3 -> GELU -> conv3x3 -> squeeze-excite gate fused
          with `+ conv_x * conv_scale` } -> LayerNorm -> Mlp (GELU, DropPath, residual fused)
  * OCAB  LayerNorm -> qkv GEMM -> overlapping cross-attention (16x16 queries, zero-padded 24x24 keys:
          nn.Unfold + einops.rearrange become addressing) -> proj + shortcut -> LayerNorm -> Mlp
  * RHAG  6 HAB + OCAB + conv3x3 with the residual in its epilogue
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


def _rpi_sa(ws: int) -> torch.Tensor:
    """[i, j] -> (yi - yj + ws-1)(2ws-1) + (xi - xj + ws-1)"""
    y, x = torch.arange(ws * ws) // ws, torch.arange(ws * ws) % ws
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def _rpi_oca(ws: int, overlap_ratio: float) -> torch.Tensor:
    """[query i of the ws window, key j of the wse window] -> (yj - yi + ws-wse+1)(ws+wse-1) + (xj - xi + ws-wse+1).
    The offset is the reference's (hat_arch.py:1058-1063): values run negative and index the bias table
    from its end; the attention kernel applies the same wrap."""
    wse = ws + int(overlap_ratio * ws)
    yi, xi = torch.arange(ws * ws) // ws, torch.arange(ws * ws) % ws
    yj, xj = torch.arange(wse * wse) // wse, torch.arange(wse * wse) % wse
    off = ws - wse + 1
    return (yj[None, :] - yi[:, None] + off) * (ws + wse - 1) + (xj[None, :] - xi[:, None] + off)


class ChannelAttention(nn.Module):
    def __init__(self, num_feat: int, squeeze_factor: int = 16) -> None:
        super().__init__()
        self.attention = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(num_feat, num_feat // squeeze_factor, 1, padding=0),
                                       nn.ReLU(inplace=True), nn.Conv2d(num_feat // squeeze_factor, num_feat, 1, padding=0),
                                       nn.Sigmoid())


class CAB(nn.Module):
    def __init__(self, num_feat: int, compress_ratio: int = 3, squeeze_factor: int = 30) -> None:
        super().__init__()
        self.cab = nn.Sequential(nn.Conv2d(num_feat, num_feat // compress_ratio, 3, 1, 1), nn.GELU(),
                                 nn.Conv2d(num_feat // compress_ratio, num_feat, 3, 1, 1),
                                 ChannelAttention(num_feat, squeeze_factor))

    def forward(self, y: torch.Tensor, res: torch.Tensor | None, alpha: float) -> torch.Tensor:
        """res + alpha * CAB(y) on channels-last (B,H,W,C) tensors"""
        c0, c2, ca = self.cab[0], self.cab[2], self.cab[3].attention
        t = T.Gelu.apply(L.conv3x3(y, c0.weight, c0.bias))
        t = L.conv3x3(t, c2.weight, c2.bias)
        return T.ChannelGate.apply(t, ca[1].weight, ca[1].bias, ca[3].weight, ca[3].bias, res, alpha)


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int) -> None:
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, window_size: int, num_heads: int, qkv_bias: bool = True, qk_scale=None) -> None:
        super().__init__()
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=0.02)


class HAB(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, compress_ratio=3,
                 squeeze_factor=30, conv_scale=0.01, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_path=0.0) -> None:
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        if min(input_resolution) <= window_size:
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias, qk_scale)
        self.conv_scale = conv_scale
        self.conv_block = CAB(dim, compress_ratio, squeeze_factor)
        self.drop_prob = float(drop_path)
        if self.drop_prob > 0.0:
            droppath_ctor_reseed()  # where the reference constructs DropPath(drop_path)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _plan_images(self, b: int, h: int, w: int) -> dict:
        """packed weight images of the two CAB convolutions, forward and backward-data, as `L.Conv3x3` would pick them"""
        c0, c2 = self.conv_block.cab[0].weight, self.conv_block.cab[2].weight
        # (valid while no weight changed: the fused optimizers bump `_C.WEIGHTS_EPOCH`, torch in-place writes `_version`;
        # a stale key goes through `L.packed_weights`, which re-packs every stale image of the device in one call)
        # The image TENSORS outlive an optimizer step (re-packed in place by the batched refresh, `L._repack_stale`): once any
        # convolution of this device has triggered that refresh for the current epoch (`L.images_fresh`), the cached dict
        # is valid again without eight per-image lookups per block and pass (~100 us of host time per HAB, hat_l: 7 ms / step)
        key = (c0._version, c2._version, c0.data_ptr(), b, h, w)
        hit = getattr(self, "_plan_imgs", None)
        # (not while a hipGraph capture is PENDING: `L._packed` must see the first request inside the capture — it re-packs
        # every image as a node of the graph, utils/graph.py — and clears the flag; a refresh made inside a capture does
        # not mark the images fresh, so the blocks of a captured pass keep asking)
        if hit is not None and hit[0] == key and not L.FORCE_REPACK_IN_CAPTURE and L.images_fresh(c0.device):
            return hit[1]
        out = {}
        for tag, conv in (("c0", self.conv_block.cab[0]), ("c2", self.conv_block.cab[2])):
            wt = conv.weight
            for m, mode, n_out in (("f", L.ops.CONV_FWD, wt.shape[0]), ("d", L.ops.CONV_DGRAD, wt.shape[1])):
                out[f"{tag}_pack_{m}"] = L.packed_weights(wt, mode)
                for k, v in L.wino_images(wt, mode, b, h, w, n_out).items():
                    out[f"{tag}_{k[2:]}_{m}"] = v
        self._plan_imgs = (key, out)
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # (B, H, W, C)
        b, h, w, _ = x.shape
        a = self.attn
        if T.use_block_plan(x) and a.qkv.bias is not None:   # the whole block as one library call per direction (csrc/blocks.hip)
            meta = getattr(self, "_plan_meta", None)
            cab, ca = self.conv_block.cab, self.conv_block.cab[3].attention
            if meta is None:
                meta = self._plan_meta = T.PlanMeta({
                    "names": _C.TBLOCK_PARAMS[:7] + _C.TBLOCK_CAB_PARAMS + _C.TBLOCK_PARAMS[7:],
                    "ints": {"heads": self.num_heads, "ws": self.window_size, "ks": self.window_size,
                             "shift": self.shift_size, "hidden": self.mlp.fc1.out_features, "attn": 1,
                             "cab_mid": cab[0].weight.shape[0], "cab_sq": ca[1].weight.shape[0]},
                    "floats": {"scale": float(a.scale), "eps1": self.norm1.eps, "eps2": self.norm2.eps,
                               "conv_scale": float(self.conv_scale)},
                    "images": self._plan_images})
            m = self.mlp
            params = (self.norm1.weight, self.norm1.bias, a.relative_position_bias_table, a.qkv.weight, a.qkv.bias,
                      a.proj.weight, a.proj.bias, cab[0].weight, cab[0].bias, cab[2].weight, cab[2].bias,
                      ca[1].weight, ca[1].bias, ca[3].weight, ca[3].bias, self.norm2.weight, self.norm2.bias,
                      m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
            rs = drop_scale(self.drop_prob, self.training, b, x.device)    # two sites, two draws (hat_arch.py:343,349)
            rs2 = drop_scale(self.drop_prob, self.training, b, x.device)
            return T.tblock(x, rs, rs2, meta, params)
        x, y = T.residual_layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)  # (shortcut, norm)
        qkv = T.linear(y, a.qkv.weight, a.qkv.bias)
        at = T.flash_window_attention(qkv, a.relative_position_bias_table, self.num_heads, self.window_size,
                                      self.shift_size, a.scale, self.window_size)
        x = T.linear(at, a.proj.weight, a.proj.bias, x, drop_scale(self.drop_prob, self.training, b, x.device), h * w)
        x = self.conv_block(y, x, self.conv_scale)
        x, y = T.residual_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return T.mlp(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, x,
                     drop_scale(self.drop_prob, self.training, b, x.device), h * w)


class OCAB(nn.Module):
    def __init__(self, dim, input_resolution, window_size, overlap_ratio, num_heads, qkv_bias=True, qk_scale=None,
                 mlp_ratio=2) -> None:
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.overlap_win_size = int(window_size * overlap_ratio) + window_size
        self.norm1 = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((window_size + self.overlap_win_size - 1) ** 2, num_heads))
        trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if T.use_block_plan(x) and self.qkv.bias is not None:   # one library call per direction (csrc/blocks.hip)
            meta = getattr(self, "_plan_meta", None)
            if meta is None:
                meta = self._plan_meta = T.PlanMeta({
                    "names": ("rpb", "n1_w", "n1_b") + _C.TBLOCK_PARAMS[3:],   # named_parameters(): the table comes first
                    "ints": {"heads": self.num_heads, "ws": self.window_size, "ks": self.overlap_win_size, "shift": 0,
                             "hidden": self.mlp.fc1.out_features, "attn": 1},
                    "floats": {"scale": float(self.scale), "eps1": self.norm1.eps, "eps2": self.norm2.eps}})
            m = self.mlp
            params = (self.relative_position_bias_table, self.norm1.weight, self.norm1.bias, self.qkv.weight,
                      self.qkv.bias, self.proj.weight, self.proj.bias, self.norm2.weight, self.norm2.bias,
                      m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
            return T.tblock(x, None, None, meta, params)
        x, y = T.residual_layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)  # (shortcut, norm)
        qkv = T.linear(y, self.qkv.weight, self.qkv.bias)
        at = T.flash_window_attention(qkv, self.relative_position_bias_table, self.num_heads, self.overlap_win_size, 0,
                                      self.scale, self.window_size)
        x = T.linear(at, self.proj.weight, self.proj.bias, x)
        x, y = T.residual_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return T.mlp(y, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, x)


class AttenBlocks(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, compress_ratio, squeeze_factor,
                 conv_scale, overlap_ratio, mlp_ratio, qkv_bias, qk_scale, drop_path) -> None:
        super().__init__()
        self.blocks = nn.ModuleList([
            HAB(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, compress_ratio,
                squeeze_factor, conv_scale, mlp_ratio, qkv_bias, qk_scale,
                drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)
        ])
        self.overlap_attn = OCAB(dim, input_resolution, window_size, overlap_ratio, num_heads, qkv_bias, qk_scale, mlp_ratio)


class RHAG(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, compress_ratio, squeeze_factor, conv_scale,
                 overlap_ratio, mlp_ratio, qkv_bias, qk_scale, drop_path, resi_connection) -> None:
        super().__init__()
        self.residual_group = AttenBlocks(dim, input_resolution, depth, num_heads, window_size, compress_ratio,
                                          squeeze_factor, conv_scale, overlap_ratio, mlp_ratio, qkv_bias, qk_scale,
                                          drop_path)
        self.conv = nn.Conv2d(dim, dim, 3, 1, 1) if resi_connection == "1conv" else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = x
        for blk in self.residual_group.blocks:
            y = blk(y)
        y = self.residual_group.overlap_attn(y)
        if isinstance(self.conv, nn.Conv2d):
            return L.conv3x3(y, self.conv.weight, self.conv.bias, res=x)
        return L.Add.apply(y, x)


class _Norm(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class hat(nn.Module):
    def __init__(self, img_size=64, patch_size=1, in_chans=3, embed_dim=96, depths=(6, 6, 6, 6), num_heads=(6, 6, 6, 6),
                 window_size=7, compress_ratio=3, squeeze_factor=30, conv_scale=0.01, overlap_ratio=0.5, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True, upscale=None, img_range=1.0, upsampler="",
                 resi_connection="1conv", **kwargs) -> None:
        super().__init__()
        if ape or drop_rate or attn_drop_rate or patch_size != 1 or norm_layer is not nn.LayerNorm:
            raise _C.NeosrAmdError("hat: ape / dropout / patch_size != 1 are not implemented")
        if window_size not in (16, 8) or 2 * int(window_size * overlap_ratio) != window_size:
            raise _C.NeosrAmdError("hat: the attention kernels implement window_size 16 (hat_s / hat_m / hat_l) or 8, "
                                   "with overlap_ratio 0.5")
        if upsampler != "pixelshuffle":
            raise _C.NeosrAmdError("hat: only upsampler='pixelshuffle' does anything in the reference forward "
                                   "(hat_arch.py:1134-1147)")
        self.window_size, self.shift_size, self.overlap_ratio = window_size, window_size // 2, overlap_ratio
        num_in_ch = num_out_ch = in_chans
        num_feat = 64
        self.img_range, self.in_chans = img_range, in_chans
        self.mean = 0.5 if in_chans == 3 else 0.0
        self.upscale = net_opt()[0] if upscale is None else upscale
        self.upsampler = upsampler
        self.register_buffer("relative_position_index_SA", _rpi_sa(window_size))
        self.register_buffer("relative_position_index_OCA", _rpi_oca(window_size, overlap_ratio))
        res = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)

        self.conv_first = nn.Conv2d(num_in_ch, embed_dim, 3, 1, 1)
        self.patch_embed = _Norm(embed_dim) if patch_norm else nn.Module()
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i, depth in enumerate(depths):
            self.layers.append(RHAG(embed_dim, res, depth, num_heads[i], window_size, compress_ratio, squeeze_factor,
                                    conv_scale, overlap_ratio, mlp_ratio, qkv_bias, qk_scale,
                                    dpr[sum(depths[:i]): sum(depths[: i + 1])], resi_connection))
        self.norm = nn.LayerNorm(embed_dim)
        self.conv_after_body = nn.Conv2d(embed_dim, embed_dim, 3, 1, 1) if resi_connection == "1conv" else nn.Identity()
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
        if x.shape[2] % self.window_size or x.shape[3] % self.window_size:
            raise _C.NeosrAmdError(f"hat: input {tuple(x.shape[2:])} must be a multiple of window_size "
                                   f"{self.window_size}")
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
        if isinstance(self.conv_after_body, nn.Conv2d):
            y = L.conv3x3(tok, self.conv_after_body.weight, self.conv_after_body.bias, res=x0)
        else:
            y = L.Add.apply(tok, x0)
        c = self.conv_before_upsample[0]
        y = L.conv3x3(y, c.weight, c.bias, L.ACT_LRELU, 0.01)
        for m in self.upsample:
            y = L.conv3x3(y, m.weight, m.bias) if isinstance(m, nn.Conv2d) else T.PixelShuffleNHWC.apply(y, m.upscale_factor)
        y = L.conv3x3(y, self.conv_last.weight, self.conv_last.bias)
        out = L.ToNCHW.apply(y, self.in_chans)
        return T.Affine.apply(out, self.mean * self.img_range, 1.0 / self.img_range)


_COMMON = dict(in_chans=3, window_size=16, conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, mlp_ratio=2,
               upsampler="pixelshuffle", resi_connection="1conv")


@ARCH_REGISTRY.register()
def hat_s(**kwargs):
    return hat(compress_ratio=24, squeeze_factor=24, depths=[6] * 6, embed_dim=144, num_heads=[6] * 6, **_COMMON, **kwargs)


@ARCH_REGISTRY.register()
def hat_m(**kwargs):
    return hat(compress_ratio=3, squeeze_factor=30, depths=[6] * 6, embed_dim=180, num_heads=[6] * 6, **_COMMON, **kwargs)


@ARCH_REGISTRY.register()
def hat_l(**kwargs):
    return hat(compress_ratio=3, squeeze_factor=30, depths=[6] * 12, embed_dim=180, num_heads=[6] * 12, **_COMMON, **kwargs)
