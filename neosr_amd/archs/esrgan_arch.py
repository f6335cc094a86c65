"""``esrgan`` — RRDBNet generator (drop-in for neosr/archs/esrgan_arch.py:145-214).

Same constructor, same ``state_dict`` keys / shapes / initialisation order (so a seeded reference
run and a seeded run here start from identical weights, and reference ``.pth`` files load), but
``forward`` does not execute torch convolutions: it hands the parameter pointers to the HIP plan
``neosr_rrdbnet_forward`` (fp32 MFMA implicit-GEMM convs, concat-free RDB buffers, fused
bias/LeakyReLU/residual epilogues, nearest-x2 folded into the conv loader).  The ``nn.Conv2d``
objects below are parameter holders only.
"""

from __future__ import annotations

import torch
from torch import nn

from neosr_amd.archs.arch_util import HipNet, default_init_weights, net_opt
from neosr_amd.hip.nets import RRDBNetFunction
from neosr_amd.utils.registry import ARCH_REGISTRY


def pixel_unshuffle(x: torch.Tensor, scale: int) -> torch.Tensor:
    """(b,c,hh,hw) -> (b,c*s*s,hh/s,hw/s); index order of neosr/archs/esrgan_arch.py:60-79."""
    b, c, hh, hw = x.size()
    assert hh % scale == 0 and hw % scale == 0
    h, w = hh // scale, hw // scale
    return x.view(b, c, h, scale, w, scale).permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale**2, h, w)


class ResidualDenseBlock(nn.Module):
    """Parameter holder: conv1..conv5 of an RDB (esrgan_arch.py:82-107)."""

    def __init__(self, num_feat: int = 64, num_grow_ch: int = 32) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(num_feat, num_grow_ch, 3, 1, 1)
        self.conv2 = nn.Conv2d(num_feat + num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv3 = nn.Conv2d(num_feat + 2 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv4 = nn.Conv2d(num_feat + 3 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv5 = nn.Conv2d(num_feat + 4 * num_grow_ch, num_feat, 3, 1, 1)
        default_init_weights([self.conv1, self.conv2, self.conv3, self.conv4, self.conv5], 0.1)


class RRDB(nn.Module):
    def __init__(self, num_feat: int, num_grow_ch: int = 32) -> None:
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)


@ARCH_REGISTRY.register()
class esrgan(HipNet):
    plan_sends_grad_buckets = True   # models/image.py: the C++ backward plan reduces arena suffixes itself (GradSync marks)
    def __init__(self, num_in_ch: int = 3, num_out_ch: int = 3, scale: int | None = None,
                 num_feat: int = 64, num_block: int = 23, num_grow_ch: int = 32) -> None:
        super().__init__()
        self.scale = net_opt()[0] if scale is None else scale
        if self.scale == 2:
            num_in_ch = num_in_ch * 4
        elif self.scale == 1:
            num_in_ch = num_in_ch * 16
        self.conv_first = nn.Conv2d(num_in_ch, num_feat, 3, 1, 1)
        self.body = nn.Sequential(*[RRDB(num_feat, num_grow_ch) for _ in range(num_block)])
        self.conv_body = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)
        self._hp = {"num_out_ch": num_out_ch, "num_feat": num_feat, "num_block": num_block,
                    "num_grow_ch": num_grow_ch}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.scale == 2:
            feat = pixel_unshuffle(x, scale=2)
        elif self.scale == 1:
            feat = pixel_unshuffle(x, scale=4)
        else:
            feat = x
        # `_neosr_grad_sync`: set by the model on data-parallel runs (overlapped all-reduce of the gradient arena)
        hp = dict(self._hp, training=self.training, sync=getattr(self, "_neosr_grad_sync", None))
        return RRDBNetFunction.apply(feat, hp, *self._plan_params())
