"""``compact`` — SRVGGNetCompact (drop-in for neosr/archs/compact_arch.py:11-85).

``body`` is the same ``nn.ModuleList`` of Conv2d / activation modules (identical state-dict keys
``body.{0,2,..}.weight|bias`` and PReLU ``body.{1,3,..}.weight``); they only hold parameters.
``forward`` runs the HIP plan ``neosr_compact_forward``: conv3x3 with the previous layer's PReLU
applied in the loader, then a bit-exact PixelShuffle fused with the nearest-upsampled residual.
"""

from __future__ import annotations

import torch
from torch import nn

from neosr_amd import _C
from neosr_amd.archs.arch_util import HipNet, net_opt
from neosr_amd.hip.nets import CompactFunction, direct_state
from neosr_amd.utils.registry import ARCH_REGISTRY

_ACT_IDS = {"prelu": _C.ACT_PRELU, "relu": _C.ACT_RELU, "leakyrelu": _C.ACT_LRELU}


def _make_act(act_type: str, num_feat: int) -> nn.Module:
    if act_type == "relu":
        return nn.ReLU(inplace=True)
    if act_type == "prelu":
        return nn.PReLU(num_parameters=num_feat)
    if act_type == "leakyrelu":
        return nn.LeakyReLU(negative_slope=0.1, inplace=True)
    msg = f"compact: unknown act_type {act_type!r}"
    raise ValueError(msg)


@ARCH_REGISTRY.register()
class compact(HipNet):
    def __init__(self, num_in_ch: int = 3, num_out_ch: int = 3, num_feat: int = 64,
                 num_conv: int = 16, upscale: int | None = None, act_type: str = "prelu",
                 **kwargs) -> None:  # noqa: ARG002  (unknown keys tolerated like the reference)
        super().__init__()
        self.num_in_ch, self.num_out_ch = num_in_ch, num_out_ch
        self.num_feat, self.num_conv = num_feat, num_conv
        self.upscale = net_opt()[0] if upscale is None else upscale
        self.act_type = act_type

        self.body = nn.ModuleList()
        self.body.append(nn.Conv2d(num_in_ch, num_feat, 3, 1, 1))
        self.body.append(_make_act(act_type, num_feat))
        for _ in range(num_conv):
            self.body.append(nn.Conv2d(num_feat, num_feat, 3, 1, 1))
            self.body.append(_make_act(act_type, num_feat))
        self.body.append(nn.Conv2d(num_feat, num_out_ch * self.upscale * self.upscale, 3, 1, 1))
        self.upsampler = nn.PixelShuffle(self.upscale)  # kept for module-tree parity; no params

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hp = {"num_out_ch": self.num_out_ch, "num_feat": self.num_feat, "num_conv": self.num_conv,
              "upscale": self.upscale, "act_type": _ACT_IDS[self.act_type],
              "training": self.training}
        params = self._plan_params()
        st = direct_state(self, params) if self.training else None
        if st is not None:  # (inside the model's `direct_param_grads()` scope: backward assigns `.grad` itself)
            hp["direct"] = st
            return CompactFunction.apply(x, hp, st.anchor)
        return CompactFunction.apply(x, hp, *params)
