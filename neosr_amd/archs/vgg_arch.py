"""``VGGFeatureExtractor`` — VGG19 taps for the perceptual loss (drop-in for
neosr/archs/vgg_arch.py:76-199).

Same layer naming (`vgg_net.conv1_1 … conv5_4`), taps returned BEFORE the ReLU that follows them,
input normalisation with mean 0.5 / std 0.25 (NOT ImageNet statistics — vgg_arch.py:159-173), frozen
weights.  `forward` runs conv3x3+ReLU on the MFMA kernel, `neosr_maxpool2`, and applies the
normalisation inside the NCHW->NHWC layout kernel.

Weights: torchvision's ImageNet VGG19 file cannot be downloaded here.  If
`experiments/pretrained_models/vgg19-dcbb9e9d.pth` (the reference's VGG_PRETRAIN_PATH) or
`$NEOSR_VGG19_WEIGHTS` exists it is loaded by key (`features.N.weight|bias`); otherwise the
extractor is randomly initialised and says so loudly — parity of this path is pinned on structure
with seeded random weights only.
"""

from __future__ import annotations

import os
from collections import OrderedDict
from pathlib import Path

import torch
from torch import nn

from neosr_amd.hip import layers as L
from neosr_amd.utils.misc import get_root_logger
from neosr_amd.utils.registry import ARCH_REGISTRY

VGG_PRETRAIN_PATH = "experiments/pretrained_models/vgg19-dcbb9e9d.pth"
_CFG_E = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def _names() -> list[str]:
    names, blk, j = [], 1, 1
    for v in _CFG_E:
        if v == "M":
            names.append(f"pool{blk}")
            blk, j = blk + 1, 1
        else:
            names += [f"conv{blk}_{j}", f"relu{blk}_{j}"]
            j += 1
    return names


NAMES = {"vgg19": _names()}


@ARCH_REGISTRY.register()
class VGGFeatureExtractor(nn.Module):
    def __init__(self, layer_name_list: list[str], vgg_type: str = "vgg19", use_input_norm: bool = True,
                 range_norm: bool = False, requires_grad: bool = False, remove_pooling: bool = False,
                 pooling_stride: int = 2) -> None:
        super().__init__()
        if vgg_type != "vgg19" or remove_pooling or pooling_stride != 2 or requires_grad:
            raise NotImplementedError("only the frozen vgg19 / 2x2-pool configuration has HIP kernels")
        self.layer_name_list = list(layer_name_list)
        self.use_input_norm, self.range_norm = use_input_norm, range_norm
        self.names = NAMES[vgg_type]
        max_idx = max(self.names.index(v) for v in self.layer_name_list)
        net: OrderedDict[str, nn.Module] = OrderedDict()
        c = 3
        feat_index = {}
        for i, (name, v) in enumerate(zip(self.names, [x for cfg in _CFG_E for x in ((cfg,) if cfg == "M" else (cfg, "R"))])):
            if i > max_idx:
                break
            if v == "M":
                net[name] = nn.MaxPool2d(kernel_size=2, stride=2)
            elif v == "R":
                net[name] = nn.ReLU(inplace=True)
            else:
                net[name] = nn.Conv2d(c, v, kernel_size=3, padding=1)
                nn.init.kaiming_normal_(net[name].weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(net[name].bias, 0)
                feat_index[name] = i  # torchvision `features.<i>`
                c = v
        self.vgg_net = nn.Sequential(net)
        self._load_pretrained(feat_index)
        self.vgg_net.eval()
        for p in self.parameters():
            p.requires_grad = False
            p._neosr_frozen = True  # packed conv images survive the optimizer steps of the other networks
        if self.use_input_norm:
            self.register_buffer("mean", torch.tensor([0.5, 0.5, 0.5]).view(1, 3, 1, 1))
            self.register_buffer("std", torch.tensor([0.25, 0.25, 0.25]).view(1, 3, 1, 1))

    def _load_pretrained(self, feat_index: dict[str, int]) -> None:
        logger = get_root_logger()
        for cand in (os.environ.get("NEOSR_VGG19_WEIGHTS"), VGG_PRETRAIN_PATH):
            if cand and Path(cand).exists():
                sd = torch.load(cand, map_location="cpu", weights_only=True)
                for name, i in feat_index.items():
                    getattr(self.vgg_net, name).weight.data.copy_(sd[f"features.{i}.weight"])
                    getattr(self.vgg_net, name).bias.data.copy_(sd[f"features.{i}.bias"])
                logger.info(f"VGG19 weights loaded from {cand}")
                return
        logger.warning("VGG19 ImageNet weights not found (no network access): perceptual features use "
                       "RANDOM weights. Put vgg19-dcbb9e9d.pth at %s or set NEOSR_VGG19_WEIGHTS.", VGG_PRETRAIN_PATH)

    def features_nhwc(self, x: torch.Tensor) -> dict[str, torch.Tensor]:
        """taps as channels-last (B,H,W,C) tensors (what the HIP loss kernels consume)"""
        if self.range_norm:
            raise NotImplementedError("range_norm")
        mean, std = (0.5, 0.25) if self.use_input_norm else (0.0, 1.0)
        t = L.VGGInput.apply(x, mean, std, 4)
        out = {}
        mods = self.vgg_net._modules
        keys = list(mods.keys())
        i = 0
        while i < len(keys):
            key, layer = keys[i], mods[keys[i]]
            if isinstance(layer, nn.Conv2d):
                tapped = key in self.layer_name_list
                has_relu = i + 1 < len(keys) and isinstance(mods[keys[i + 1]], nn.ReLU)
                if tapped or not has_relu:
                    t = L.conv3x3(t, layer.weight, layer.bias)          # pre-activation tap
                    if tapped:
                        out[key] = t
                    if has_relu:
                        t = L.LeakyReLU.apply(t, 0.0)
                else:
                    t = L.conv3x3(t, layer.weight, layer.bias, L.ACT_RELU, 0.0)  # fused bias+ReLU
                i += 2 if has_relu else 1
            elif isinstance(layer, nn.MaxPool2d):
                t = L.MaxPool2.apply(t)
                i += 1
            else:
                i += 1
        return out

    def forward(self, x: torch.Tensor) -> dict[str, torch.Tensor]:
        """dict of (n, c, h, w)-shaped taps (channels-last memory; logical NCHW like the reference)"""
        return {k: v.permute(0, 3, 1, 2) for k, v in self.features_nhwc(x).items()}
