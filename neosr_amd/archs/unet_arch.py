"""``unet`` — U-Net discriminator with spectral norm (drop-in for neosr/archs/unet_arch.py:9-67).

Same constructor, same state-dict (conv0/conv9 `weight|bias`; conv1-8 `weight_orig`, `weight_u`,
`weight_v` as registered by `torch.nn.utils.spectral_norm`, which we call on the parameter
holders so seeded initialisation matches the reference draw for draw).  `forward` composes HIP
kernels: spectral-norm power iteration (`neosr_spectral_norm_fwd`, u/v advance on every train-mode
forward exactly like the hook), 4x4/s2 convs as space-to-depth + the MFMA 3x3 kernel, bilinear x2
(`neosr_bilinear_up2`), skip adds, fused bias + LeakyReLU(0.2).
"""

from __future__ import annotations

import torch
from torch import nn
from torch.nn.utils import spectral_norm

from neosr_amd.hip import layers as L
from neosr_amd.utils.registry import ARCH_REGISTRY


_FUSED_SKIP = __import__("os").environ.get("NEOSR_AMD_UNET_FUSED_SKIP", "1") != "0"   # A/B switch (bit-identical results)


@ARCH_REGISTRY.register()
class unet(nn.Module):
    def __init__(self, num_in_ch: int = 3, num_feat: int = 64, skip_connection: bool = True) -> None:
        super().__init__()
        self.skip_connection = skip_connection
        norm = spectral_norm
        self.conv0 = nn.Conv2d(num_in_ch, num_feat, kernel_size=3, stride=1, padding=1)
        self.conv1 = norm(nn.Conv2d(num_feat, num_feat * 2, 4, 2, 1, bias=False))
        self.conv2 = norm(nn.Conv2d(num_feat * 2, num_feat * 4, 4, 2, 1, bias=False))
        self.conv3 = norm(nn.Conv2d(num_feat * 4, num_feat * 8, 4, 2, 1, bias=False))
        self.conv4 = norm(nn.Conv2d(num_feat * 8, num_feat * 4, 3, 1, 1, bias=False))
        self.conv5 = norm(nn.Conv2d(num_feat * 4, num_feat * 2, 3, 1, 1, bias=False))
        self.conv6 = norm(nn.Conv2d(num_feat * 2, num_feat, 3, 1, 1, bias=False))
        self.conv7 = norm(nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=False))
        self.conv8 = norm(nn.Conv2d(num_feat, num_feat, 3, 1, 1, bias=False))
        self.conv9 = nn.Conv2d(num_feat, 1, 3, 1, 1)
        self.num_in_ch = num_in_ch

    def _w(self, conv: nn.Conv2d) -> torch.Tensor:
        """spectrally normalised weight; advances weight_u / weight_v in train mode"""
        return L.SpectralNorm.apply(conv.weight_orig, conv.weight_u, conv.weight_v, self.training, 1e-12)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lr = L.ACT_LRELU
        t = L.ToNHWC.apply(x, (self.num_in_ch + 3) // 4 * 4)
        x0 = L.conv3x3(t, self.conv0.weight, self.conv0.bias, lr, 0.2)
        if self.skip_connection and _FUSED_SKIP:
            # x0 / x1 / x2 feed a strided convolution and a skip addition, nothing else: the depth-to-space of the
            # convolution's input gradient, the skip gradient and the LeakyReLU derivative are ONE backward pass
            x1, x0 = L.conv4x4s2_skip(x0, 0.2, self._w(self.conv1), None, lr, 0.2)
            x2, x1 = L.conv4x4s2_skip(x1, 0.2, self._w(self.conv2), None, lr, 0.2)
            x3, x2 = L.conv4x4s2_skip(x2, 0.2, self._w(self.conv3), None, lr, 0.2)
        else:
            x1 = L.conv4x4s2(x0, self._w(self.conv1), None, lr, 0.2)
            x2 = L.conv4x4s2(x1, self._w(self.conv2), None, lr, 0.2)
            x3 = L.conv4x4s2(x2, self._w(self.conv3), None, lr, 0.2)
        x3 = L.BilinearUp2.apply(x3)
        x4 = L.conv3x3(x3, self._w(self.conv4), None, lr, 0.2)
        if self.skip_connection:
            x4 = L.Add.apply(x4, x2)
        x4 = L.BilinearUp2.apply(x4)
        x5 = L.conv3x3(x4, self._w(self.conv5), None, lr, 0.2)
        if self.skip_connection:
            x5 = L.Add.apply(x5, x1)
        x5 = L.BilinearUp2.apply(x5)
        x6 = L.conv3x3(x5, self._w(self.conv6), None, lr, 0.2)
        if self.skip_connection:
            x6 = L.Add.apply(x6, x0)
        # conv7 -> conv8 -> conv9: each output feeds exactly one conv, so the LeakyReLU derivatives are applied in the
        # consumers' backward-data epilogues instead of by an elementwise pass over two 256^2 x 64-channel gradients
        out = L.conv3x3(x6, self._w(self.conv7), None, lr, 0.2, sole_consumer_is_conv=True)
        out = L.conv3x3(out, self._w(self.conv8), None, lr, 0.2, sole_consumer_is_conv=True)
        out = L.conv3x3(out, self.conv9.weight, self.conv9.bias)
        return L.ToNCHW.apply(out, 1)
