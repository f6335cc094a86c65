"""Partitioned inference of `image.test()` (neosr/models/image.py:684-783) as a pure function of the
network callable, so the band arithmetic is testable on CPU against the reference's outputs."""

from __future__ import annotations

import torch

SHAVE = 16  # overlap (LR pixels) added on the inner sides of every band


def tiled_inference(fn, lq: torch.Tensor, tile: int, scale: int, out_channels: int | None = None) -> torch.Tensor:
    """(h // tile + 1) x (w // tile + 1) bands; the image is mirror-padded up to a multiple of the band
    count, each band is run with SHAVE extra pixels towards its neighbours, the centre parts are merged
    and the padding is cropped."""
    B, C, h, w = lq.shape
    C = C if out_channels is None else out_channels
    nh, nw = h // tile + 1, w // tile + 1
    pad_h, pad_w = (-h) % nh, (-w) % nw
    img = torch.cat([lq, torch.flip(lq, [2])], 2)[:, :, : h + pad_h, :]
    img = torch.cat([img, torch.flip(img, [3])], 3)[:, :, :, : w + pad_w]
    H, W = h + pad_h, w + pad_w
    sh, sw = H // nh, W // nw

    def band(i, n, step):  # (source lo, source hi, offset of the kept part inside the band)
        return (i * step - (SHAVE if i > 0 else 0), (i + 1) * step + (SHAVE if i < n - 1 else 0),
                SHAVE if i > 0 else 0)

    out = torch.zeros(B, C, H * scale, W * scale, device=lq.device)
    for i in range(nh):
        t0, t1, to = band(i, nh, sh)
        for j in range(nw):
            l0, l1, lo = band(j, nw, sw)
            y = fn(img[..., t0:t1, l0:l1].contiguous())
            out[..., i * sh * scale: (i + 1) * sh * scale, j * sw * scale: (j + 1) * sw * scale] = \
                y[..., to * scale: (to + sh) * scale, lo * scale: (lo + sw) * scale]
    return out[:, :, : h * scale, : w * scale]
