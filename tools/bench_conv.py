#!/usr/bin/env python
"""Per-shape micro-benchmark of the conv kernels (GPU box only): TFLOP/s of fwd / dgrad / wgrad for
the layer shapes of BASELINE configs[1] (esrgan, B=16, 64x64 LR).  torch.cuda events on the current
stream (the stream the C ABI launches on)."""
from __future__ import annotations

import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neosr_amd.hip import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    shapes = [  # name, H, W, K, N, buffer channels
        ("rdb.conv1 64->32", 64, 64, 64, 32, 192),
        ("rdb.conv2 96->32", 64, 64, 96, 32, 192),
        ("rdb.conv3 128->32", 64, 64, 128, 32, 192),
        ("rdb.conv4 160->32", 64, 64, 160, 32, 192),
        ("rdb.conv5 192->64", 64, 64, 192, 64, 192),
        ("conv_body 64->64", 64, 64, 64, 64, 64),
        ("conv_hr 64->64 @256", 256, 256, 64, 64, 64),
    ]
    print(f"B={B}")
    print(f"{'layer':24s} {'fwd TF':>8s} {'dgrad TF':>9s} {'wgrad TF':>9s}   (us: fwd/dgrad/wgrad)")
    for name, H, W, K, N, CC in shapes:
        x = torch.randn(B, H, W, CC, device=DEV)
        w = torch.randn(N, K, 3, 3, device=DEV) * 0.05
        bias = torch.randn(N, device=DEV)
        out = torch.empty(B, H, W, N, device=DEV)
        gy = torch.randn(B, H, W, N, device=DEV)
        gx = torch.zeros(B, H, W, K, device=DEV)
        flops = 2.0 * B * H * W * K * N * 9
        tf = timeit(lambda: ops.conv3x3(x, w, bias, out=out, k_in=K, act=ops.ACT_LRELU, slope=0.2))
        td = timeit(lambda: ops.conv3x3(gy, w, None, mode=ops.CONV_DGRAD, out=gx, in_mask=out, mask_slope=0.2))
        dw = torch.empty_like(w)
        db = torch.empty(N, device=DEV)
        tw = timeit(lambda: ops.conv3x3_wgrad(x, gy, N, K, dw=dw, db=db))
        print(f"{name:24s} {flops / tf / 1e12:8.1f} {flops / td / 1e12:9.1f} {flops / tw / 1e12:9.1f}"
              f"   ({tf * 1e6:.0f}/{td * 1e6:.0f}/{tw * 1e6:.0f})")


if __name__ == "__main__":
    main()
