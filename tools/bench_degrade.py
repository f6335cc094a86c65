#!/usr/bin/env python
"""Micro-benchmark of the degradation-bank kernels at BASELINE config-3 sizes (B=32, 512^2 GT) and of
one full `otf.feed_data`.  Reports time and algorithmic HBM GB/s (1 read + 1 write of the tensor unless
noted) against the 8 TB/s peak.  GPU box only."""
from __future__ import annotations

import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from neosr_amd.data.degradations import KernelSampler  # noqa: E402
from neosr_amd.hip import degrade as D  # noqa: E402

DEV = "cuda"

# options/train_esrgan_otf.toml [degradations] of the reference (values only)
_K = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
_P = [0.45, 0.25, 0.12, 0.03, 0.12, 0.03]
DEG_TABLE = {"resize_prob": [0.3, 0.4, 0.3], "resize_range": [0.5, 1.5], "gaussian_noise_prob": 0.2,
             "noise_range": [0, 2], "poisson_scale_range": [0.05, 0.25], "gray_noise_prob": 0.1,
             "jpeg_range": [40, 95], "second_blur_prob": 0.4, "resize_prob2": [0.3, 0.4, 0.3],
             "resize_range2": [0.3, 1.5], "gaussian_noise_prob2": 0.2, "noise_range2": [0, 2],
             "poisson_scale_range2": [0.05, 0.1], "gray_noise_prob2": 0.1, "jpeg_range2": [35, 95],
             "kernel_list": _K, "kernel_prob": _P, "sinc_prob": 0.1, "blur_sigma": [0.2, 3],
             "betag_range": [0.5, 4], "betap_range": [1, 2], "kernel_list2": _K, "kernel_prob2": _P,
             "sinc_prob2": 0.1, "blur_sigma2": [0.2, 1.5], "betag_range2": [0.5, 4], "betap_range2": [1, 2],
             "final_sinc_prob": 0.8}


def timeit(fn, iters=10, warm=2):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    img = torch.rand(B, 3, 512, 512, device=DEV)
    k = torch.rand(B, 21, 21, device=DEV)
    k /= k.sum((1, 2), keepdim=True)
    nbytes = img.numel() * 4
    rows = []

    def add(name, t, rd_wr_bytes, flops=None):
        rows.append((name, t * 1e3, rd_wr_bytes / t / 1e9, (flops / t / 1e12) if flops else None))

    t = timeit(lambda: D.filter2d(img, k))
    add("filter2d 21x21 (B,3,512,512)", t, 2 * nbytes, 2.0 * img.numel() * 441)
    for mode in ("area", "bilinear", "bicubic"):
        t = timeit(lambda: D.resize(img, scale_factor=0.73, mode=mode))
        add(f"resize {mode} x0.73", t, nbytes * (1 + 0.73**2))
    noise = torch.randn_like(img)
    sig = torch.rand(B, device=DEV) * 2
    gr = torch.zeros(B, device=DEV)
    t = timeit(lambda: D.gaussian_noise(img, noise, None, sig, gr))
    add("gaussian_noise apply", t, 3 * nbytes)
    t = timeit(lambda: D.poisson_rate(img, gray=False))
    add("poisson_rate (bitmap+rate)", t, 3 * nbytes)
    rate, vals = D.poisson_rate(img, gray=False)
    P = torch.poisson(rate)
    t = timeit(lambda: D.poisson_noise(img, P, vals, None, None, sig, gr))
    add("poisson_noise apply", t, 3 * nbytes)
    q = torch.rand(B, device=DEV) * 55 + 40
    t = timeit(lambda: D.diffjpeg(img, q))
    add("diffjpeg fused", t, 2 * nbytes)
    t = timeit(lambda: D.quantize_u8(img))
    add("quantize_u8", t, 2 * nbytes)
    t = timeit(lambda: torch.randn_like(img))
    add("[torch] randn field", t, nbytes)
    t = timeit(lambda: torch.poisson(rate))
    add("[torch] poisson draw", t, 2 * nbytes)
    print(f"B={B}  tensor (B,3,512,512) = {nbytes / 1e6:.0f} MB")
    print(f"{'kernel':34s} {'ms':>8s} {'algo GB/s':>10s} {'%8TB/s':>7s} {'TFLOP/s':>8s}")
    for name, ms, gbs, tf in rows:
        print(f"{name:34s} {ms:8.3f} {gbs:10.0f} {100 * gbs / 8000:7.1f} {'' if tf is None else f'{tf:8.2f}'}")

    # one full feed_data through the model plugin
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import set_global_opt
    import logging
    logging.getLogger("neosr").setLevel(logging.WARNING)
    deg = DEG_TABLE
    _unused = {"resize_prob": [0.3, 0.4, 0.3], "resize_range": [0.5, 1.5], "gaussian_noise_prob": 0.2,
           "noise_range": [0, 2], "poisson_scale_range": [0.05, 0.25], "gray_noise_prob": 0.1,
           "jpeg_range": [40, 95], "second_blur_prob": 0.4, "resize_prob2": [0.3, 0.4, 0.3],
           "resize_range2": [0.3, 1.5], "gaussian_noise_prob2": 0.2, "noise_range2": [0, 2],
           "poisson_scale_range2": [0.05, 0.1], "gray_noise_prob2": 0.1, "jpeg_range2": [35, 95],
           "kernel_list": ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"],
           "kernel_prob": [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], "sinc_prob": 0.1, "blur_sigma": [0.2, 3],
           "betag_range": [0.5, 4], "betap_range": [1, 2],
           "kernel_list2": ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"],
           "kernel_prob2": [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], "sinc_prob2": 0.1, "blur_sigma2": [0.2, 1.5],
           "betag_range2": [0.5, 4], "betap_range2": [1, 2], "final_sinc_prob": 0.8}
    opt = {"name": "bench_otf", "model_type": "otf", "scale": 4, "manual_seed": 1024, "is_train": True,
           "dist": False, "rank": 0, "world_size": 1, "num_gpu": 1, "degradations": deg,
           "datasets": {"train": {"type": "otf", "patch_size": 64, "batch_size": B, "queue_size": 180}},
           "path": {}, "network_g": {"type": "compact", "num_feat": 16, "num_conv": 1},
           "train": {"ema": -1, "grad_clip": True, "optim_g": {"type": "adamw", "lr": 1e-4},
                     "pixel_opt": {"type": "L1Loss"}}, "logger": {"total_iter": 10}}
    set_global_opt(opt)
    random.seed(1024)
    model = build_model(opt)
    sampler = KernelSampler(np.random.default_rng(1024))
    batch = {"gt": img, **{k_: v.to(DEV) for k_, v in sampler.otf_kernel_batch(deg, B).items()}}
    for _ in range(3):
        model.feed_data(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        model.feed_data(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"otf.feed_data (B={B}, 512^2 -> 128^2 -> 64^2 crop, random branches): {dt * 1e3:.2f} ms/call "
          f"= {B / dt:.0f} LR-patches/s of degradation throughput")


if __name__ == "__main__":
    main()
