#!/usr/bin/env python
"""Host (enqueue) time of the U-Net-SN discriminator and the VGG19 tap stack, op-by-op composed (hip/layers.py) — what
VERDICT r4 #5 wants as C++ plans.  Device drained before every call so that the launching thread is never throttled; prints
ms per call and the cProfile top of ONE forward + backward.   python tools/host_unet_vgg.py [batch]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from neosr_amd.archs import build_network  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda"
torch.manual_seed(0)
d = build_network({"type": "unet"}).to(dev).train()
x = torch.rand(B, 3, 256, 256, device=dev, requires_grad=True)


def run_d(full):
    for p in d.parameters():
        p.requires_grad_(full)
    y = d(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = d(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    g = torch.ones_like(y)
    t2 = time.perf_counter()
    y.backward(g)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e3 * (t1 - t0), 1e3 * (t3 - t2)


for full in (True, False):
    r = [run_d(full) for _ in range(5)]
    print(f"unet B={B} param-grads={full}: forward enqueue {min(a for a, _ in r):.2f} ms, backward enqueue {min(b for _, b in r):.2f} ms")

from neosr_amd.losses import build_loss  # noqa: E402

vl = build_loss({"type": "vgg_perceptual_loss", "loss_weight": 1.0}).to(dev)
gt = torch.rand(B, 3, 256, 256, device=dev)


def run_v():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = vl(x, gt)
    loss = loss[0] if isinstance(loss, tuple) else loss
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e3 * (t1 - t0), 1e3 * (t3 - t2)


run_v()
r = [run_v() for _ in range(5)]
print(f"vgg perceptual B={B}: forward (pred + target) enqueue {min(a for a, _ in r):.2f} ms, backward enqueue {min(b for _, b in r):.2f} ms")

for p in d.parameters():
    p.requires_grad_(True)
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
y = d(x)
y.backward(torch.ones_like(y))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
