"""CPU oracle for the batch augmentations — TEST INFRASTRUCTURE ONLY.

PyTorch-CPU restatement of neosr/data/augmentations.py (mixup :13-43, cutmix :46-92, resizemix :95-160,
cutblur :163-204, apply_augment :207-310) as a deterministic function of a recorded draw sequence (any
object with the `neosr_amd.data.draws` interface: `choice`, `choices`, `random`, `integers`, `uniform`,
`randperm`).  Resizes are `F.interpolate(..., antialias=True)` on CPU.

Parity status: PINNED against tests/golden/aug.npz (reference run on CPU with every draw recorded,
tests/golden/gen_golden_aug.py).
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _bbox(size, cut_w, cut_h, d):
    W, H = size[2], size[3]
    cx, cy = d.integers(W), d.integers(H)
    return (int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cy - cut_h // 2, 0, H)),
            int(np.clip(cx + cut_w // 2, 0, W)), int(np.clip(cy + cut_h // 2, 0, H)))


def mixup(gt, lq, d, alpha_min=0.4, alpha_max=0.6):
    lam = d.uniform(alpha_min, alpha_max)
    idx = d.randperm(gt.size(0))
    img_ = gt[idx]
    return lam * gt + (1 - lam) * img_, lam * lq + (1 - lam) * img_  # LQ is blended with the permuted GT


def cutmix(gt, lq, d, alpha=0.9):
    lam = d.uniform(0, alpha)
    idx = d.randperm(gt.size(0))
    gt_, lq_ = gt[idx], lq[idx]
    rat = np.sqrt(1.0 - lam)
    x1, y1, x2, y2 = _bbox(gt.size(), int(gt.size(2) * rat), int(gt.size(3) * rat), d)
    gt, lq = gt.clone(), lq.clone()
    gt[:, :, x1:x2, y1:y2] = gt_[:, :, x1:x2, y1:y2]
    lq[:, :, x1:x2, y1:y2] = lq_[:, :, x1:x2, y1:y2]
    return gt, lq


def resizemix(gt, lq, d, scope=(0.2, 0.9)):
    idx = d.randperm(gt.size(0))
    gt_r, lq_r = gt.clone()[idx], lq.clone()[idx]
    tao = d.uniform(scope[0], scope[1])
    x1, y1, x2, y2 = _bbox(gt.size(), int(gt.size(2) * tao), int(gt.size(3) * tao), d)
    gt_r = torch.clamp(F.interpolate(gt_r, (y2 - y1, x2 - x1), mode="bicubic", antialias=True), 0, 1)
    lq_r = torch.clamp(F.interpolate(lq_r, (y2 - y1, x2 - x1), mode="bicubic", antialias=True), 0, 1)
    gt, lq = gt.clone(), lq.clone()
    gt[:, :, y1:y2, x1:x2] = gt_r
    lq[:, :, y1:y2, x1:x2] = lq_r
    return gt, lq


def cutblur(gt, lq, d, alpha=0.7):
    lam = d.uniform(0.2, alpha)
    x1, y1, x2, y2 = _bbox(gt.size(), int(gt.size(2) * lam), int(gt.size(3) * lam), d)
    lq = lq.clone()
    lq[:, :, x1:x2, y1:y2] = gt[:, :, x1:x2, y1:y2]
    return gt, lq


@torch.no_grad()
def apply_augment(gt, lq, d, scale=1, augs=("none", "mixup", "cutmix", "resizemix", "cutblur"),
                  prob=(0.1, 0.3, 0.2, 0.7, 0.8), multi_prob=0.3):
    if scale > 1:
        lq = torch.clamp(F.interpolate(lq, scale_factor=scale, mode=d.choice(["bilinear", "bicubic"]), antialias=True), 0, 1)
    fns = {"cutmix": cutmix, "mixup": mixup, "resizemix": resizemix, "cutblur": cutblur}
    if d.random() < multi_prob:
        n = d.integers(2, len(augs)) if len(augs) > 2 else len(augs)
        remaining, picked = list(augs), []
        for _ in range(n):
            p = d.choices(remaining)
            picked.append(p)
            remaining.remove(p)
        for name in ("cutmix", "mixup", "resizemix", "cutblur"):
            if name in picked:
                gt, lq = fns[name](gt, lq, d)
    else:
        aug = d.choices(list(augs), list(prob))
        for name in ("cutmix", "mixup", "resizemix", "cutblur"):
            if name in aug:
                gt, lq = fns[name](gt, lq, d)
                break
    if scale > 1:
        lq = torch.clamp(F.interpolate(lq, scale_factor=1 / scale, mode="bicubic", antialias=True), 0, 1)
    return gt, lq
