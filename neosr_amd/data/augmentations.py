"""Batch augmentations on the HIP device (drop-in for neosr/data/augmentations.py:13-310).

`apply_augment` keeps the reference's control flow and RNG consumption order — python `random`
(`choice` of the up-sampling mode, `choices` of the augmentation), the numpy `Generator`
(`random`, `integers`, `uniform`) and `torch.randperm` — all routed through a `draws` object
(neosr_amd/data/draws.py) so that a recorded reference run can be replayed.  Pixels move through
`neosr_resize_aa` (antialiased bilinear / bicubic, ATen weights) and `neosr_box_blend`.
Reference quirks kept on purpose: mixup blends the LQ with the permuted *GT* (augmentations.py:40-42);
the multi-augmentation branch draws uniformly, not by `prob` (:272); boxes are indexed [x-range, y-range]
in cutmix / cutblur but [y-range, x-range] in resizemix.
"""

from __future__ import annotations

import numpy as np
import torch

from neosr_amd import _C

_BILINEAR, _BICUBIC = 1, 2  # NEOSR_RESIZE_* (include/neosr_amd.h)


def _st():
    return _C.stream_ptr()


def resize_aa(x, out_h: int, out_w: int, mode: str, *, clamp: bool = True, perm=None, into=None, y0: int = 0,
              x0: int = 0):
    """F.interpolate(x, (out_h, out_w), mode, antialias=True) [+ clamp(0,1)], optionally reading batch entry
    perm[b] and writing into the box (y0, x0) of `into`."""
    lib = _C.load()
    x = _C.require_device(x, "x").contiguous()
    B, C_, H, W = x.shape
    out = torch.empty(B, C_, out_h, out_w, device=x.device, dtype=torch.float32) if into is None else into
    tmp = torch.empty(B * C_ * H * out_w, device=x.device, dtype=torch.float32)
    _C.check(lib.neosr_resize_aa(x.data_ptr(), out.data_ptr(), tmp.data_ptr(), None if perm is None else perm.data_ptr(),
                                 B, C_, H, W, out_h, out_w, out.shape[2], out.shape[3], y0, x0,
                                 _BILINEAR if mode == "bilinear" else _BICUBIC, int(clamp), _st()), "neosr_resize_aa")
    return out


def box_blend(x, src, perm, box, lam: float):
    """out = inside box (y0, y1, x0, x1) ? lam * x + (1 - lam) * src[perm] : x"""
    lib = _C.load()
    x, src = _C.require_device(x, "x").contiguous(), _C.require_device(src, "src").contiguous()
    B, C_, H, W = x.shape
    out = torch.empty_like(x)
    y0, y1, x0, x1 = box
    _C.check(lib.neosr_box_blend(x.data_ptr(), src.data_ptr(), None if perm is None else perm.data_ptr(), out.data_ptr(),
                                 B, C_, H, W, y0, y1, x0, x1, lam, _st()), "neosr_box_blend")
    return out


def _perm32(draws, n: int, device) -> torch.Tensor:
    return draws.randperm(n).to(device=device, dtype=torch.int32)


def mixup(img_gt, img_lq, draws, alpha_min: float = 0.4, alpha_max: float = 0.6):
    if img_gt.size() != img_lq.size():
        raise ValueError("img_gt and img_lq have to be the same resolution.")
    lam = draws.uniform(alpha_min, alpha_max)
    perm = _perm32(draws, img_gt.size(0), img_gt.device)
    full = (0, img_gt.shape[2], 0, img_gt.shape[3])
    return box_blend(img_gt, img_gt, perm, full, lam), box_blend(img_lq, img_gt, perm, full, lam)


def _rand_bbox(size, cut_w: int, cut_h: int, draws):
    W, H = size[2], size[3]
    cx, cy = draws.integers(W), draws.integers(H)
    return (int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cy - cut_h // 2, 0, H)),
            int(np.clip(cx + cut_w // 2, 0, W)), int(np.clip(cy + cut_h // 2, 0, H)))


def cutmix(img_gt, img_lq, draws, alpha: float = 0.9):
    if img_gt.size() != img_lq.size():
        raise ValueError("img_gt and img_lq have to be the same resolution.")
    lam = draws.uniform(0, alpha)
    perm = _perm32(draws, img_gt.size(0), img_gt.device)
    cut_rat = np.sqrt(1.0 - lam)
    bbx1, bby1, bbx2, bby2 = _rand_bbox(img_gt.size(), int(img_gt.size(2) * cut_rat), int(img_gt.size(3) * cut_rat), draws)
    box = (bbx1, bbx2, bby1, bby2)  # the reference slices dim 2 with the "x" pair and dim 3 with the "y" pair
    return box_blend(img_gt, img_gt, perm, box, 0.0), box_blend(img_lq, img_lq, perm, box, 0.0)


def resizemix(img_gt, img_lq, draws, scope=(0.2, 0.9)):
    if img_gt.size() != img_lq.size():
        raise ValueError("img_gt and img_lq have to be the same resolution.")
    perm = _perm32(draws, img_gt.size(0), img_gt.device)
    tao = draws.uniform(scope[0], scope[1])
    bbx1, bby1, bbx2, bby2 = _rand_bbox(img_gt.size(), int(img_gt.size(2) * tao), int(img_gt.size(3) * tao), draws)
    if bby2 - bby1 <= 0 or bbx2 - bbx1 <= 0:
        raise RuntimeError("resizemix: empty box (the reference fails in F.interpolate here as well)")
    gt, lq = img_gt.clone(), img_lq.clone()
    resize_aa(img_gt, bby2 - bby1, bbx2 - bbx1, "bicubic", perm=perm, into=gt, y0=bby1, x0=bbx1)
    resize_aa(img_lq, bby2 - bby1, bbx2 - bbx1, "bicubic", perm=perm, into=lq, y0=bby1, x0=bbx1)
    return gt, lq


def cutblur(img_gt, img_lq, draws, alpha: float = 0.7):
    if img_gt.size() != img_lq.size():
        raise ValueError("img_gt and img_lq have to be the same resolution.")
    lam = draws.uniform(0.2, alpha)
    bbx1, bby1, bbx2, bby2 = _rand_bbox(img_gt.size(), int(img_gt.size(2) * lam), int(img_gt.size(3) * lam), draws)
    return img_gt, box_blend(img_lq, img_gt, None, (bbx1, bbx2, bby1, bby2), 0.0)


@torch.no_grad()
def apply_augment(img_gt, img_lq, draws, scale: int = 1, augs=("none", "mixup", "cutmix", "resizemix", "cutblur"),
                  prob=(0.1, 0.3, 0.2, 0.7, 0.8), multi_prob: float = 0.3):
    """augmentations.py:219-310."""
    if len(augs) != len(prob):
        raise ValueError("Length of 'augmentation' and aug_prob don't match!")
    if img_gt.shape[0] == 1:
        raise ValueError("Augmentations need batch >1 to work.")
    if scale > 1:
        mode = draws.choice(["bilinear", "bicubic"])
        img_lq = resize_aa(img_lq, img_lq.shape[2] * scale, img_lq.shape[3] * scale, mode)
    if draws.random() < multi_prob:
        num_augs = draws.integers(2, len(augs)) if len(augs) > 2 else len(augs)
        remaining = list(augs)
        aug = []
        for _ in range(num_augs):
            pick = draws.choices(remaining)
            aug.append(pick)
            remaining.remove(pick)
        if "cutmix" in aug:
            img_gt, img_lq = cutmix(img_gt, img_lq, draws)
        if "mixup" in aug:
            img_gt, img_lq = mixup(img_gt, img_lq, draws)
        if "resizemix" in aug:
            img_gt, img_lq = resizemix(img_gt, img_lq, draws)
        if "cutblur" in aug:
            img_gt, img_lq = cutblur(img_gt, img_lq, draws)
    else:
        aug = draws.choices(list(augs), list(prob))
        if "cutmix" in aug:
            img_gt, img_lq = cutmix(img_gt, img_lq, draws)
        elif "mixup" in aug:
            img_gt, img_lq = mixup(img_gt, img_lq, draws)
        elif "resizemix" in aug:
            img_gt, img_lq = resizemix(img_gt, img_lq, draws)
        elif "cutblur" in aug:
            img_gt, img_lq = cutblur(img_gt, img_lq, draws)
    if scale > 1:
        img_lq = resize_aa(img_lq, img_lq.shape[2] // scale, img_lq.shape[3] // scale, "bicubic")
    return img_gt, img_lq
