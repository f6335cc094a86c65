"""``otf`` — on-the-fly degradation model (drop-in for neosr/models/otf.py:23-291).

`feed_data` turns a GT batch + per-sample blur/sinc kernels into a (LQ, GT) training pair with the
Real-ESRGAN 2nd-order pipeline, entirely on the HIP device through `libneosr_amd`
(`neosr_filter2d`, `neosr_resize`, `neosr_gaussian_noise`, `neosr_poisson_rate/noise`,
`neosr_diffjpeg`, `neosr_quantize_u8`, `neosr_crop`, `neosr_gather_rows`).  Control flow and the
ORDER in which the three RNG families are consumed follow the reference line by line, so that
replaying the reference's recorded draws reproduces its output (tests/test_hip_degrade.py).

MI355X-first differences: the fused JPEG kernel reads/writes the image once (the reference
launches ~25 small kernels and syncs the host once per sample for `quality_to_factor`); the
Poisson `vals` come from a 256-bit level bitmap built with integer atomics (the reference sorts
every image with `torch.unique` in a Python loop and syncs twice per sample); no `.item()`
except the one data-dependent branch the reference also takes (`torch.sum(gray_noise) > 0`).
"""

from __future__ import annotations

from typing import Any

import torch

from neosr_amd.data.augmentations import apply_augment
from neosr_amd.data.draws import LiveDraws
from neosr_amd.hip import degrade as D
from neosr_amd.models.image import image
from neosr_amd.utils.misc import tc
from neosr_amd.utils.registry import MODEL_REGISTRY

_MODES = ["area", "bilinear", "bicubic"]


@MODEL_REGISTRY.register()
class otf(image):
    """On The Fly degradations, based on the RealESRGAN pipeline."""

    def __init__(self, opt: dict[str, Any]) -> None:
        super().__init__(opt)
        ds = opt["datasets"]["train"]
        queue = ds.get("queue_size", 180)
        batch = ds["batch_size"]
        self.queue_size: int = (queue // batch) * batch
        self.patch_size = ds.get("patch_size")
        self.device = torch.device("cuda")
        self.draws = LiveDraws(opt.get("manual_seed"), self.device)
        # train.py:69-70 merges [degradations] into datasets.train; accept either layout
        self.dopt = dict(opt.get("degradations") or {})
        for k, v in ds.items():
            self.dopt.setdefault(k, v)

    # ------------------------------------------------------------------------------------
    @torch.no_grad()
    def _dequeue_and_enqueue(self) -> None:
        """Training-pair pool (otf.py:37-90): enqueue until full, then shuffle + swap a batch."""
        b = self.lq.size(0)
        if not hasattr(self, "queue_lr"):
            assert self.queue_size % b == 0, (
                f"queue size {self.queue_size} should be divisible by batch size {b}")
            self.queue_lr = torch.zeros(self.queue_size, *self.lq.shape[1:], device=self.device)
            self.queue_gt = torch.zeros(self.queue_size, *self.gt.shape[1:], device=self.device)
            self.queue_ptr = 0
        if self.queue_ptr == self.queue_size:
            idx = self.draws.randperm(self.queue_size)
            self.queue_lr = D.gather_rows(self.queue_lr, idx)
            self.queue_gt = D.gather_rows(self.queue_gt, idx)
            lq_dequeue = self.queue_lr[0:b].clone()
            gt_dequeue = self.queue_gt[0:b].clone()
            self.queue_lr[0:b] = self.lq
            self.queue_gt[0:b] = self.gt
            self.lq, self.gt = lq_dequeue, gt_dequeue
        else:
            self.queue_lr[self.queue_ptr:self.queue_ptr + b] = self.lq
            self.queue_gt[self.queue_ptr:self.queue_ptr + b] = self.gt
            self.queue_ptr += b

    # ------------------------------------------------------------------------------------
    def _updown_scale(self, suffix: str) -> float:
        kind = self.draws.choices(["up", "down", "keep"], self.dopt.get(f"resize_prob{suffix}"))
        lo, hi = self.dopt.get(f"resize_range{suffix}")
        if kind == "up":
            return self.draws.uniform(1, hi)
        if kind == "down":
            return self.draws.uniform(lo, 1)
        return 1

    @property
    def _no_sync(self) -> bool:
        return isinstance(self.draws, LiveDraws) and self.draws.device.type == "cuda"

    def _add_noise(self, out: torch.Tensor, suffix: str) -> torch.Tensor:
        """gaussian (p = gaussian_noise_prob) else poisson; clip, no round (otf.py:128-148,188-210)."""
        d = self.draws
        gray_prob = self.dopt.get(f"gray_noise_prob{suffix}")
        b, _, h, w = out.shape
        if d.uniform() < self.dopt.get(f"gaussian_noise_prob{suffix}"):
            lo, hi = self.dopt.get(f"noise_range{suffix}")
            sigma = d.rand(b) * (hi - lo) + lo
            gray = (d.rand(b) < gray_prob).float()
            # the reference draws the shared gray field only if some sample is gray (a device->host sync,
            # degradations.py:593-598).  A replayed stream must follow that; the live stream (our own counter-based
            # sampler, not torch's sequence anyway) draws it unconditionally and never stalls the host
            noise_gray = d.randn(h, w) if (self._no_sync or bool(gray.sum() > 0)) else None
            noise = d.randn(b, 3, h, w)
            return D.gaussian_noise(out, noise, noise_gray, sigma, gray)
        lo, hi = self.dopt.get(f"poisson_scale_range{suffix}")
        scale = d.rand(b) * (hi - lo) + lo
        gray = (d.rand(b) < gray_prob).float()
        p_gray = vals_gray = None
        if self._no_sync or bool(gray.sum() > 0):
            rate_g, vals_gray = D.poisson_rate(out, gray=True)
            p_gray = d.poisson(rate_g)
        rate, vals = D.poisson_rate(out, gray=False)
        p = d.poisson(rate)
        return D.poisson_noise(out, p, vals, p_gray, vals_gray, scale, gray)

    @torch.no_grad()
    def feed_data(self, data: dict[str, Any]) -> None:
        """Accept data from the dataloader, then add two-order degradations to obtain LQ images."""
        if not self.is_train:
            self.lq = data["lq"].to(self.device, non_blocking=True)
            if "gt" in data:
                self.gt = data["gt"].to(self.device, non_blocking=True)
            return
        d = self.draws
        scale = self.opt["scale"]
        to = lambda t: t.to(device=self.device, dtype=torch.float32, non_blocking=True)  # noqa: E731
        self.gt = to(data["gt"])
        self.kernel1, self.kernel2 = to(data["kernel1"]), to(data["kernel2"])
        self.sinc_kernel = to(data["sinc_kernel"])
        b = self.gt.size(0)
        ori_h, ori_w = self.gt.shape[2:4]

        # ----------------------- the first degradation process -----------------------
        out = D.filter2d(self.gt, self.kernel1)
        s = self._updown_scale("")
        out = D.resize(out, scale_factor=s, mode=d.choice(_MODES))
        out = self._add_noise(out, "")
        jpeg_p = d.uniform_tensor(b, *self.dopt.get("jpeg_range"))
        out = D.diffjpeg(D.clamp01(out), jpeg_p)

        # ----------------------- the second degradation process ----------------------
        if d.uniform() < self.dopt.get("second_blur_prob"):
            out = D.filter2d(out, self.kernel2)
        s = self._updown_scale("2")
        out = D.resize(out, size=(int(ori_h / scale * s), int(ori_w / scale * s)), mode=d.choice(_MODES))
        out = self._add_noise(out, "2")
        final = (ori_h // scale, ori_w // scale)
        if d.uniform() < 0.5:
            # [resize back + sinc filter] + JPEG
            out = D.resize(out, size=final, mode=d.choice(_MODES))
            out = D.filter2d(out, self.sinc_kernel)
            jpeg_p = d.uniform_tensor(b, *self.dopt.get("jpeg_range2"))
            out = D.diffjpeg(D.clamp01(out), jpeg_p)
        else:
            # JPEG + [resize back + sinc filter]
            jpeg_p = d.uniform_tensor(b, *self.dopt.get("jpeg_range2"))
            out = D.diffjpeg(D.clamp01(out), jpeg_p)
            out = D.resize(out, size=final, mode=d.choice(_MODES))
            out = D.filter2d(out, self.sinc_kernel)

        lq = D.quantize_u8(out)  # clamp((out*255).round(), 0, 255)/255

        # random crop: one window for the whole batch (transforms.py:100-119)
        patch = self.dopt.get("patch_size", self.patch_size)
        h_lq, w_lq = lq.shape[2:4]
        if ori_h != h_lq * scale or ori_w != w_lq * scale:
            raise ValueError(f"{tc.red}Scale mismatches. GT ({ori_h}, {ori_w}) is not {scale}x of "
                             f"LQ ({h_lq}, {w_lq}).{tc.end}")
        if h_lq < patch or w_lq < patch:
            raise ValueError(f"{tc.red}LQ ({h_lq}, {w_lq}) is smaller than patch size ({patch}).{tc.end}")
        top = d.randint(0, h_lq - patch)
        left = d.randint(0, w_lq - patch)
        self.lq = D.crop(lq, top, left, patch, patch)
        self.gt = D.crop(self.gt, top * scale, left * scale, patch * scale, patch * scale)

        self._dequeue_and_enqueue()
        self.lq = self.lq.contiguous()
        # otf.py:266-278: with `augmentation` set the batch always goes through apply_augment (its x scale
        # up / down resize round trip happens even when "none" is drawn)
        if self.aug is not None:
            if self.patch_size % 4 != 0:
                raise ValueError(f"{tc.red}The patch_size value must be a multiple of 4 while using augmentations.{tc.end}")
            self.gt, self.lq = apply_augment(self.gt, self.lq, d, scale=self.scale, augs=self.aug, prob=self.aug_prob)
