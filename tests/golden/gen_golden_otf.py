#!/usr/bin/env python
"""Golden fixtures for the on-the-fly degradation bank, produced by RUNNING THE REFERENCE on CPU
(build container only):  python tests/golden/gen_golden_otf.py

  degrade_prims.npz  filter2D / F.interpolate call forms / noise fns (with their draws captured) /
                     DiffJPEG / quantise, each called through the reference's own functions
  otf_feed.npz       three `otf.feed_data` calls (B=2, 128^2 GT -> 32^2 LQ -> 16^2 crop, queue 4 so
                     the pair pool fills and shuffles) with EVERY random draw recorded in order
Same shims / cuda->cpu redirect as gen_golden.py.  Data only; no reference source is copied.
"""

from __future__ import annotations

import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
from gen_golden import REF, install_reference, save  # noqa: E402

DEG = """
resize_prob = [ 0.3, 0.4, 0.3 ]
resize_range = [ 0.5, 1.5 ]
gaussian_noise_prob = 0.5
noise_range = [ 0, 2 ]
poisson_scale_range = [ 0.05, 0.25 ]
gray_noise_prob = 0.4
jpeg_range = [ 40, 95 ]
second_blur_prob = 0.5
resize_prob2 = [ 0.3, 0.4, 0.3 ]
resize_range2 = [ 0.3, 1.5 ]
gaussian_noise_prob2 = 0.5
noise_range2 = [ 0, 2 ]
poisson_scale_range2 = [ 0.05, 0.1 ]
gray_noise_prob2 = 0.4
jpeg_range2 = [ 35, 95 ]
"""

TOML = f"""
name = "golden_otf"
model_type = "otf"
scale = 4
manual_seed = 1024

[datasets.train]
type = "otf"
dataroot_gt = "/tmp/none_gt"
patch_size = 16
batch_size = 2
queue_size = 4

[degradations]
{DEG}

[path]

[network_g]
type = "compact"
num_feat = 8
num_conv = 1

[train]
ema = 0.999
grad_clip = true

[train.optim_g]
type = "adamw"
lr = 1e-3

[train.pixel_opt]
type = "L1Loss"

[logger]
total_iter = 10
save_checkpoint_freq = 1000
use_tb_logger = false
"""


class Recorder:
    """Wraps the RNG entry points the reference uses inside feed_data and logs (kind, value)."""

    def __init__(self) -> None:
        self.log: list[tuple[str, object]] = []
        self.on = False

    def rec(self, kind, value):
        if self.on:
            if isinstance(value, torch.Tensor):
                value = value.detach().cpu().numpy().copy()
            self.log.append((kind, value))

    def install(self, otf_mod, transforms_mod):
        R = self
        py_random = random

        class PyRandom:
            def choices(self, pop, weights=None, **kw):
                v = py_random.choices(pop, weights, **kw)
                R.rec("choices", v[0])
                return v

            def choice(self, seq):
                v = py_random.choice(seq)
                R.rec("choice", v)
                return v

            def randint(self, a, b):
                v = py_random.randint(a, b)
                R.rec("randint", v)
                return v

        otf_mod.random = PyRandom()
        transforms_mod.random = PyRandom()

        np_rng = otf_mod.rng

        class NpRng:
            def uniform(self, *a, **k):
                v = np_rng.uniform(*a, **k)
                R.rec("uniform", float(v))
                return v

        otf_mod.rng = NpRng()

        def wrap(name, kind):
            orig = getattr(torch, name)

            def inner(*a, **k):
                v = orig(*a, **k)
                R.rec(kind, v)
                return v

            setattr(torch, name, inner)

        for n in ("rand", "randn", "poisson", "randperm"):
            wrap(n, n)
        orig_uniform_ = torch.Tensor.uniform_

        def uniform_(self, *a, **k):
            v = orig_uniform_(self, *a, **k)
            R.rec("uniform_", v)
            return v

        torch.Tensor.uniform_ = uniform_

    def drain(self):
        log, self.log = self.log, []
        return log


def pack_draws(arrays: dict, prefix: str, log) -> None:
    kinds = []
    for i, (kind, val) in enumerate(log):
        kinds.append(kind)
        if isinstance(val, str):
            arrays[f"{prefix}/{i:03d}"] = np.array(val)
        else:
            arrays[f"{prefix}/{i:03d}"] = np.asarray(val)
    arrays[f"{prefix}/kinds"] = np.array(kinds)


def smooth_images(gen, b, h, w):
    """low-pass random images (so blur / JPEG have structure to act on), in [0,1]"""
    x = torch.rand(b, 3, h // 8, w // 8, generator=gen)
    x = torch.nn.functional.interpolate(x, size=(h, w), mode="bicubic", align_corners=False)
    return (x + 0.05 * torch.rand(b, 3, h, w, generator=gen)).clamp(0, 1).contiguous()


def main():
    tmp = Path(tempfile.mkdtemp()) / "golden_otf.toml"
    tmp.write_text(TOML)
    (HERE / "golden_otf.toml").write_text(TOML)
    install_reference(str(tmp))
    import neosr.data.degradations as degr
    import neosr.data.transforms as transforms_mod
    import neosr.models.otf as otf_mod
    import neosr.utils.diffjpeg as dj
    from neosr.models import build_model
    from neosr.utils.options import parse_options

    dj.device = torch.device("cpu")
    orig_filter2d = dj.filter2D
    otf_mod.filter2D = lambda img, k: orig_filter2d(img.contiguous(), k)

    R = Recorder()
    R.install(otf_mod, transforms_mod)
    gen = torch.Generator().manual_seed(2024)
    rng = np.random.default_rng(7)

    # blur kernels through the reference's own generators (21x21, like the otf dataset emits)
    def kernels(b):
        ks = []
        for _ in range(b):
            size = int(rng.choice([7, 9, 11, 13, 15, 17, 19, 21]))
            k = degr.random_mixed_kernels(
                ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"],
                [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], size, [0.2, 3], [0.2, 3], [-np.pi, np.pi],
                [0.5, 4], [1, 2], noise_range=None)
            p = (21 - size) // 2
            ks.append(np.pad(k, ((p, p), (p, p))))
        return torch.from_numpy(np.stack(ks)).float()

    def sinc(b):
        ks = []
        for _ in range(b):
            size = int(rng.choice([7, 9, 11, 13, 15, 17, 19, 21]))
            ks.append(degr.circular_lowpass_kernel(rng.uniform(np.pi / 3, np.pi), size, pad_to=21))
        return torch.from_numpy(np.stack(ks)).float()

    # ---------------------------------------------------------------- primitives
    A = {}
    img = smooth_images(gen, 2, 50, 70)
    k21 = kernels(2)
    A["f2d_img"], A["f2d_k"] = img.numpy(), k21.numpy()
    A["f2d_out"] = orig_filter2d(img, k21).numpy()
    k7 = torch.from_numpy(np.stack([degr.bivariate_Gaussian(7, 1.2, 2.0, 0.4, isotropic=False)])).float()
    A["f2d_k1"] = k7.numpy()
    A["f2d_out1"] = orig_filter2d(img, k7).numpy()   # single kernel for the whole batch
    for mode in ("area", "bilinear", "bicubic"):
        for s in (0.5, 0.73, 1.37):
            A[f"rs_sf_{mode}_{s}"] = torch.nn.functional.interpolate(img, scale_factor=s, mode=mode).numpy()
        for size in ((25, 35), (32, 32), (61, 90)):
            A[f"rs_sz_{mode}_{size[0]}x{size[1]}"] = torch.nn.functional.interpolate(img, size=size, mode=mode).numpy()
    # noise functions with their draws recorded
    R.on = True
    torch.manual_seed(5)
    out = degr.random_add_gaussian_noise_pt(img, sigma_range=(1, 30), clip=True, rounds=False, gray_prob=0.6)
    pack_draws(A, "gn_draws", R.drain())
    A["gn_out"] = out.numpy()
    out = degr.random_add_gaussian_noise_pt(img, sigma_range=(1, 30), clip=True, rounds=False, gray_prob=0.0)
    pack_draws(A, "gn0_draws", R.drain())
    A["gn0_out"] = out.numpy()
    out = degr.random_add_poisson_noise_pt(img, scale_range=(0.05, 3), gray_prob=0.6, clip=True, rounds=False)
    pack_draws(A, "pn_draws", R.drain())
    A["pn_out"] = out.numpy()
    out = degr.random_add_poisson_noise_pt(img, scale_range=(0.05, 3), gray_prob=0.0, clip=True, rounds=False)
    pack_draws(A, "pn0_draws", R.drain())
    A["pn0_out"] = out.numpy()
    R.on = False
    jpeger = dj.DiffJPEG(differentiable=False)
    for name, im in (("a", smooth_images(gen, 4, 64, 48)), ("b", smooth_images(gen, 2, 37, 53))):
        q = torch.tensor([35.0, 50.0, 72.5, 95.0][: im.size(0)])
        A[f"jpg_{name}_img"], A[f"jpg_{name}_q"] = im.numpy(), q.numpy().copy()
        A[f"jpg_{name}_out"] = jpeger(im.clone(), quality=q.clone()).detach().contiguous().numpy()
    x = torch.rand(2, 3, 9, 11, generator=gen) * 1.2 - 0.1
    A["q_in"], A["q_out"] = x.numpy(), (torch.clamp((x * 255.0).round(), 0, 255) / 255.0).numpy()
    A["qf_q"] = np.arange(1, 101, dtype=np.float32)
    A["qf_f"] = np.array([dj.quality_to_factor(float(q)) for q in range(1, 101)], dtype=np.float32)
    save("degrade_prims.npz", **A)

    # ---------------------------------------------------------------- full feed_data
    opt, _ = parse_options(str(REF), is_train=True)
    opt["datasets"]["train"].update(opt["degradations"])  # what train.py:69-70 does
    torch.manual_seed(1024)
    random.seed(1024)
    model = build_model(opt)
    model.device = torch.device("cpu")
    F_ = {}
    for it in range(1, 4):
        batch = {"gt": smooth_images(gen, 2, 128, 128), "kernel1": kernels(2), "kernel2": kernels(2),
                 "sinc_kernel": sinc(2)}
        for k, v in batch.items():
            F_[f"it{it}/{k}"] = v.numpy()
        R.on = True
        model.feed_data(batch)
        R.on = False
        pack_draws(F_, f"it{it}/draws", R.drain())
        F_[f"it{it}/lq"] = model.lq.numpy().copy()
        F_[f"it{it}/gt_out"] = model.gt.numpy().copy()
    save("otf_feed.npz", **F_)


if __name__ == "__main__":
    main()
