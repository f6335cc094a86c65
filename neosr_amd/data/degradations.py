"""Blur / sinc kernel synthesis for the on-the-fly degradation pipeline: the random draws (host, reference order) and
two evaluators — numpy fp64 on the host (`otf_kernel_batch`) and `neosr_blur_kernels` on the device
(`otf_kernel_batch_device`, float64 too).

These are the *inputs* of `neosr_filter2d` (`kernel1`, `kernel2`, `sinc_kernel` of the otf batch
dict).  Behaviour follows neosr/data/degradations.py:24-512 and the sampling order of
neosr/data/otf_dataset.py:189-246 (python `random` for discrete choices, one numpy `Generator`
for the continuous draws) so that identically seeded runs emit identical kernels; the arithmetic is
organised differently: every parametric family is `f(q)` of the quadratic form
`q = g^T Sigma^-1 g` on the centred pixel grid.
"""

from __future__ import annotations

import math
import random as _pyrandom

import numpy as np
from scipy import special

KERNEL_SIZES = [2 * v + 1 for v in range(3, 11)]  # 7 .. 21 (otf_dataset.py:117)


def _grid(k: int) -> np.ndarray:
    """(k, k, 2) centred coordinates, x fastest (mesh_grid, degradations.py:46-66)."""
    ax = np.arange(-k // 2 + 1.0, k // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    return np.stack([xx, yy], axis=-1)


def _sigma(sig_x: float, sig_y: float, theta: float, isotropic: bool) -> np.ndarray:
    if isotropic:
        return np.array([[sig_x**2, 0.0], [0.0, sig_x**2]])
    rot = np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    return rot @ np.array([[sig_x**2, 0.0], [0.0, sig_y**2]]) @ rot.T


def _quadratic_form(k: int, sig_x: float, sig_y: float, theta: float, isotropic: bool) -> np.ndarray:
    g = _grid(k)
    inv = np.linalg.inv(_sigma(sig_x, sig_y, theta, isotropic))
    return np.sum(np.dot(g, inv) * g, 2)


def _normalise(kernel: np.ndarray) -> np.ndarray:
    return kernel / np.sum(kernel)


def bivariate_gaussian(k, sig_x, sig_y=None, theta=0.0, isotropic=True) -> np.ndarray:
    """exp(-q/2), normalised (bivariate_Gaussian, degradations.py:100-127)."""
    return _normalise(np.exp(-0.5 * _quadratic_form(k, sig_x, sig_y, theta, isotropic)))


def bivariate_generalized_gaussian(k, sig_x, sig_y, theta, beta, isotropic=True) -> np.ndarray:
    """exp(-q^beta / 2) (degradations.py:130-167)."""
    return _normalise(np.exp(-0.5 * np.power(_quadratic_form(k, sig_x, sig_y, theta, isotropic), beta)))


def bivariate_plateau(k, sig_x, sig_y, theta, beta, isotropic=True) -> np.ndarray:
    """1 / (1 + q^beta) (degradations.py:170-207)."""
    return _normalise(np.reciprocal(np.power(_quadratic_form(k, sig_x, sig_y, theta, isotropic), beta) + 1))


def circular_lowpass_kernel(cutoff: float, k: int, pad_to: int = 0) -> np.ndarray:
    """2-D circularly symmetric sinc: wc*J1(wc*r)/(2*pi*r), centre wc^2/(4*pi) (degradations.py:478-512)."""
    assert k % 2 == 1, "Kernel size must be an odd number."
    c = (k - 1) / 2
    y, x = np.meshgrid(np.arange(k, dtype=np.float64), np.arange(k, dtype=np.float64))
    r = np.sqrt((x - c) ** 2 + (y - c) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        kernel = cutoff * special.j1(cutoff * r) / (2 * np.pi * r)
    kernel[(k - 1) // 2, (k - 1) // 2] = cutoff**2 / (4 * np.pi)
    kernel = kernel / np.sum(kernel)
    if pad_to > k:
        p = (pad_to - k) // 2
        kernel = np.pad(kernel, ((p, p), (p, p)))
    return kernel


class KernelSampler:
    """Random kernels with the reference's draw order.  `rng`: numpy Generator (continuous draws);
    `pyrandom`: object with `choices/choice` (python `random` module by default)."""

    def __init__(self, rng: np.random.Generator, pyrandom=_pyrandom) -> None:
        self.rng = rng
        self.random = pyrandom

    def _shape_params(self, sigma_x_range, sigma_y_range, rotation_range, isotropic):
        assert sigma_x_range[0] < sigma_x_range[1], "Wrong sigma_x_range."
        sx = self.rng.uniform(sigma_x_range[0], sigma_x_range[1])
        if isotropic:
            return sx, sx, 0
        assert sigma_y_range[0] < sigma_y_range[1], "Wrong sigma_y_range."
        assert rotation_range[0] < rotation_range[1], "Wrong rotation_range."
        sy = self.rng.uniform(sigma_y_range[0], sigma_y_range[1])
        return sx, sy, self.rng.uniform(rotation_range[0], rotation_range[1])

    def _beta(self, beta_range):
        if self.rng.uniform() < 0.5:
            return self.rng.uniform(beta_range[0], 1)
        return self.rng.uniform(1, beta_range[1])

    def _noise(self, kernel, noise_range):
        if noise_range is not None:
            assert noise_range[0] < noise_range[1], "Wrong noise range."
            kernel = kernel * self.rng.uniform(noise_range[0], noise_range[1], size=kernel.shape)
        return _normalise(kernel)

    def mixed_params(self, kernel_list, kernel_prob, k, sigma_x_range, sigma_y_range, rotation_range, betag_range,
                     betap_range) -> list[float]:
        """the DRAWS of random_mixed_kernels (degradations.py:410-475) without evaluating the kernel: one row of the
        `neosr_blur_kernels` parameter table {type, k, sig_x, sig_y, theta, beta, isotropic, 0}"""
        assert k % 2 == 1, "Kernel size must be an odd number."
        kind = self.random.choices(kernel_list, kernel_prob)[0]
        iso = not kind.endswith("aniso")
        sx, sy, th = self._shape_params(sigma_x_range, sigma_y_range, rotation_range, iso)
        if kind in ("iso", "aniso"):
            return [0.0, k, sx, sy, th, 1.0, float(iso), 0.0]
        if kind.startswith("generalized"):
            return [1.0, k, sx, sy, th, self._beta(betag_range), float(iso), 0.0]
        if kind.startswith("plateau"):
            return [2.0, k, sx, sy, th, self._beta(betap_range), float(iso), 0.0]
        msg = f"unknown kernel type {kind!r}"
        raise ValueError(msg)

    def mixed(self, kernel_list, kernel_prob, k=21, sigma_x_range=(0.6, 5), sigma_y_range=(0.6, 5),
              rotation_range=(-math.pi, math.pi), betag_range=(0.5, 8), betap_range=(0.5, 8),
              noise_range=None) -> np.ndarray:
        """random_mixed_kernels (degradations.py:410-475)."""
        assert k % 2 == 1, "Kernel size must be an odd number."
        kind = self.random.choices(kernel_list, kernel_prob)[0]
        iso = not kind.endswith("aniso")
        sx, sy, th = self._shape_params(sigma_x_range, sigma_y_range, rotation_range, iso)
        if kind in ("iso", "aniso"):
            return self._noise(bivariate_gaussian(k, sx, sy, th, iso), noise_range)
        if kind.startswith("generalized"):
            beta = self._beta(betag_range)
            return self._noise(bivariate_generalized_gaussian(k, sx, sy, th, beta, iso), noise_range)
        if kind.startswith("plateau"):
            beta = self._beta(betap_range)
            return self._noise(bivariate_plateau(k, sx, sy, th, beta, iso), None)  # no noise (ref :455-472)
        msg = f"unknown kernel type {kind!r}"
        raise ValueError(msg)

    def _blur_or_sinc(self, opt, suffix: str) -> np.ndarray:
        k = self.random.choice(KERNEL_SIZES)
        if self.rng.uniform() < opt.get(f"sinc_prob{suffix}"):
            lo = np.pi / 3 if k < 13 else np.pi / 5
            kernel = circular_lowpass_kernel(self.rng.uniform(lo, np.pi), k, pad_to=0)
        else:
            sig = opt.get(f"blur_sigma{suffix}")
            kernel = self.mixed(opt.get(f"kernel_list{suffix}"), opt.get(f"kernel_prob{suffix}"), k,
                                sig, sig, [-math.pi, math.pi], opt.get(f"betag_range{suffix}"),
                                opt.get(f"betap_range{suffix}"), noise_range=None)
        p = (21 - k) // 2
        return np.pad(kernel, ((p, p), (p, p)))

    def otf_kernels(self, opt: dict) -> dict[str, np.ndarray]:
        """kernel1, kernel2, sinc_kernel of one sample, float32 21x21 (otf_dataset.py:189-246)."""
        k1 = self._blur_or_sinc(opt, "")
        k2 = self._blur_or_sinc(opt, "2")
        if self.rng.uniform() < opt.get("final_sinc_prob"):
            k = self.random.choice(KERNEL_SIZES)
            sinc = circular_lowpass_kernel(self.rng.uniform(np.pi / 3, np.pi), k, pad_to=21)
        else:
            sinc = np.zeros((21, 21))
            sinc[10, 10] = 1
        return {"kernel1": k1.astype(np.float32), "kernel2": k2.astype(np.float32),
                "sinc_kernel": sinc.astype(np.float32)}

    # ---- the same draws as parameter rows for the device generator (`neosr_blur_kernels`)
    def _blur_or_sinc_params(self, opt, suffix: str) -> list[float]:
        k = self.random.choice(KERNEL_SIZES)
        if self.rng.uniform() < opt.get(f"sinc_prob{suffix}"):
            lo = np.pi / 3 if k < 13 else np.pi / 5
            return [3.0, k, 0.0, 0.0, 0.0, self.rng.uniform(lo, np.pi), 1.0, 0.0]
        sig = opt.get(f"blur_sigma{suffix}")
        return self.mixed_params(opt.get(f"kernel_list{suffix}"), opt.get(f"kernel_prob{suffix}"), k, sig, sig,
                                 [-math.pi, math.pi], opt.get(f"betag_range{suffix}"), opt.get(f"betap_range{suffix}"))

    def otf_kernel_params(self, opt: dict) -> list[list[float]]:
        """kernel1, kernel2, sinc_kernel of one sample as three parameter rows, consuming the RNG streams exactly like
        `otf_kernels` (otf_dataset.py:189-246)"""
        k1 = self._blur_or_sinc_params(opt, "")
        k2 = self._blur_or_sinc_params(opt, "2")
        if self.rng.uniform() < opt.get("final_sinc_prob"):
            k = self.random.choice(KERNEL_SIZES)
            sinc = [3.0, k, 0.0, 0.0, 0.0, self.rng.uniform(np.pi / 3, np.pi), 1.0, 0.0]
        else:
            sinc = [4.0, 21, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]
        return [k1, k2, sinc]

    def otf_kernel_batch_device(self, opt: dict, batch: int, device="cuda"):
        """`otf_kernel_batch` with the 3 x batch kernels evaluated on the device in one launch: only the 24 floats of
        parameters per sample cross PCIe instead of three 21 x 21 kernels, and the float64 grid arithmetic leaves the
        host (the reference does it in its DataLoader workers)."""
        import torch

        from neosr_amd.hip import degrade as D

        rows = [self.otf_kernel_params(opt) for _ in range(batch)]
        table = torch.tensor(rows, dtype=torch.float64).reshape(batch * 3, 8).to(device, non_blocking=True)
        ks = D.blur_kernels(table).view(batch, 3, 21, 21)
        return {"kernel1": ks[:, 0].contiguous(), "kernel2": ks[:, 1].contiguous(), "sinc_kernel": ks[:, 2].contiguous()}

    def otf_kernel_batch(self, opt: dict, batch: int):
        import torch

        ks = [self.otf_kernels(opt) for _ in range(batch)]
        return {n: torch.from_numpy(np.stack([k[n] for k in ks])) for n in ("kernel1", "kernel2", "sinc_kernel")}
