"""CPU oracle for the neosr training hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU fp32 restatement of the reference algorithm for the path named by
BASELINE.json (`feed_data` -> `optimize_parameters`).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; nothing under `neosr_amd/` does.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against
fixtures in tests/golden/*.npz that were produced by importing and running the reference itself
(/root/reference @ 2024-10-16) on CPU in the build container — script: tests/golden/gen_golden.py.
Exception: VGG19 ImageNet weights are not available offline, so anything perceptual is pinned on
structure with seeded random weights only ("parity unpinned" for real VGG weights).

Functional style on purpose: parameters are passed as a ``dict[str, Tensor]`` keyed exactly like
the reference ``state_dict`` so that fixtures, the reference and the HIP product all share names.
"""

from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# index ops (bit-exact)
# --------------------------------------------------------------------------------------------


def pixel_shuffle_np(x: np.ndarray, r: int) -> np.ndarray:
    """nn.PixelShuffle(r): out[b,c,h*r+i,w*r+j] = in[b,c*r*r+i*r+j,h,w]
    (compact_arch.py:74,81; swinir_arch.py:782-783).  Explicit index loops over (i, j)."""
    b, crr, h, w = x.shape
    c = crr // (r * r)
    out = np.empty((b, c, h * r, w * r), dtype=x.dtype)
    for i in range(r):
        for j in range(r):
            out[:, :, i::r, j::r] = x[:, i * r + j :: r * r, :, :][:, :c]
    return out


def pixel_unshuffle_np(x: np.ndarray, s: int) -> np.ndarray:
    """esrgan_arch.py:60-79: (b,c,hh,hw) -> (b,c*s*s,hh/s,hw/s), channel index c*s*s + i*s + j."""
    b, c, hh, hw = x.shape
    h, w = hh // s, hw // s
    out = np.empty((b, c * s * s, h, w), dtype=x.dtype)
    for i in range(s):
        for j in range(s):
            out[:, i * s + j :: s * s][:, :c] = x[:, :, i::s, j::s]
    return out


# --------------------------------------------------------------------------------------------
# generators
# --------------------------------------------------------------------------------------------


def _conv(P, name: str, x: torch.Tensor) -> torch.Tensor:
    return F.conv2d(x, P[f"{name}.weight"], P[f"{name}.bias"], stride=1, padding=1)


def rdb_forward(P, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """ResidualDenseBlock.forward (esrgan_arch.py:109-116)."""
    lrelu = lambda t: F.leaky_relu(t, 0.2)  # noqa: E731
    x1 = lrelu(_conv(P, f"{prefix}.conv1", x))
    x2 = lrelu(_conv(P, f"{prefix}.conv2", torch.cat((x, x1), 1)))
    x3 = lrelu(_conv(P, f"{prefix}.conv3", torch.cat((x, x1, x2), 1)))
    x4 = lrelu(_conv(P, f"{prefix}.conv4", torch.cat((x, x1, x2, x3), 1)))
    x5 = _conv(P, f"{prefix}.conv5", torch.cat((x, x1, x2, x3, x4), 1))
    return x5 * 0.2 + x


def rrdb_forward(P, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """RRDB.forward (esrgan_arch.py:137-142)."""
    out = rdb_forward(P, f"{prefix}.rdb1", x)
    out = rdb_forward(P, f"{prefix}.rdb2", out)
    out = rdb_forward(P, f"{prefix}.rdb3", out)
    return out * 0.2 + x


def rrdbnet_forward(P, x: torch.Tensor, scale: int = 4) -> torch.Tensor:
    """esrgan.forward (esrgan_arch.py:196-214)."""
    if scale in (1, 2):  # esrgan_arch.py:197-200: scale 2 -> unshuffle 2, scale 1 -> unshuffle 4
        s = 2 if scale == 2 else 4
        b, c, hh, hw = x.shape
        feat = x.view(b, c, hh // s, s, hw // s, s).permute(0, 1, 3, 5, 2, 4)
        feat = feat.reshape(b, c * s * s, hh // s, hw // s)
    else:
        feat = x
    num_block = 1 + max(int(k.split(".")[1]) for k in P if k.startswith("body."))
    feat = _conv(P, "conv_first", feat)
    body = feat
    for n in range(num_block):
        body = rrdb_forward(P, f"body.{n}", body)
    feat = feat + _conv(P, "conv_body", body)
    lrelu = lambda t: F.leaky_relu(t, 0.2)  # noqa: E731
    feat = lrelu(_conv(P, "conv_up1", F.interpolate(feat, scale_factor=2, mode="nearest")))
    feat = lrelu(_conv(P, "conv_up2", F.interpolate(feat, scale_factor=2, mode="nearest")))
    return _conv(P, "conv_last", lrelu(_conv(P, "conv_hr", feat)))


def compact_forward(P, x: torch.Tensor, upscale: int = 4, act_type: str = "prelu") -> torch.Tensor:
    """compact.forward (compact_arch.py:76-85): conv/act chain, PixelShuffle, + nearest(x)."""
    conv_ids = sorted({int(k.split(".")[1]) for k in P if k.startswith("body.") and P[k].dim() == 4})
    out = x
    for n, i in enumerate(conv_ids):
        out = F.conv2d(out, P[f"body.{i}.weight"], P[f"body.{i}.bias"], stride=1, padding=1)
        if n == len(conv_ids) - 1:
            break
        if act_type == "prelu":
            out = F.prelu(out, P[f"body.{i + 1}.weight"])
        elif act_type == "relu":
            out = F.relu(out)
        else:
            out = F.leaky_relu(out, 0.1)
    out = F.pixel_shuffle(out, upscale)
    return out + F.interpolate(x, scale_factor=upscale, mode="nearest")


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------


def l1_loss(pred: torch.Tensor, target: torch.Tensor, loss_weight: float = 1.0) -> torch.Tensor:
    """L1Loss.forward, reduction='mean' (basic_loss.py:10-11,44-53)."""
    return loss_weight * (pred - target).abs().mean()


# --------------------------------------------------------------------------------------------
# optimizer step pieces
# --------------------------------------------------------------------------------------------


def clip_grad_norm_(grads: list[torch.Tensor], max_norm: float = 1.0) -> float:
    """torch.nn.utils.clip_grad_norm_(…, max_norm, error_if_nonfinite=False) as called at
    image.py:533-544: total = || [||g_i||_2] ||_2 ; g *= min(1, max_norm / (total + 1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adamw_step(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.999),
               eps: float = 1e-8, weight_decay: float = 0.01) -> None:
    """torch.optim.AdamW single-tensor update (what base.get_optimizer("adamw") steps,
    base.py:151-172): decoupled decay, lerp first moment, bias-corrected step."""
    b1, b2 = betas
    bc1 = 1 - b1**step
    bc2 = 1 - b2**step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        p.mul_(1 - lr * weight_decay)
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_update(ema, params, decay: float, first: bool) -> None:
    """AveragedModel.update_parameters with get_ema_multi_avg_fn(decay) (image.py:82-86,661-662):
    first call copies, afterwards ema.lerp_(p, 1 - decay)."""
    for e, p in zip(ema, params):
        if first:
            e.copy_(p)
        else:
            e.lerp_(p, 1 - decay)


class ImageTrainer:
    """`image.feed_data` + `image.optimize_parameters` (image.py:374-391,427-662) for the subset on
    the benchmarked path: generator only, L1 pixel loss, AdamW, grad clip 1.0, EMA, accumulate=1."""

    def __init__(self, forward_fn, params: "OrderedDict[str, torch.Tensor]", lr: float,
                 betas=(0.9, 0.999), weight_decay: float = 0.01, eps: float = 1e-8,
                 ema: float = 0.999, grad_clip: bool = True, loss_weight: float = 1.0) -> None:
        self.forward_fn = forward_fn
        self.P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
        self.names = list(self.P)
        self.m = [torch.zeros_like(v) for v in self.P.values()]
        self.v = [torch.zeros_like(v) for v in self.P.values()]
        self.ema = [v.detach().clone() for v in self.P.values()]
        self.lr, self.betas, self.wd, self.eps = lr, betas, weight_decay, eps
        self.ema_decay, self.grad_clip, self.loss_weight = ema, grad_clip, loss_weight
        self.step = 0
        self.log: dict[str, float] = {}
        self.output = None

    def feed_data(self, lq: torch.Tensor, gt: torch.Tensor) -> None:
        self.lq, self.gt = lq, gt

    def optimize_parameters(self) -> None:
        out = self.forward_fn(self.P, self.lq)
        l_pix = l1_loss(out, self.gt, self.loss_weight)
        l_total = torch.zeros(1) + l_pix
        grads = torch.autograd.grad(l_total.sum(), list(self.P.values()))
        grads = [g.clone() for g in grads]
        if self.grad_clip:
            clip_grad_norm_(grads, 1.0)
        self.step += 1
        with torch.no_grad():
            plist = list(self.P.values())
            adamw_step(plist, grads, self.m, self.v, self.step, self.lr, self.betas, self.eps, self.wd)
            if self.ema_decay > 0:
                ema_update(self.ema, plist, self.ema_decay, first=self.step == 1)
        self.output = out.detach()
        self.log = {"l_g_pix": float(l_pix), "l_g_total": float(l_total)}


# --------------------------------------------------------------------------------------------
# Schedule-Free Adan (neosr/optimizers/adan_sf.py) — per-tensor restatement
# --------------------------------------------------------------------------------------------


class AdanSF:
    """`adan_sf` (adan_sf.py:10-330): `step(grads)` follows `_multi_tensor_adan` op by op on plain
    tensors; `train()` / `eval()` are the y <-> x lerps; group scalars (`step`, `weight_sum`,
    `lr_max`) live on the object.  `max_grad_norm` (the optimizer's own clip) is 0 as in the templates."""

    def __init__(self, params: list[torch.Tensor], lr: float, betas=(0.98, 0.92, 0.99), eps: float = 1e-8,
                 weight_decay: float = 0.02, warmup_steps: int = 0, r: float = 0.0, weight_lr_power: float = 2.0,
                 schedule_free: bool = True) -> None:
        self.p = params
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.warmup_steps, self.r, self.wlp, self.sf = warmup_steps, r, weight_lr_power, schedule_free
        self.step_n, self.weight_sum, self.lr_max, self.train_mode = 0, 0.0, -1.0, True
        self.state: list[dict[str, torch.Tensor]] = [{} for _ in params]

    @torch.no_grad()
    def eval(self) -> None:
        if self.train_mode:
            for p, s in zip(self.p, self.state):
                if "z" in s:
                    p.lerp_(s["z"], 1 - 1 / self.betas[0])
            self.train_mode = False

    @torch.no_grad()
    def train(self) -> None:
        if not self.train_mode:
            for p, s in zip(self.p, self.state):
                if "z" in s:
                    p.lerp_(s["z"], 1 - self.betas[0])
            self.train_mode = True

    @torch.no_grad()
    def step(self, grads: list[torch.Tensor]) -> None:
        b1, b2, b3 = self.betas
        self.step_n += 1
        t = self.step_n
        bc1, bc2, bc3 = 1.0 - b1**t, 1.0 - b2**t, 1.0 - b3**t
        ckp1 = None
        if self.sf:
            sched = t / self.warmup_steps if t < self.warmup_steps else 1.0
            lr_eff = self.lr * sched * math.sqrt(bc3)
            self.lr_max = max(lr_eff, self.lr_max)
            weight = (t**self.r) * (self.lr_max**self.wlp)
            self.weight_sum += weight
            ckp1 = weight / self.weight_sum if self.weight_sum != 0 else 0
            assert self.train_mode, "Not in train mode!"
        for p, g, s in zip(self.p, grads, self.state):
            if not s:
                s.update(exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p),
                         exp_avg_diff=torch.zeros_like(p), z=p.detach().clone())
            if "neg_pre_grad" not in s or t == 1:
                s["neg_pre_grad"] = g.clone().mul_(-1.0)
            m, n, d, z, npg = s["exp_avg"], s["exp_avg_sq"], s["exp_avg_diff"], s["z"], s["neg_pre_grad"]
            npg.add_(g)
            m.mul_(b1).add_(g, alpha=1 - b1)
            d.mul_(b2).add_(npg, alpha=1 - b2)
            npg.mul_(b2).add_(g)
            n.mul_(b3).addcmul_(npg, npg, value=1 - b3)
            denom = n.sqrt().div_(math.sqrt(bc3)).add_(self.eps)
            p.mul_(1 - self.lr * self.wd)
            if self.sf:
                p.lerp_(z, ckp1)
                p.addcdiv_(m, denom, value=-(self.lr * (bc1 * (1 - ckp1))))
                p.addcdiv_(d, denom, value=-(self.lr * (b2 / bc2 * (1 - ckp1))))
                z.sub_(g, alpha=self.lr)
            else:
                p.addcdiv_(m, denom, value=-(self.lr / bc1))
                p.addcdiv_(d, denom, value=-(self.lr * b2 / bc2))
            npg.zero_().add_(g, alpha=-1.0)


class AdanImageTrainer(ImageTrainer):
    """ImageTrainer with `optim_g.type = "adan_sf"` (models/image.py:627-662 with a schedule-free optimizer)."""

    def __init__(self, forward_fn, params, lr: float, betas=(0.98, 0.92, 0.99), weight_decay: float = 0.02,
                 warmup_steps: int = 0, schedule_free: bool = True, ema: float = 0.999, grad_clip: bool = True,
                 loss_weight: float = 1.0) -> None:
        super().__init__(forward_fn, params, lr, ema=ema, grad_clip=grad_clip, loss_weight=loss_weight)
        self.opt = AdanSF(list(self.P.values()), lr, betas, 1e-8, weight_decay, warmup_steps,
                          schedule_free=schedule_free)

    def optimize_parameters(self) -> None:
        out = self.forward_fn(self.P, self.lq)
        l_pix = l1_loss(out, self.gt, self.loss_weight)
        l_total = torch.zeros(1) + l_pix
        grads = [g.clone() for g in torch.autograd.grad(l_total.sum(), list(self.P.values()))]
        if self.grad_clip:
            clip_grad_norm_(grads, 1.0)
        self.step += 1
        self.opt.step(grads)
        with torch.no_grad():
            if self.ema_decay > 0:
                ema_update(self.ema, list(self.P.values()), self.ema_decay, first=self.step == 1)
        self.output = out.detach()
        self.log = {"l_g_pix": float(l_pix.detach()), "l_g_total": float(l_total.detach())}


# --------------------------------------------------------------------------------------------
# Friendly SAM (neosr/optimizers/fsam.py) — functional restatement
# --------------------------------------------------------------------------------------------
class FSAM:
    """`fsam.first_step` / `second_step` (optimizers/fsam.py:36-95) around an AdamW base optimizer that owns
    its state (the reference builds a SECOND optimizer instance for SAM, image.py:322-347)."""

    def __init__(self, params: list[torch.Tensor], lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, rho: float = 0.5, sigma: float = 1.0, lmbda: float = 0.9,
                 adaptive: bool = True) -> None:
        self.params = params
        self.rho, self.sigma, self.lmbda, self.adaptive = rho, sigma, lmbda, adaptive
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.momentum: list[torch.Tensor] | None = None
        self.old_p: list[torch.Tensor] = []
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def first_step(self, grads: list[torch.Tensor]) -> None:
        """fsam.py:36-66: g' = g - sigma * momentum (not on the first call), momentum <- EMA of the raw
        gradient, then climb to w + rho * w^2 * g' / || |w| * g' ||."""
        if self.momentum is None:
            self.momentum = [g.clone() for g in grads]
        else:
            for i, g in enumerate(grads):
                raw = g.clone()
                g.sub_(self.momentum[i] * self.sigma)
                self.momentum[i] = self.momentum[i] * self.lmbda + raw * (1 - self.lmbda)
        norms = [((p.abs() if self.adaptive else 1.0) * g).norm(p=2) for p, g in zip(self.params, grads)]
        scale = self.rho / (torch.norm(torch.stack(norms), p=2) + 1e-12)
        self.old_p = [p.detach().clone() for p in self.params]
        for p, g in zip(self.params, grads):
            p.add_((torch.pow(p, 2) if self.adaptive else 1.0) * g * scale)

    @torch.no_grad()
    def second_step(self, grads: list[torch.Tensor]) -> None:
        """fsam.py:68-79: back to w, then the base optimizer steps with the gradient taken at w + e(w)."""
        for p, o in zip(self.params, self.old_p):
            p.copy_(o)
        self.t += 1
        adamw_step(self.params, grads, self.m, self.v, self.t, self.lr, self.betas, self.eps, self.wd)


class FsamImageTrainer(ImageTrainer):
    """ImageTrainer with `train.sam = "fsam"`, `train.sam_init` (models/image.py:528-544,627-662): from
    iteration `sam_init` on there is no clipping, the closure runs twice and the SAM optimizer's own AdamW
    takes the step; before that the plain path."""

    def __init__(self, forward_fn, params, lr: float, betas=(0.9, 0.999), weight_decay: float = 0.01,
                 sam_init: int = -1, ema: float = 0.999, grad_clip: bool = True, loss_weight: float = 1.0) -> None:
        super().__init__(forward_fn, params, lr, betas, weight_decay, 1e-8, ema, grad_clip, loss_weight)
        self.sam_init = sam_init
        self.sam = FSAM(list(self.P.values()), lr, betas, 1e-8, weight_decay)
        self.iters = 0

    def _closure(self):
        out = self.forward_fn(self.P, self.lq)
        l_pix = l1_loss(out, self.gt, self.loss_weight)
        l_total = torch.zeros(1) + l_pix
        grads = [g.clone() for g in torch.autograd.grad(l_total.sum(), list(self.P.values()))]
        self.output = out.detach()
        self.log = {"l_g_pix": float(l_pix.detach()), "l_g_total": float(l_total.detach())}
        return grads

    def optimize_parameters(self, current_iter: int) -> None:  # type: ignore[override]
        if current_iter < self.sam_init:
            super().optimize_parameters()
        else:
            self.sam.first_step(self._closure())
            self.sam.second_step(self._closure())
            with torch.no_grad():
                if self.ema_decay > 0:
                    ema_update(self.ema, list(self.P.values()), self.ema_decay, first=self.iters == 0)
        self.iters += 1
