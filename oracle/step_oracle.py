"""CPU oracle for ONE whole training iteration of any BASELINE config — TEST INFRASTRUCTURE ONLY.

`ConfigTrainer` restates `image.optimize_parameters` / `image.closure` (neosr/models/image.py:427-662)
and, for `model_type = "otf"`, `otf.feed_data` (neosr/models/otf.py:92-283) on top of the per-op
restatements in this directory, driven by the same nested `opt` dict the reference parses from a TOML:
generator = esrgan / compact / swinir_* / hat_*; losses = L1 (+ VGG19 perceptual + GAN with the U-Net-SN
discriminator); optimizers = adamw / adan_sf for G and D; grad clip; EMA.

Parity status: PINNED piecewise — every building block is held to reference-run fixtures
(tests/test_oracle_*.py), and the composed iteration to the reference-run trajectories
tests/golden/step_cfg*.npz (tests/golden/gen_golden_cfgs.py; tests/test_oracle_cfgs.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from __future__ import annotations

from collections import OrderedDict

import torch

from oracle import degrade_oracle as dorc
from oracle import gan_oracle as gorc
from oracle import hat_oracle as horc
from oracle import neosr_oracle as orc
from oracle import swinir_oracle as sorc


def generator_forward(net_opt: dict, scale: int = 4):
    """forward function `(P, x) -> y` of the registered architecture `network_g.type` with its ctor kwargs."""
    kw = {k: v for k, v in net_opt.items() if k != "type"}
    name = net_opt["type"]
    if name == "esrgan":
        return lambda P, x: orc.rrdbnet_forward(P, x, scale)
    if name == "compact":
        return lambda P, x: orc.compact_forward(P, x, kw.get("upscale", scale), kw.get("act_type", "prelu"))
    if name in sorc.VARIANTS:
        cfg = dict(sorc.VARIANTS[name], upscale=scale)
        cfg.update(kw)
        cfg.setdefault("drop_path_rate", 0.0)  # DropPath masks are replayed explicitly by the tests that use them
        return lambda P, x: sorc.swinir_forward(P, x, **cfg)
    if name in horc.VARIANTS:
        cfg = dict(horc.VARIANTS[name], upscale=scale)
        cfg.update(kw)
        cfg.setdefault("drop_path_rate", 0.0)
        return lambda P, x: horc.hat_forward(P, x, **cfg)
    msg = f"step_oracle: no restatement of network_g.type = {name!r}"
    raise ValueError(msg)


class _AdamW:
    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, **_):
        self.p, self.lr, self.betas, self.eps, self.wd = params, lr, tuple(betas), eps, weight_decay
        self.m = [torch.zeros_like(v) for v in params]
        self.v = [torch.zeros_like(v) for v in params]
        self.t = 0

    def step(self, grads):
        self.t += 1
        with torch.no_grad():
            orc.adamw_step(self.p, grads, self.m, self.v, self.t, self.lr, self.betas, self.eps, self.wd)


def make_optimizer(params, o: dict):
    o = dict(o)
    kind = o.pop("type").lower()
    if kind == "adamw":
        o.pop("schedule_free", None)
        return _AdamW(params, **o)
    if kind == "adan_sf":
        return orc.AdanSF(params, o["lr"], tuple(o.get("betas", (0.98, 0.92, 0.99))), o.get("eps", 1e-8),
                          o.get("weight_decay", 0.02), o.get("warmup_steps", 0),
                          schedule_free=o.get("schedule_free", True))
    msg = f"step_oracle: optimizer {kind!r} not restated here"
    raise ValueError(msg)


class ConfigTrainer:
    """One rank of the reference's training loop for an `opt` dict: `feed_data(batch)` then
    `optimize_parameters()`; `log` holds the reference's `log_dict` keys."""

    def __init__(self, opt: dict, g_params, d_params=None, vgg_params=None, draws=None) -> None:
        tr = opt["train"]
        self.opt, self.scale = opt, opt.get("scale", 4)
        self.fwd = generator_forward(opt["network_g"], self.scale)
        # state-dict entries: floating tensors are the trainable parameters; integer buffers (index tables) ride along
        self.G = OrderedDict((k, v.clone().requires_grad_(v.is_floating_point() and "attn_mask" not in k))
                             for k, v in g_params.items())
        self.g_train = [v for v in self.G.values() if v.requires_grad]
        self.opt_g = make_optimizer(self.g_train, tr["optim_g"])
        self.ema_decay = tr.get("ema", -1)
        self.ema = [v.detach().clone() for v in self.g_train]
        self.clip = tr.get("grad_clip", True)
        self.pix_w = tr["pixel_opt"].get("loss_weight", 1.0) if tr.get("pixel_opt") else None
        self.per_w = tr["perceptual_opt"].get("loss_weight", 1.0) if tr.get("perceptual_opt") else None
        self.gan_w = tr["gan_opt"].get("loss_weight", 0.1) if tr.get("gan_opt") else None
        self.vggP = vgg_params
        self.D = None
        if d_params is not None:
            self.D = OrderedDict((k, v.clone()) for k, v in d_params.items())
            self.d_train = [k for k in self.D if not k.endswith(("_u", "_v"))]
            for k in self.d_train:
                self.D[k].requires_grad_(True)
            self.opt_d = make_optimizer([self.D[k] for k in self.d_train], tr["optim_d"])
        self.draws = draws
        self.pool = None
        if opt.get("model_type") == "otf":
            ds = opt["datasets"]["train"]
            self.pool = dorc.PairPool(ds.get("queue_size", 180), ds["batch_size"])
            self.dopt = dict(opt.get("degradations") or {})
        self.n = 0
        self.log: dict[str, float] = {}

    def feed_data(self, batch: dict) -> None:
        if self.pool is None:
            self.lq, self.gt = batch["lq"], batch["gt"]
            return
        ps = self.opt["datasets"]["train"]["patch_size"]
        lq, gt = dorc.otf_feed_data(batch["gt"], batch["kernel1"], batch["kernel2"], batch["sinc_kernel"],
                                    self.dopt, self.scale, ps, self.draws)
        self.lq, self.gt = self.pool.step(lq, gt, self.draws)

    def optimize_parameters(self) -> None:
        log = OrderedDict()
        out = self.fwd(self.G, self.lq)
        l_g_total = torch.zeros(1)
        if self.pix_w is not None:
            l = orc.l1_loss(out, self.gt, self.pix_w)
            l_g_total = l_g_total + l
            log["l_g_pix"] = float(l.detach())
        if self.per_w is not None:
            l = gorc.perceptual_loss(self.vggP, out, self.gt, self.per_w)
            l_g_total = l_g_total + l
            log["l_g_percep"] = float(l.detach())
        if self.gan_w is not None:
            frozen = OrderedDict((k, v.detach()) if k in self.d_train else (k, v) for k, v in self.D.items())
            l = gorc.gan_loss(gorc.unet_forward(frozen, out, True), True, False, self.gan_w)
            l_g_total = l_g_total + l
            log["l_g_gan"] = float(l.detach())
        log["l_g_total"] = float(l_g_total.detach())
        gp = self.g_train
        g_grads = [g.clone() for g in torch.autograd.grad(l_g_total.sum(), gp)]
        if self.clip:
            orc.clip_grad_norm_(g_grads, 1.0)
        d_grads = None
        if self.D is not None:
            real = gorc.unet_forward(self.D, self.gt, True)
            l_real = gorc.gan_loss(real, True, True)
            fake = gorc.unet_forward(self.D, out.detach(), True)
            l_fake = gorc.gan_loss(fake, False, True)
            dp = [self.D[k] for k in self.d_train]
            d_grads = [a + b for a, b in zip(torch.autograd.grad(l_real, dp, retain_graph=True),
                                             torch.autograd.grad(l_fake, dp))]
            if self.clip:
                orc.clip_grad_norm_(d_grads, 1.0)
            log.update(l_d_real=float(l_real.detach()), out_d_real=float(real.detach().mean()), l_d_fake=float(l_fake.detach()),
                       out_d_fake=float(fake.detach().mean()), l_d_total=float((l_real + l_fake).detach() / 2))
        self.n += 1
        self.opt_g.step(g_grads)
        if d_grads is not None:
            self.opt_d.step(d_grads)
        with torch.no_grad():
            if self.ema_decay > 0:
                orc.ema_update(self.ema, gp, self.ema_decay, first=self.n == 1)
        self.output = out.detach()
        self.log = log
