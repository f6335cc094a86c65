"""``image`` — the SISR training model (drop-in for neosr/models/image.py:27-662 on the hot path).

`feed_data(batch)` / `optimize_parameters(it)` / `get_current_log()` keep the reference contract
and the same `log_dict` keys; the work is re-staged for one MI355X per process:

  closure:   net_g forward+backward = two calls into libneosr_amd (whole-network HIP plans);
             losses on HIP reduction kernels; gradients land in one flat arena
  step:      (RCCL all-reduce of the flat arena) -> fused clip + AdamW + EMA kernel
  logging:   loss scalars stay on the device until `get_current_log()`

Built besides the plain step: gradient accumulation, `train.sam` (fsam), `train.eco`, `match_lq_colors`, the GAN / perceptual /
mssim / consistency loss stack, every optimizer of the factory.  `use_amp` / `bfloat16` / `fast_matmul` select the one
reduced-precision tier of this path (models/base.py:precision_options).  What is NOT built raises `NotImplementedError`
instead of silently doing something else: `wavelet_guided`, the dists / ldl / ff / gw losses.
"""

from __future__ import annotations

import os

import math

import sys
from collections import OrderedDict
from copy import deepcopy
from typing import Any

import torch
from torch import Tensor, nn

from neosr_amd import _C, optimizers
from neosr_amd.archs import build_network
from neosr_amd.data.augmentations import apply_augment, resize_aa
from neosr_amd.data.draws import LiveDraws
from neosr_amd.hip import transformer as _tr
from neosr_amd.hip import nets as _nets
from neosr_amd.hip.nets import arena_layout, flat_grad_of, flatten_parameters_, pack_grads
from neosr_amd.losses import build_loss
from neosr_amd.losses.consistency_loss import _Clamp
from neosr_amd.models.base import base
from neosr_amd.models.tiling import tiled_inference
from neosr_amd.utils.grad_sync import GradSync
from neosr_amd.utils.misc import get_root_logger, tc
from neosr_amd.utils.registry import MODEL_REGISTRY


class EMAModel(nn.Module):
    """`AveragedModel(net, multi_avg_fn=get_ema_multi_avg_fn(decay))` (image.py:82-86) with the
    shadow weights in a flat arena so the update rides inside the optimizer kernel.
    State-dict layout is AveragedModel's: `n_averaged` + `module.*`."""

    def __init__(self, model: nn.Module, decay: float) -> None:
        super().__init__()
        self.module = deepcopy(model)
        self.module._neosr_arena = None  # noqa: SLF001  (deepcopy re-homed the params)
        self.decay = decay
        self.register_buffer("n_averaged", torch.tensor(0, dtype=torch.long, device="cpu"))
        self._count = 0
        for p in self.module.parameters():
            p.requires_grad_(False)

    def arena(self) -> Tensor:
        return flatten_parameters_(self.module)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def mark_updated(self) -> None:
        self._count += 1
        self.n_averaged.fill_(self._count)

    @property
    def first(self) -> bool:
        return self._count == 0


_UNSUPPORTED_TRAIN_FLAGS = ("wavelet_guided",)
_UNSUPPORTED_LOSSES = ("dists_opt", "ldl_opt", "ff_opt", "gw_opt")


@MODEL_REGISTRY.register()
class image(base):
    """Single-Image Super-Resolution model."""

    def __init__(self, opt: dict[str, Any]) -> None:
        super().__init__(opt)
        self.net_g = build_network(opt["network_g"])
        self.net_g = self.model_to_device(self.net_g)
        self.net_d = self.opt.get("network_d", None)
        if self.net_d is not None:
            self.net_d = build_network(self.opt["network_d"])
            self.net_d = self.model_to_device(self.net_d)
        load_path = self.opt["path"].get("pretrain_network_g", None)
        if load_path is not None:
            self.load_network(self.net_g, load_path, self.opt["path"].get("param_key_g"),
                              self.opt["path"].get("strict_load_g", True))
            flatten_parameters_(self.net_g)
        load_path = self.opt["path"].get("pretrain_network_d", None)
        if load_path is not None and self.net_d is not None:
            self.load_network(self.net_d, load_path, self.opt["path"].get("param_key_d"),
                              self.opt["path"].get("strict_load_d", True))
            flatten_parameters_(self.net_d)
        if self.is_train:
            self.init_training_settings()

    # ------------------------------------------------------------------------------------
    def init_training_settings(self) -> None:
        train_opt = self.opt["train"]
        logger = get_root_logger()
        for flag in _UNSUPPORTED_TRAIN_FLAGS:
            if train_opt.get(flag):
                raise NotImplementedError(f"train.{flag} is outside the accelerated hot path")
        for key in _UNSUPPORTED_LOSSES:
            if train_opt.get(key):
                raise NotImplementedError(f"train.{key}: no HIP implementation yet (next rows of SURVEY §8)")

        # Empirical Centroid-oriented Optimization (image.py:136-142,236-238)
        self.eco = train_opt.get("eco", False)
        self.eco_schedule = train_opt.get("eco_schedule", "sigmoid")
        self.eco_iters = train_opt.get("eco_iters", 80000)
        self.eco_init = train_opt.get("eco_init", 15000)
        self.pretrain = (self.opt.get("path") or {}).get("pretrain_network_g", None)
        if self.eco:
            logger.info("Using Empirical Centroid-oriented Optimization (ECO).")

        self.ema = train_opt.get("ema", -1)
        if self.ema > 0:
            self.net_g_ema = EMAModel(self.net_g, self.ema)
            self.net_g_ema.arena()
            logger.info("Using exponential-moving average.")
        # sharpness-aware minimization (image.py:90-91,232-257)
        self.sam = train_opt.get("sam", None)
        self.sam_init = train_opt.get("sam_init", -1)
        if self.sam is not None:
            logger.info("Sharpness-Aware Minimization enabled.")
            if (self.opt["datasets"]["train"].get("accumulate", 1) or 1) > 1:
                raise NotImplementedError(f"{tc.red}SAM can't be used with gradient accumulation yet.{tc.end}")

        self.setup_optimizers()
        self.setup_schedulers()
        self.net_g.train()
        self.graph_generator()  # `compile = true`
        if self.sf_optim_g:
            self.optimizer_g.train()  # image.py:99-105
        if self.net_d is not None:
            self.net_d.train()
            if self.sf_optim_d:
                self.optimizer_d.train()

        self.scale = self.opt["scale"]
        ds = self.opt["datasets"]["train"]
        self.patch_size = ds.get("patch_size")
        self.aug = ds.get("augmentation", None)
        self.aug_prob = ds.get("aug_prob", None)
        if self.aug is not None and self.patch_size % 4 != 0:  # image.py:275-278
            raise ValueError(f"{tc.red}The patch_size value must be a multiple of 4 while using augmentations.{tc.end}")
        if not hasattr(self, "draws"):  # python `random` / numpy Generator(manual_seed) / torch, as the reference
            self.draws = LiveDraws(self.opt.get("manual_seed"), self.device)
        self.use_amp = False
        self.total_iter = train_opt.get("total_iter", 200000)
        self.n_accumulated = 0
        self._sam_now = False
        self.accum_iters = ds.get("accumulate", 1) or 1

        def crit(key):
            return build_loss(train_opt[key]).to(self.device) if train_opt.get(key) else None

        # data-parallel exchange (base.py:140-146 wraps the nets in DDP): one GradSync per network; the RRDB plan
        # reduces gradient buckets during its backward when the step follows directly (no accumulation, no SAM)
        # discriminator phase beside the generator's backward on a second stream (see `_closure`): on by default for the
        # layer-composed generators; the RRDB / compact plans are left alone — a chain launch of the RRDB trunk needs every
        # CU for its co-resident workgroups.  NEOSR_AMD_D_OVERLAP=0|1 forces it off / on (A/B runs).
        from neosr_amd.archs.arch_util import HipNet as _HipNet

        env_ov = os.environ.get("NEOSR_AMD_D_OVERLAP", "")
        self._d_overlap = env_ov == "1" or (env_ov != "0" and not isinstance(self.net_g, _HipNet))
        self._d_stream = None
        self._vgg_prefetch = os.environ.get("NEOSR_AMD_VGG_PREFETCH", "1") != "0"
        self._sync_g = self._sync_d = None
        if self.opt["dist"]:
            self._sync_g = GradSync()
            self.net_g._neosr_grad_sync = self._sync_g  # noqa: SLF001
            if not getattr(self.net_g, "plan_sends_grad_buckets", False) and os.environ.get("NEOSR_AMD_DDP_HOOKS", "1") != "0":
                # layer-composed generator (SwinIR, HAT, compact): DDP-style buckets driven by gradient hooks
                self._sync_g.attach(list(self.net_g.parameters()))
                _tr.GRADS_READY = self._sync_g.grads_ready
                _tr.GRAD_SLOT = self._sync_g.grad_slot   # large gradients are written straight into the exchange arena
            if self.net_d is not None:
                self._sync_d = GradSync()

        self.cri_pix = crit("pixel_opt")
        self.cri_mssim = crit("mssim_opt")
        self.cri_consistency = crit("consistency_opt")
        self.cri_perceptual = crit("perceptual_opt")
        self.cri_gan = crit("gan_opt")
        self.gradclip = train_opt.get("grad_clip", True)
        if self.opt["dist"]:
            # frozen networks inside the losses (VGG19) sit outside DDP in the reference and hold the same pretrained
            # weights on every rank; the offline fallback draws them from the per-rank seeded RNG -> rank 0's
            import torch.distributed as dist
            for cri in (self.cri_pix, self.cri_mssim, self.cri_consistency, self.cri_perceptual, self.cri_gan):
                if cri is not None:
                    for t in list(cri.parameters()) + list(cri.buffers()):
                        if t.is_cuda:
                            dist.broadcast(t.data, src=0)

        optim_d = train_opt.get("optim_d", None)
        if self.cri_pix is None and self.cri_mssim is None and self.cri_perceptual is None:
            logger.error(f"{tc.red}Both pixel/mssim and perceptual losses are None. "
                         f"Please enable at least one.{tc.end}")
            sys.exit(1)
        if self.net_d is None and optim_d is not None:
            logger.error(f"{tc.red}Please set a discriminator in network_d or disable optim_d.{tc.end}")
            sys.exit(1)
        if self.net_d is not None and optim_d is None:
            logger.error(f"{tc.red}Please set an optimizer for the discriminator or disable network_d.{tc.end}")
            sys.exit(1)
        if self.net_d is not None and self.cri_gan is None:
            logger.error(f"{tc.red}Discriminator needs GAN to be enabled.{tc.end}")
            sys.exit(1)
        if self.net_d is None and self.cri_gan is not None:
            logger.error(f"{tc.red}GAN requires a discriminator to be set.{tc.end}")
            sys.exit(1)

    def setup_optimizers(self) -> None:
        train_opt = self.opt["train"]
        logger = get_root_logger()
        optim_params = []
        for k, v in self.net_g.named_parameters():
            if v.requires_grad:
                optim_params.append(v)
            else:
                logger.warning(f"Params {k} will not be optimized.")
        sf_types = {"AdamW_SF", "adamw_sf", "adan_sf", "Adan_SF"}
        og = dict(train_opt["optim_g"])
        optim_type = og.pop("type")
        if optim_type in sf_types and "schedule_free" not in og:  # image.py:307-314
            logger.error(f"{tc.red}The option 'schedule_free' must be in the config file.{tc.end}")
            sys.exit(1)
        if optim_type not in sf_types:
            og.pop("schedule_free", None)
        self.optimizer_g = self.get_optimizer(optim_type, optim_params, **og)
        self.optimizers.append(self.optimizer_g)
        if self.sam is not None:  # image.py:322-352: a second instance of the base optimizer inside fsam
            bases = {"adamw": optimizers.AdamW, "adan": optimizers.adan, "adamw_sf": optimizers.adamw_sf,
                     "adan_sf": optimizers.adan_sf}
            if optim_type.lower() not in bases:
                logger.error(f"{tc.red}SAM not supported by optimizer {optim_type} yet.{tc.end}")
                sys.exit(1)
            if self.sam not in {"FSAM", "fsam"}:
                logger.error(f"{tc.red}SAM type {self.sam} not supported yet.{tc.end}")
                sys.exit(1)
            self.sam_optimizer_g = optimizers.fsam(optim_params, bases[optim_type.lower()], rho=0.5, sigma=1,
                                                   lmbda=0.9, adaptive=True, **og)
        if self.net_d is not None:
            od = dict(train_opt["optim_d"])
            optim_type = od.pop("type")
            if optim_type in sf_types and "schedule_free" not in od:  # image.py:358-365
                logger.error(f"{tc.red}The option 'schedule_free' must be in the config file.{tc.end}")
                sys.exit(1)
            if optim_type not in sf_types:
                od.pop("schedule_free", None)
            self.optimizer_d = self.get_optimizer(optim_type, list(self.net_d.parameters()), **od)
            self.optimizers.append(self.optimizer_d)

    # ------------------------------------------------------------------------------------
    @torch.no_grad()
    def feed_data(self, data: dict[str, Any]) -> None:
        self.lq = data["lq"].to(self.device, non_blocking=True)
        if "gt" in data:
            self.gt = data["gt"].to(self.device, non_blocking=True)
        # image.py:381-391
        if self.is_train and self.aug is not None and not (len(self.aug) == 1 and "none" in self.aug):
            self.gt, self.lq = apply_augment(self.gt, self.lq, self.draws, scale=self.scale, augs=self.aug,
                                             prob=self.aug_prob)

    def _sync_grads(self, optimizer, sync: GradSync | None) -> None:
        """data-parallel exchange + clip request for one network (after its backward).  The all-reduce is only
        ENQUEUED here (behind the buckets the RRDB plan already sent during backward); `optimize_parameters`
        waits for it right before that network's optimizer step, so G's exchange overlaps the discriminator
        phase and D's exchange overlaps G's optimizer step."""
        params = optimizer.param_groups[0]["params"]
        if self.opt["dist"]:
            if not sync.end_backward():   # (hook-driven exchange: every bucket is already on the communication stream)
                flat = flat_grad_of(params)
                if flat is None:
                    flat = pack_grads(params)
                    for p, off in zip(params, arena_layout(params)[0]):
                        p.grad = flat[off : off + p.numel()].view_as(p)
                sync.start(flat)
            if self._sam_now:  # the closure runs again inside fsam.step: finish this exchange first
                sync.finish()
            optimizer.set_grad_scale(1.0 / self.opt["world_size"])
        if self.gradclip and not self._sam_now:  # image.py:533-544,597-609: no clipping under SAM
            optimizer.set_clip(1.0)

    def _d_params(self) -> list:
        """the discriminator's parameters as a cached list (walking the module tree twice per step is host time)"""
        ps = getattr(self, "_d_param_list", None)
        if ps is None:
            ps = self._d_param_list = list(self.net_d.parameters())
        return ps

    def eco_strategy(self, current_iter: int):
        """`train.eco` (image.py:393-418, "Empirical Centroid-oriented Optimization", arXiv 2312.17526): a no-grad
        prediction pulls the target towards what the network already produces — GT centroid (1-a) G(lq) + a gt, LQ centroid
        (1-a) clamp(bicubic-antialias down(G(lq))) + a lq — and the step is taken on the prediction from the LQ centroid;
        a follows a sigmoid (skewed at 0.25 eco_iters) or linear schedule."""
        lib = _C.load()
        if self.eco_schedule == "sigmoid":
            a = 1 / (1 + math.exp(-1 * (10 * (current_iter / self.eco_iters - 0.25))))
        else:
            a = min(current_iter / self.eco_iters, 1.0)
        with torch.no_grad():
            if self._sync_g is not None:
                armed, self._sync_g.armed = self._sync_g.armed, False
            net_output = self.net_g(self.lq).contiguous()
            if self._sync_g is not None:
                self._sync_g.armed = armed
            lq_scaled = resize_aa(net_output, net_output.shape[2] // self.scale, net_output.shape[3] // self.scale,
                                  "bicubic", clamp=True)  # clamp(., 0, 1)
            gt = self.gt.contiguous()
            # p <- p + w (end - p):  (1-a) net_output + a gt   and   (1-a) lq_scaled + a lq
            _C.check(lib.neosr_lerp(net_output.data_ptr(), gt.data_ptr(), net_output.numel(), float(a), _C.stream_ptr()),
                     "neosr_lerp")
            lq = self.lq.contiguous()
            _C.check(lib.neosr_lerp(lq_scaled.data_ptr(), lq.data_ptr(), lq_scaled.numel(), float(a), _C.stream_ptr()),
                     "neosr_lerp")
        return self.net_g(lq_scaled), net_output

    def closure(self, current_iter: int):
        """image.py:427-625: G forward, weighted losses, G backward; then D real/fake forward+backward."""
        # a plan generator may write `.grad` itself (hip/nets.py: direct_param_grads) when this backward is the one stepped
        # (not on data-parallel runs: the hook-driven gradient exchange listens to autograd's accumulation)
        with _nets.direct_param_grads(self.accum_iters == 1 and not self._sam_now and not self.opt["dist"]):
            return self._closure(current_iter)

    def _closure(self, current_iter: int):  # noqa: ARG002
        if self.net_d is not None:
            for p in self._d_params():
                p.requires_grad = False

        self.n_accumulated += 1
        if self.n_accumulated >= self.accum_iters:
            self.n_accumulated = 0
        step_now = self.n_accumulated % self.accum_iters == 0

        if self._sync_g is not None:  # buckets may go during backward only if this backward is the one stepped
            self._sync_g.armed = step_now and self.accum_iters == 1 and not self._sam_now
        eco_now = self.eco and current_iter <= self.eco_iters and not (current_iter < self.eco_init and self.pretrain is None)
        if (self._d_overlap and self._vgg_prefetch and self.cri_perceptual and not eco_now
                and hasattr(self.cri_perceptual, "prefetch_gt")):
            # the perceptual loss's target features beside the generator's forward (second stream; see `d_phase` below)
            main = torch.cuda.current_stream(self.device)
            if self._d_stream is None:
                self._d_stream = torch.cuda.Stream(self.device)
                self._d_fork, self._d_join = torch.cuda.Event(), torch.cuda.Event()
            self._d_fork.record(main)
            self._d_stream.wait_event(self._d_fork)
            self.cri_perceptual.prefetch_gt(self.gt, self._d_stream)
        if eco_now:
            self.output, self.gt = self.eco_strategy(current_iter)  # image.py:441-446
        else:
            self.output = self.net_g(self.lq)

        # (the reference starts from zeros(1) and adds every term, image.py:448: 0 + x = x exactly for these non-negative
        # terms, so the first term is taken as it is — two launches fewer per step, and none for `/ 1`)
        l_g_total = None
        loss_dict = OrderedDict()

        def _acc(total, term):
            return term if total is None else total + term

        if self.cri_pix:
            l_g_pix = self.cri_pix(self.output, self.gt)
            l_g_total = _acc(l_g_total, l_g_pix)
            loss_dict["l_g_pix"] = l_g_pix
        if self.cri_mssim:  # image.py:478-481
            l_g_mssim = self.cri_mssim(self.output, self.gt)
            l_g_total = _acc(l_g_total, l_g_mssim)
            loss_dict["l_g_mssim"] = l_g_mssim
        if self.cri_consistency:  # image.py:451-462,483-489
            target = self.gt
            if self.opt["train"].get("match_lq_colors", False):
                with torch.no_grad():  # clamp(bicubic-antialias upsample of the LQ, 1/255, 1)
                    up = resize_aa(self.lq, self.lq.shape[2] * self.scale, self.lq.shape[3] * self.scale, "bicubic",
                                   clamp=False)
                    target = _Clamp.apply(up, 1 / 255, 1.0)
            l_g_consistency = self.cri_consistency(self.output, target)
            l_g_total = _acc(l_g_total, l_g_consistency)
            loss_dict["l_g_consistency"] = l_g_consistency
        if self.cri_perceptual:
            l_g_percep = self.cri_perceptual(self.output, self.gt)
            l_g_total = _acc(l_g_total, l_g_percep)
            loss_dict["l_g_percep"] = l_g_percep
        if self.cri_gan:
            self.broadcast_buffers(self.net_d)
            fake_g_pred = self.net_d(self.output)
            l_g_gan = self.cri_gan(fake_g_pred, target_is_real=True, is_disc=False)
            l_g_total = _acc(l_g_total, l_g_gan)
            loss_dict["l_g_gan"] = l_g_gan
        if l_g_total is None:
            l_g_total = torch.zeros(1, device=self.device)
        loss_dict["l_g_total"] = l_g_total
        if self.accum_iters != 1:
            l_g_total = l_g_total / self.accum_iters
        # ---- discriminator phase (image.py:546-609): both forwards first, then both backwards (image.py:559,574,593-594).
        # It needs the generator's OUTPUT and the discriminator's weights, not the generator's gradients, so with
        # `self._d_overlap` it is enqueued on a second stream BEFORE the generator's backward and runs beside it: a
        # transformer generator's backward is a chain of dependent launches that leaves CUs idle at every boundary, the
        # U-Net's convolutions fill them.  Same kernels on the same operands in the same per-network order (the
        # spectral-norm vectors advance G-phase forward -> real -> fake as in the reference): bit-identical results.
        def d_phase() -> None:
            for p in self._d_params():
                p.requires_grad = True
            if self.cri_gan:
                self.broadcast_buffers(self.net_d)
                real_d_pred = self.net_d(self.gt)
                l_d_real = self.cri_gan(real_d_pred, target_is_real=True, is_disc=True) / self.accum_iters
                loss_dict["l_d_real"] = l_d_real
                loss_dict["out_d_real"] = self.cri_gan.last_mean
                self.broadcast_buffers(self.net_d)
                fake_d_pred = self.net_d(self.output.detach())
                l_d_fake = self.cri_gan(fake_d_pred, target_is_real=False, is_disc=True) / self.accum_iters
                loss_dict["l_d_fake"] = l_d_fake
                loss_dict["out_d_fake"] = self.cri_gan.last_mean
                loss_dict["l_d_total"] = (l_d_real + l_d_fake) / 2
                with _tr.deferred_reductions():
                    l_d_real.backward()
                    l_d_fake.backward()

        overlap = (self._d_overlap and self.net_d is not None and bool(self.cri_gan) and self.accum_iters == 1
                   and not self._sam_now)
        if overlap:
            main = torch.cuda.current_stream(self.device)
            if self._d_stream is None:
                self._d_stream = torch.cuda.Stream(self.device)
                self._d_fork, self._d_join = torch.cuda.Event(), torch.cuda.Event()
            self._d_fork.record(main)   # behind the G-phase forward through net_d (its spectral-norm update comes first)
            with torch.cuda.stream(self._d_stream):
                self._d_stream.wait_event(self._d_fork)
                d_phase()
                self._d_join.record(self._d_stream)

        if self._sync_g is not None:
            self._sync_g.arm_backward()   # hook-driven buckets leave from inside this backward (no-op for the RRDB plan)
        with _tr.deferred_reductions():   # parameter-gradient column sums batched at the end of the pass (opt-in)
            l_g_total.backward()
        if step_now:
            self._sync_grads(self.sam_optimizer_g if self._sam_now else self.optimizer_g, self._sync_g)

        if self.net_d is not None:
            if overlap:
                torch.cuda.current_stream(self.device).wait_event(self._d_join)
            else:
                d_phase()
            if step_now:
                self._sync_grads(self.optimizer_d, self._sync_d)

        self.reduce_loss_dict(loss_dict)
        return l_g_total

    def optimize_parameters(self, current_iter: int) -> None:
        _tr.reset_deferred()  # (reductions queued by a backward pass that raised)
        self._iters_seen += 1   # (base.chain_slow_grace_iters counts these)
        self.n_accumulated += 1
        if self.n_accumulated >= self.accum_iters:
            self.n_accumulated = 0
        self._sam_now = self.sam is not None and current_iter >= self.sam_init
        self.closure(current_iter)
        if self.n_accumulated % self.accum_iters == 0:
            opt_g = self.sam_optimizer_g if self._sam_now else self.optimizer_g
            if self.ema > 0:
                opt_g.set_ema(self.net_g_ema.arena(), self.ema, self.net_g_ema.first)
            if self._sync_g is not None:
                self._sync_g.finish()
            if self._sam_now:  # image.py:639-640: first_step, closure at w + e(w), second_step
                self.sam_optimizer_g.step(self.closure, current_iter)
            else:
                self.optimizer_g.step()
            if self.net_d is not None:
                if self._sync_d is not None:
                    self._sync_d.finish()
                self.optimizer_d.step()
            opt_g.zero_grad(set_to_none=True)
            if self.net_d is not None:
                self.optimizer_d.zero_grad(set_to_none=True)
            if self.ema > 0:
                self.net_g_ema.mark_updated()

    # ------------------------------------------------------------------------------------
    def _eval_net(self):
        """which weights `test()` runs (image.py:672-680,741-760): the EMA copy while training with EMA"""
        if getattr(self, "ema", -1) > 0 and hasattr(self, "net_g_ema"):  # `is_train` is False inside validation
            return self.net_g_ema
        return self.net_g

    def test(self) -> None:
        """image.py:664-783: inference on `self.lq`, whole image (`val.tile = -1`) or partitioned into
        (h // tile + 1) x (w // tile + 1) bands with 16-pixel overlaps and mirror padding."""
        tile = self.opt["val"].get("tile", -1)
        scale = self.opt["scale"]
        sf = bool(self.sf_optim_g) and self.is_train
        if sf:
            self.optimizer_g.eval()  # schedule-free: evaluate at the averaged weights
        net = self._eval_net()
        net.eval()
        with torch.inference_mode():
            if tile == -1:
                # (image.py:672-676 leaves `output` untouched when training without EMA; we run net_g)
                self.output = net(self.lq)
            else:
                C = 1 if self.opt.get("color", None) == "y" else None
                self.output = tiled_inference(net, self.lq, tile, scale, C)
        self.net_g.train()
        if sf:
            self.optimizer_g.train()

    def dist_validation(self, dataloader, current_iter: int, tb_logger, save_img: bool = True) -> None:
        if self.opt["rank"] == 0:
            self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def nondist_validation(self, dataloader, current_iter: int, tb_logger, save_img: bool = True) -> None:  # noqa: ARG002
        """image.py:792-922: per validation image `feed_data` -> `test()` (the HIP inference path, whole image or
        partitioned) -> uint8 image -> metrics (`val.metrics`: calculate_psnr / calculate_ssim on the host, as in the
        reference) -> optional PNG; running means, best-so-far record, log line, TensorBoard scalars.  The dataloader is
        the caller's (any iterable of {"lq", "gt"?, "lq_path"} batches of one image with `.dataset.opt`)."""
        from pathlib import Path

        from neosr_amd.metrics import calculate_metric, imwrite_png, tensor2img

        self.is_train = False  # no augmentation during validation
        dataset_name = dataloader.dataset.opt["name"]
        with_metrics = dataloader.dataset.opt.get("type") != "single" and self.opt["val"].get("metrics") is not None
        if with_metrics:
            self._initialize_best_metric_results(dataset_name)
            self.metric_results = dict.fromkeys(self.opt["val"]["metrics"].keys(), 0)
        n = 0
        try:
            for val_data in dataloader:
                n += 1
                img_name = Path(Path(val_data["lq_path"][0]).name).stem
                self.feed_data(val_data)
                self.test()
                visuals = self.get_current_visuals()
                metric_data = {"img": tensor2img(visuals["result"])}
                if "gt" in visuals:
                    metric_data["img2"] = tensor2img(visuals["gt"])
                    del self.gt
                del self.lq
                del self.output
                if self.opt["val"].get("save_img", True):
                    vis = Path(self.opt["path"]["visualization"])
                    suffix = self.opt["val"].get("suffix", None)
                    if self.opt["is_train"]:
                        out = vis / img_name / f"{img_name}_{current_iter}.png"
                    elif suffix is not None:
                        out = vis / dataset_name / f"{img_name}_{suffix}.png"
                    else:
                        out = vis / dataset_name / f'{img_name}_{self.opt["name"]}.png'
                    imwrite_png(metric_data["img"], out)
                if with_metrics:
                    for name, opt_ in self.opt["val"]["metrics"].items():
                        self.metric_results[name] += calculate_metric(metric_data, opt_)
        finally:
            self.is_train = True
        if with_metrics and n:
            for metric in self.metric_results:
                self.metric_results[metric] /= n
                self._update_best_metric_result(dataset_name, metric, self.metric_results[metric], current_iter)
            self._log_validation_metric_values(current_iter, dataset_name, tb_logger)

    def _log_validation_metric_values(self, current_iter: int, dataset_name: str, tb_logger) -> None:
        """image.py:903-922"""
        log_str = f"Validation {dataset_name}\n\n"
        for metric, value in self.metric_results.items():
            log_str += f"\t # {metric}: {value:.4f}"
            if hasattr(self, "best_metric_results"):
                best = self.best_metric_results[dataset_name][metric]
                log_str += f'{tc.light_green}........ Best: {best["val"]:.4f} @ {best["iter"]} iter{tc.end}'
            log_str += "\n"
        get_root_logger().info(log_str)
        if tb_logger:
            for metric, value in self.metric_results.items():
                tb_logger.add_scalar(f"metrics/{dataset_name}/{metric}", value, current_iter)

    def get_current_visuals(self) -> OrderedDict:
        """image.py:924-930"""
        out = OrderedDict()
        out["lq"] = self.lq.detach().cpu()
        out["result"] = self.output.detach().cpu()
        if hasattr(self, "gt"):
            out["gt"] = self.gt.detach().cpu()
        return out

    # ------------------------------------------------------------------------------------
    def save(self, epoch: int, current_iter: int) -> None:
        """image.py:932-942: EMA weights are saved as `net_g` when EMA is on."""
        if self.ema > 0:
            self.save_network(self.net_g_ema, "net_g", current_iter, param_key="params")
        else:
            self.save_network(self.net_g, "net_g", current_iter)
        self.save_training_state(epoch, current_iter)
