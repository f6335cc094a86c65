"""``base`` model: device placement, optimizer/scheduler factories, LR warm-up, loss-dict reduce,
checkpoint I/O.  Mirrors the public surface of neosr/models/base.py:21-526.

MI355X-first differences (behaviour-preserving):
* no DistributedDataParallel wrapper: each rank owns flat parameter/gradient arenas and the model
  issues one RCCL all-reduce per network on the flat gradient (`allreduce_flat_`), with the
  1/world_size folded into the fused optimizer kernel;
* `reduce_loss_dict` implements the *intent* of base.py:498-526 (mean over ranks, visible on
  rank 0) — the reference assigns `None` from `dist.reduce` and crashes under DDP
  — and defers the device->host read of the scalars to `get_current_log()`.
"""

from __future__ import annotations

import os

import sys
import time
from collections import OrderedDict
from copy import deepcopy
from pathlib import Path
from typing import Any

import torch
import torch.distributed as dist
from torch import nn

from neosr_amd import _C, optimizers
from neosr_amd.hip.nets import flatten_parameters_
from neosr_amd.utils.dist_util import master_only
from neosr_amd.utils.misc import get_root_logger, tc


def allreduce_flat_(flat: torch.Tensor, bucket_bytes: int = 64 << 20) -> None:
    """SUM all-reduce of a flat gradient arena in a few large buckets (RCCL over xGMI on ROCm;
    gloo in the CPU tests).  xGMI is point-to-point: few, large messages beat many small ones;
    the division by world_size happens inside the optimizer kernel (`grad_scale`)."""
    n = flat.numel()
    per = max(1, bucket_bytes // 4)
    handles = []
    for lo in range(0, n, per):
        handles.append(dist.all_reduce(flat[lo : min(n, lo + per)], op=dist.ReduceOp.SUM, async_op=True))
    for h in handles:
        h.wait()


def flatten_sn_buffers_(net: nn.Module) -> torch.Tensor | None:
    """Re-home every spectral-norm `weight_u` / `weight_v` buffer of `net` into one flat fp32 arena (views), so
    the per-forward buffer broadcast of the data-parallel path is ONE message.  None when the net has none."""
    bufs = [b for n, b in net.named_buffers() if n.endswith(("weight_u", "weight_v")) and b.dtype == torch.float32]
    if not bufs:
        return None
    arena = torch.empty(sum(b.numel() for b in bufs), device=bufs[0].device, dtype=torch.float32)
    off = 0
    with torch.no_grad():
        for b in bufs:
            view = arena[off: off + b.numel()].view(b.shape)
            view.copy_(b)
            b.data = view
            off += b.numel()
    return arena


class base:
    """Default model."""

    # Iterations (optimize_parameters calls of this process) whose slow-wait marks are ignored: RCCL sets its connections up
    # inside the first collectives and holds CUs for milliseconds while it does.  Option: [train] chain_slow_grace_iters.
    CHAIN_SLOW_GRACE_ITERS = 20

    def __init__(self, opt: dict[str, Any]) -> None:
        self.opt = opt
        self.device = torch.device("cuda")
        self.is_train = opt["is_train"]
        self.optimizers: list[Any] = []
        self.schedulers: list[Any] = []
        self.log_dict: dict[str, Any] = OrderedDict()
        self._log_dev: tuple[list[str], torch.Tensor] | None = None
        self._log_work = None
        self._log_health = False     # the reduced scalars end with the chain launches' two health words
        self._iters_seen = 0         # optimize_parameters calls (image.optimize_parameters counts)
        self._log_iters = 0          # ... at the time the pending scalars were reduced
        self.chain_slow_grace_iters = int((opt.get("train") or {}).get("chain_slow_grace_iters", self.CHAIN_SLOW_GRACE_ITERS))
        self.chain_fallback = False  # all ranks left the chain launches after a slow flag wait (get_current_log)
        self.n_accumulated = 0
        if self.is_train:
            self.sf_optim_g = opt["train"]["optim_g"].get("schedule_free", False)
            self.net_d = opt.get("network_d")
            self.sf_optim_d = None  # the reference leaves it unset here and only reads it when net_d exists
            if self.net_d is not None:
                self.sf_optim_d = opt["train"]["optim_d"].get("schedule_free", False)
        else:
            self.sf_optim_g = None
            self.sf_optim_d = None

    # -- interface stubs (image/otf override) ----------------------------------------------
    def feed_data(self, data) -> None: ...
    def optimize_parameters(self, current_iter: int) -> None: ...
    def get_current_visuals(self): ...
    def save(self, epoch: int, current_iter: int) -> None: ...

    def validation(self, dataloader, current_iter: int, tb_logger, save_img: bool = True) -> None:
        """base.py:54-71: rank 0 validates under a launcher, everyone otherwise."""
        if self.opt["dist"]:
            self.dist_validation(dataloader, current_iter, tb_logger, save_img)
        else:
            self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def _initialize_best_metric_results(self, dataset_name: str) -> None:
        """base.py:87-104"""
        if not hasattr(self, "best_metric_results"):
            self.best_metric_results = {}
        if dataset_name in self.best_metric_results:
            return
        record = {}
        for metric, content in self.opt["val"]["metrics"].items():
            better = content.get("better", "higher")
            record[metric] = {"better": better, "val": float("-inf") if better == "higher" else float("inf"), "iter": -1}
        self.best_metric_results[dataset_name] = record

    def _update_best_metric_result(self, dataset_name, metric: str, val, current_iter: int) -> None:
        """base.py:106-115"""
        rec = self.best_metric_results[dataset_name][metric]
        if (rec["better"] == "higher" and val >= rec["val"]) or (rec["better"] != "higher" and val <= rec["val"]):
            rec["val"], rec["iter"] = val, current_iter

    # -- logging ------------------------------------------------------------------------
    def get_current_log(self, act_on_health: bool = True) -> dict[str, Any]:
        """Materialise the (possibly rank-reduced) loss scalars: the only device->host read of the
        iteration, paid when the caller actually logs.  NaN in the generator loss raises here
        (reference: ValueError at image.py:611-619, checked every iteration).

        `act_on_health=False` is for callers that run on ONE rank only (the `@master_only` checkpoint writers): the
        scalars are read and the NaN check runs, but the chain launches' health words are neither acknowledged nor acted
        on — that happens at the next read EVERY rank makes (both words are sticky on the device until then), so ranks
        never leave the chain launches, or raise, alone.  The return value of `chain_health_ok` tells such a caller
        whether the iterations behind the scalars are valid."""
        if self._log_dev is not None:
            keys, vals = self._log_dev
            if self._log_work is not None:  # the rank reduce of these scalars was only enqueued (reduce_loss_dict)
                self._log_work.wait()
                self._log_work = None
                vals = vals / self.opt["world_size"]   # (an all-reduce: every rank holds the sums)
            host = vals.detach().float().cpu().tolist()
            # The chain launches of the RRDB trunk need all their workgroups resident at once.  Two health words travel with
            # the loss scalars (neosr_conv_chain_health; on data-parallel runs through the same all-reduce, so EVERY rank sees
            # their sum and acts at the same iteration — a rank that raised alone would leave the others in a collective):
            #   slow   > 0: some chain launch waited a millisecond or more for missing workgroups (a collective or another
            #               process held CUs).  Results are valid; all ranks switch to one launch per convolution, which
            #               loses almost nothing under held CUs.  Marks of the first `chain_slow_grace_iters`
            #               iterations (RCCL sets its connections up inside the first collectives) are ignored.
            #   status > 0: a wait ran into its spin bound and the launch finished on unfinished neighbour data: raise,
            #               on every rank, instead of training on from garbage.
            slow = st = 0.0
            if self._log_health:
                slow, st = host[-2] * self.opt["world_size"], host[-1] * self.opt["world_size"]
                host = host[:-2]
            elif vals.is_cuda:   # one process: the device was just synchronised by the read above, the word is one 4-byte copy
                st = float(_C.load().neosr_conv_chain_status())
            self.log_dict = OrderedDict(zip(keys, host))
            self._log_dev = None
            self.chain_health_ok = st <= 0
            if act_on_health or not self.opt.get("dist", False):
                self._act_on_chain_health(slow, st)
            tot = self.log_dict.get("l_g_total")
            if tot is not None and tot != tot:
                msg = (f"{tc.red}NaN found, aborting training. Make sure you're using a proper "
                       f"learning rate.{tc.end}")
                raise ValueError(msg)
        return self.log_dict

    chain_health_ok = True

    def _act_on_chain_health(self, slow: float, st: float) -> None:
        """Every rank calls this with the same (all-reduced) words at the same iteration."""
        if st > 0:
            _C.load().neosr_set_conv_chain(0)
            msg = (f"conv chain launch aborted (status {int(st)}): a chain launch did not get all its workgroups resident; "
                   "the iterations since the last log read are invalid.  Chain launches are now off in this process "
                   "(NEOSR_AMD_CHAIN=0 avoids them from the start when the GPU is shared).")
            raise _C.NeosrAmdError(msg)
        if slow > 0:
            _C.check(_C.load().neosr_conv_chain_ack(_C.stream_ptr()), "neosr_conv_chain_ack")
        if slow > 0 and self._log_iters > self.chain_slow_grace_iters:
            _C.load().neosr_set_conv_chain(0)
            get_root_logger().warning(
                "conv chain launches waited >= 1 ms for resident workgroups on %d rank(s): switching every rank to one "
                "launch per convolution (results unaffected)", int(slow))
            self.chain_fallback = True

    def reduce_loss_dict(self, loss_dict: dict[str, torch.Tensor]) -> None:
        """Average the losses over ranks (intent of base.py:498-526); result read lazily."""
        with torch.no_grad():
            keys = list(loss_dict.keys())
            vals = torch.stack([v.detach().reshape(-1)[0].float() for v in loss_dict.values()])
            self._log_health = False
            if self.opt["dist"]:
                if vals.is_cuda:   # the chain launches' health words ride along (see get_current_log)
                    health = torch.empty(2, device=vals.device, dtype=torch.float32)
                    _C.check(_C.load().neosr_conv_chain_health(health.data_ptr(), _C.stream_ptr()), "neosr_conv_chain_health")
                    vals = torch.cat((vals, health))
                    self._log_health = True
                # asynchronous: nothing on the compute stream waits for it (it queues behind the gradient buckets
                # on the communication stream); completed and divided when the caller reads the log.  An all-reduce
                # rather than the reference's reduce-to-0: the health words must reach every rank.
                self._log_work = dist.all_reduce(vals, async_op=True)
            self._log_dev = (keys, vals)
            self._log_iters = self._iters_seen

    # -- device / parallel ----------------------------------------------------------------
    def model_to_device(self, net: nn.Module) -> nn.Module:
        """Move to the HIP device and re-home the parameters into one flat arena
        (base.py:120-149 wraps in DDP here; we all-reduce the flat gradient arena instead)."""
        self.precision_options()
        if not torch.cuda.is_available():
            msg = "neosr_amd models need a HIP device (no CPU fallback on the product path)"
            raise RuntimeError(msg)
        flatten_parameters_(net, self.device)   # host gather + ONE transfer; the parameters become device views
        net = net.to(self.device)               # (moves what is left: the buffers)
        if self.opt["dist"]:
            # DDP's constructor broadcasts rank 0's parameters AND buffers (base.py:140-146): ranks seed with
            # manual_seed + rank, so without this the spectral-norm weight_u / weight_v start different per rank
            dist.broadcast(net._neosr_arena, src=0)  # noqa: SLF001
            for b in net.buffers():
                if b.is_cuda:
                    dist.broadcast(b, src=0)
            net._neosr_sn_arena = flatten_sn_buffers_(net)  # noqa: SLF001
        return net

    def precision_options(self) -> None:
        """`fast_matmul` (train.py:168-173: TF32 convolutions, "medium" matmul precision), `use_amp` / `bfloat16`
        (neosr/models/image.py:117-127, 438-440: autocast + GradScaler) ask the reference for NARROWER arithmetic than fp32.
        Here each of them selects the one reduced-precision tier this path has — `neosr_set_fast_matmul(1)`: the F(4x4,3x3)
        forward / backward-data convolutions with two bf16 pieces per operand (16 significant bits; TF32 has 11, bf16
        autocast 8) on the bf16 MFMA, AND every nn.Linear product (forward NT, backward-data NN, weight-gradient TN) as
        the 3-term instead of the 6-term bf16 split (~2e-5 per product instead of ~1e-6: gemm_mfma.hip `fast3`), fp32
        accumulation, everything else fp32 — and nothing else: no autocast region, fp32 storage and gradients, so the GradScaler is the identity (scale 1, never skips a step; fp32 gradients do not
        underflow the way fp16 ones do) and the `log_dict` keys are the reference's.  Process-wide (the library switch is),
        logged once.  Without these keys the path is fp32 throughout."""
        asked = [k for k in ("fast_matmul", "use_amp", "bfloat16") if self.opt.get(k, False) is True]
        env = os.environ.get("NEOSR_AMD_FAST_MATMUL", "")   # "1" / "0" force the tier on / off whatever the options say (A/B runs)
        want = env == "1" or (bool(asked) and env != "0")
        _C.load()
        if want != bool(_C.FAST_MATMUL):   # (every model build sets it from ITS options: a later model without the keys is fp32)
            _C.set_fast_matmul(want)
        if asked and not getattr(base, "_precision_note", False):
            base._precision_note = True
            get_root_logger().warning(
                "%s: the F(4x4,3x3) convolutions run their products on the bf16 MFMA with two bf16 pieces per fp32 operand "
                "(16-bit significands, fp32 accumulation; ~1e-4 per layer) and the nn.Linear GEMMs (forward, backward-data and "
                "weight gradients) use the 3-term instead of the 6-term bf16 split (~2e-5 per product); storage, gradients, "
                "convolution weight gradients and every other kernel stay fp32, GradScaler = identity", " / ".join(asked))
        self.use_amp = False   # (what the closures test: nothing to scale)

    def graph_generator(self) -> None:
        """`compile = true` (base.py:136-137 gives the network to torch.compile): capture the train-mode forward /
        backward of an op-by-op dispatched generator into hipGraphs (neosr_amd/utils/graph.py).  The esrgan /
        compact generators already run as one C++ plan per pass and are left alone."""
        want = self.opt.get("compile", False) is True
        if os.environ.get("NEOSR_AMD_COMPILE") is not None:  # A/B switch for measurements
            want = os.environ["NEOSR_AMD_COMPILE"] == "1"
        if not want or not self.is_train:
            return
        logger = get_root_logger()
        from neosr_amd.archs.arch_util import HipNet
        from neosr_amd.utils.graph import graph_train_forward

        ds = self.opt["datasets"]["train"]
        if isinstance(self.net_g, HipNet) or not ds.get("patch_size") or not ds.get("batch_size"):
            logger.info("`compile = true`: nothing to capture for this generator")
            return
        p = int(ds["patch_size"])
        sample = torch.rand(int(ds["batch_size"]), int(self.opt["network_g"].get("in_chans", 3)), p, p,
                            device=self.device)
        graph_train_forward(self.net_g, sample)
        logger.info("`compile = true`: generator forward / backward captured as hipGraphs (%s)", tuple(sample.shape))

    def broadcast_buffers(self, net: nn.Module) -> None:
        """DDP `broadcast_buffers=True` (base.py:140-146): before every train-mode forward the buffers that a
        forward mutates — the spectral-norm power-iteration vectors — are re-sent from rank 0 (one message: they
        live in one flat arena).  The power iteration depends only on (W, u, v), never on the data, so with
        identical weights and deterministic kernels this keeps the ranks bit-identical rather than making them so;
        `opt["broadcast_buffers"] = false` drops the message."""
        arena = getattr(net, "_neosr_sn_arena", None)
        if self.opt["dist"] and arena is not None and net.training and self.opt.get("broadcast_buffers", True):
            dist.broadcast(arena, src=0)

    def get_bare_model(self, net: nn.Module) -> nn.Module:
        return getattr(net, "module", net) if not hasattr(net, "_neosr_arena") else net

    def get_optimizer(self, optim_type: str, params, lr: float, **kwargs):
        if optim_type in {"AdamW", "adamw"}:
            return optimizers.AdamW(params, lr, **kwargs)
        if optim_type in {"Adan_SF", "adan_sf"}:
            return optimizers.adan_sf(params, lr, **kwargs)
        if optim_type in {"Adam", "adam"}:
            return optimizers.Adam(params, lr, **kwargs)
        if optim_type in {"NAdam", "nadam"}:
            return optimizers.NAdam(params, lr, **kwargs)
        if optim_type in {"Adan", "adan"}:
            return optimizers.adan(params, lr, **kwargs)
        if optim_type in {"AdamW_Win", "adamw_win"}:
            return optimizers.adamw_win(params, lr, **kwargs)
        if optim_type in {"AdamW_SF", "adamw_sf"}:
            return optimizers.adamw_sf(params, lr, **kwargs)
        logger = get_root_logger()
        logger.error(f"{tc.red}Optimizer {optim_type} is not supported yet.{tc.end}")
        sys.exit(1)

    def setup_schedulers(self) -> None:
        train_opt = self.opt["train"]
        if train_opt.get("scheduler") is None:
            return
        sched = dict(train_opt["scheduler"])
        scheduler_type = sched.pop("type")
        if scheduler_type in {"MultiStepLR", "multisteplr"}:
            cls = torch.optim.lr_scheduler.MultiStepLR
        elif scheduler_type in {"CosineAnnealing", "cosineannealing"}:
            cls = torch.optim.lr_scheduler.CosineAnnealingLR
        else:
            get_root_logger().error(f"{tc.red}Scheduler {scheduler_type} is not implemented yet.{tc.end}")
            sys.exit(1)
        for optimizer in self.optimizers:
            self.schedulers.append(cls(optimizer, **sched))

    def _set_lr(self, lr_groups_l) -> None:
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for param_group, lr in zip(optimizer.param_groups, lr_groups):
                param_group["lr"] = lr

    def _get_init_lr(self):
        return [[v["initial_lr"] for v in o.param_groups] for o in self.optimizers]

    def update_learning_rate(self, current_iter: int, warmup_iter: int = -1) -> None:
        """base.py:229-254: scheduler step once per optimizer step, linear warm-up."""
        if current_iter > 0 and self.n_accumulated == 0:
            for scheduler in self.schedulers:
                scheduler.step()
        if current_iter < warmup_iter:
            init = self._get_init_lr()
            self._set_lr([[v / warmup_iter * current_iter for v in g] for g in init])

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    # -- checkpoints (wire format of base.py:281-475) ----------------------------------------
    @master_only
    def save_network(self, net, net_label: str, current_iter: int, param_key: str = "params") -> None:
        # the deferred NaN check (image.py:611-619) fires BEFORE anything is written.  This runs on rank 0 only: the chain
        # health words are left for the next read every rank makes (get_current_log), and nothing is written from
        # iterations an aborted chain launch invalidated — that read raises on all ranks.
        self.get_current_log(act_on_health=False)
        if not self.chain_health_ok:
            get_root_logger().error("save_network(%s): skipped — a conv chain launch was aborted (see the next log read)", net_label)
            return
        it = "latest" if current_iter == -1 else current_iter
        path = Path(self.opt["path"]["models"]) / f"{net_label}_{it}.pth"
        path.parent.mkdir(parents=True, exist_ok=True)
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        save_dict = {}
        for n, k in zip(nets, keys):
            sd = OrderedDict()
            for name, p in n.state_dict().items():
                if name == "n_averaged":
                    continue
                sd[name.removeprefix("module.")] = p.detach().cpu().clone()
            save_dict[k] = sd
        # base.py:325-354: schedule-free optimizers are switched to eval (x weights) around the write
        sf = [o for o, on in ((getattr(self, "optimizer_g", None), self.sf_optim_g),
                              (getattr(self, "optimizer_d", None), self.sf_optim_d)) if o is not None and on and self.is_train]
        self._write_with_retry(save_dict, path, sf, "model")

    def _write_with_retry(self, obj, path: Path, sf_optimizers, what: str) -> None:
        """base.py:325-354,446-470: schedule-free optimizers sit in eval mode (x weights) around the write; three
        attempts one second apart, then log and abort like the reference."""
        logger = get_root_logger()
        for o in sf_optimizers:
            o.eval()
        try:
            for retry in range(3):
                try:
                    torch.save(obj, path)
                    return
                except OSError as e:
                    logger.warning(f"{tc.red}Save {what} error ({e}). Remaining retry times: {2 - retry}{tc.end}")
                    time.sleep(1)
        finally:
            for o in sf_optimizers:
                o.train()
        logger.error(f"{tc.red}Cannot save {path}.{tc.end}")
        sys.exit(1)

    def load_network(self, net, load_path, param_key: str | None = None, strict: bool = True) -> None:
        load_net = torch.load(load_path, map_location="cpu", weights_only=True)
        if param_key is None:
            for k in ("params-ema", "params_ema", "params"):
                if k in load_net:
                    param_key = k
                    break
        if param_key is not None and param_key in load_net:
            load_net = load_net[param_key]
        load_net = OrderedDict((k.removeprefix("module."), v) for k, v in load_net.items())
        self.get_bare_model(net).load_state_dict(load_net, strict=strict)

    @master_only
    def save_training_state(self, epoch: int, current_iter: int) -> None:
        if current_iter == -1:
            return
        self.get_current_log(act_on_health=False)  # NaN check before the write, as in save_network (rank 0 only: no health action)
        if not self.chain_health_ok:
            get_root_logger().error("save_training_state: skipped — a conv chain launch was aborted (see the next log read)")
            return
        state = {"epoch": epoch, "iter": current_iter,
                 "optimizers": [o.state_dict() for o in self.optimizers],
                 "schedulers": [s.state_dict() for s in self.schedulers]}
        path = Path(self.opt["path"]["training_states"]) / f"{int(current_iter)}.state"
        path.parent.mkdir(parents=True, exist_ok=True)
        # base.py:446-470: the same schedule-free eval()/train() round trip as around save_network (the
        # state dicts were taken before it, so the file holds train-mode groups)
        sf = [o for o, on in ((getattr(self, "optimizer_g", None), self.sf_optim_g),
                              (getattr(self, "optimizer_d", None), self.sf_optim_d)) if o is not None and on and self.is_train]
        self._write_with_retry(state, path, sf, "training state")

    def resume_training(self, resume_state) -> None:
        assert len(resume_state["optimizers"]) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(resume_state["schedulers"]) == len(self.schedulers), "Wrong lengths of schedulers"
        for i, o in enumerate(resume_state["optimizers"]):
            self.optimizers[i].load_state_dict(o)
        for i, s in enumerate(resume_state["schedulers"]):
            self.schedulers[i].load_state_dict(s)


__all__ = ["allreduce_flat_", "base", "deepcopy"]
