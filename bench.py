#!/usr/bin/env python
"""bench.py — LR-patches/sec of one full training iteration on the hot path (BASELINE.json metric).

Workload at N GPUs (weak scaling, one process per GPU, RCCL all-reduce of the flat grad arena):
BASELINE.json configs[1] = "esrgan RRDB x4, paired 64^2 LR synthetic, L1 only, batch=16" per GPU:
`feed_data` (inputs already resident in HBM) + `optimize_parameters` (RRDBNet fwd + L1 + bwd +
grad-clip + AdamW + EMA) through the same `image` model plugin a neosr TOML would build.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     — dominant conv kernel class: algorithmic FLOPs / HIP-event time of its launches,
                 measured in a dedicated profiled pass of the same K steps (events on the launch
                 stream, collected inside libneosr_amd), against the dense fp32 MFMA peak.
  cpu_baseline — the CPU oracle (oracle/neosr_oracle.py, a pure-PyTorch port of the same
                 iteration) timed on this host's cores on a bounded sample (rank 0, N=1 only).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 MFMA = fp32 vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
# SURVEY §8(d) / BASELINE.md §3: esrgan fwd+bwd per LR patch
GFLOP_PER_PATCH = 440.56
ALGO_MB_PER_PATCH = 2852.8


def make_opt(batch: int, world: int, rank: int, arch: str) -> dict:
    nets = {
        "esrgan": {"type": "esrgan"},
        "esrgan_small": {"type": "esrgan", "num_block": 2, "num_feat": 32, "num_grow_ch": 16},
        "compact": {"type": "compact"},
        "swinir_small": {"type": "swinir_small"},
        "swinir_medium": {"type": "swinir_medium"},
        "hat_s": {"type": "hat_s"},
        "hat_m": {"type": "hat_m"},
        "hat_l": {"type": "hat_l"},
    }
    return {
        "name": f"bench_{arch}", "model_type": "image", "scale": 4, "manual_seed": 1024,
        "is_train": True, "dist": world > 1, "rank": rank, "world_size": world, "num_gpu": world,
        "datasets": {"train": {"type": "paired", "patch_size": 64, "batch_size": batch,
                               "phase": "train", "scale": 4}},
        "path": {},
        "network_g": nets[arch],
        "train": {"ema": 0.999, "grad_clip": True,
                  "optim_g": {"type": "adamw", "lr": 1e-4, "betas": [0.9, 0.99], "weight_decay": 0.0},
                  "pixel_opt": {"type": "L1Loss", "loss_weight": 1.0}},
        "logger": {"total_iter": 1000000, "print_freq": 100, "save_checkpoint_freq": 100000,
                   "use_tb_logger": False},
    }


def pmc_traffic(kernel_class: str) -> dict | None:
    """HBM traffic per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (tools/profile_round.sh -> profiles/*_pmc_summary.json; bench.py cannot drive rocprofv3 on
    itself).  FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for gfx950 (it counts
    128-B requests at 64 B); both counters are in KiB."""
    files = sorted((ROOT / "profiles").glob("r*_bench_pmc_summary.json"))
    if not files:
        return None
    summ = json.loads(files[-1].read_text())
    # forward and backward-data launches of the RDB trunk are the same kernel symbol (gather form)
    want = "conv3x3_wgrad_multi_kernel" if "wgrad" in kernel_class else "conv3x3_glds_kernel"
    hits = [(d.get("dispatches_fetch", 0), name, d) for name, d in summ.items()
            if want in name and "FETCH_SIZE_per_dispatch" in d]
    if not hits:
        return None
    _, name, d = max(hits, key=lambda h: h[0])  # the instantiation the RDB trunk launches
    rd = 2.0 * d["FETCH_SIZE_per_dispatch"] * 1024
    wr = d.get("WRITE_SIZE_per_dispatch", 0.0) * 1024
    return {"bytes_per_launch": round(rd + wr), "read": round(rd), "write": round(wr),
            "source": files[-1].name, "kernel": name[:96]}


def cpu_baseline(arch: str, budget_s: float) -> dict:
    """Time the CPU oracle on a bounded sample of the same workload (B=1 batches of the same shapes)."""
    from oracle import neosr_oracle as orc
    from neosr_amd.archs import build_network

    torch.manual_seed(1024)
    net = build_network({"type": arch} if arch != "esrgan_small" else
                        {"type": "esrgan", "num_block": 2, "num_feat": 32, "num_grow_ch": 16})
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    fwd = (lambda P, x: orc.compact_forward(P, x, 4, "prelu")) if arch == "compact" else (
        lambda P, x: orc.rrdbnet_forward(P, x, 4))
    tr = orc.ImageTrainer(fwd, params, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.0, ema=0.999)
    b = 2
    lq, gt = torch.rand(b, 3, 64, 64), torch.rand(b, 3, 256, 256)
    tr.feed_data(lq, gt)
    tr.optimize_parameters()  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        tr.feed_data(lq, gt)
        tr.optimize_parameters()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 6:
            break
    return {"value": round(n * b / el, 4), "unit": "LR-patches/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n} iterations of the same training step at batch {b} (64x64 LR -> 256x256), "
                      f"torch {torch.__version__} CPU fp32, after 1 warm-up; host has {os.cpu_count()} logical cpus"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch (BASELINE configs[1]: 16)")
    ap.add_argument("--arch", default="esrgan", choices=["esrgan", "esrgan_small", "compact", "swinir_small", "swinir_medium", "hat_s", "hat_m", "hat_l"])
    ap.add_argument("--workload", default="paired_l1", choices=["paired_l1", "otf_gan", "swinir_percep"],
                    help="paired_l1 = BASELINE configs[1] (headline); otf_gan = configs[2]: otf degradation + "
                         "unet D + VGG perceptual + GAN (use --batch 32; with --arch hat_l --batch 4 = configs[4]); swinir_percep = configs[3]: "
                         "swinir_medium, L1 + VGG perceptual (use --batch 8)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-oracle timing (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--template-losses", action="store_true",
                    help="otf_gan: the shipped template loss stack (options/train_esrgan_otf.toml:118-141): "
                         "mssim + consistency + perceptual + gan, no L1")
    ap.add_argument("--augment", action="store_true",
                    help="also enable the template batch augmentations (options/train_esrgan_otf.toml:17-18)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback")
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; NEOSR_BENCH_BACKEND=gloo lets the control flow be exercised with several
        # ranks on ONE device (RCCL refuses duplicate GPUs)
        dist.init_process_group(os.environ.get("NEOSR_BENCH_BACKEND", "nccl"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from neosr_amd import _C
    from neosr_amd.models import build_model
    from neosr_amd.utils.options import set_global_opt

    import logging
    logging.getLogger("neosr").setLevel(logging.WARNING)

    if args.workload == "swinir_percep" and not args.arch.startswith("swinir"):
        args.arch = "swinir_medium"
    opt = make_opt(args.batch, world, rank, args.arch)
    if args.workload == "swinir_percep":
        opt["train"]["perceptual_opt"] = {"type": "vgg_perceptual_loss", "loss_weight": 1.0, "criterion": "chc"}
        opt["train"]["optim_g"] = {"type": "adan_sf", "lr": 1e-3, "betas": [0.98, 0.92, 0.987], "weight_decay": 0.02,
                                   "schedule_free": True, "warmup_steps": 1600}
    if args.workload == "otf_gan":
        from tools.bench_degrade import DEG_TABLE
        opt["model_type"] = "otf"
        opt["degradations"] = dict(DEG_TABLE)
        opt["datasets"]["train"].update({"type": "otf", "queue_size": 180})
        opt["network_d"] = {"type": "unet"}
        # template optimizers (options/train_esrgan_otf.toml:103-117)
        opt["train"]["optim_g"] = {"type": "adan_sf", "lr": 8e-4, "betas": [0.98, 0.92, 0.987], "weight_decay": 0.02,
                                   "schedule_free": True, "warmup_steps": 1600}
        opt["train"]["optim_d"] = {"type": "adan_sf", "lr": 5e-4, "betas": [0.98, 0.92, 0.99], "weight_decay": 0.02,
                                   "schedule_free": True}
        opt["train"]["perceptual_opt"] = {"type": "vgg_perceptual_loss", "loss_weight": 0.5, "criterion": "chc"}
        opt["train"]["gan_opt"] = {"type": "gan_loss", "gan_type": "bce", "loss_weight": 0.3}
    if args.template_losses:
        opt["train"].pop("pixel_opt", None)
        opt["train"]["mssim_opt"] = {"type": "mssim_loss", "loss_weight": 1.0}
        opt["train"]["consistency_opt"] = {"type": "consistency_loss", "loss_weight": 1.0}
    if args.augment:
        opt["datasets"]["train"].update({"augmentation": ["none", "mixup", "cutmix", "resizemix", "cutblur"],
                                         "aug_prob": [0.5, 0.1, 0.1, 0.1, 0.5]})
    set_global_opt(opt)
    torch.manual_seed(1024 + rank)
    model = build_model(opt)
    dev = torch.device("cuda")
    B = args.batch
    if args.workload == "otf_gan":
        import random
        import numpy as np
        from neosr_amd.data.degradations import KernelSampler
        random.seed(1024 + rank)
        ks = KernelSampler(np.random.default_rng(1024 + rank)).otf_kernel_batch(opt["degradations"], B)
        batch = {"gt": torch.rand(B, 3, 512, 512, device=dev), **{k: v.to(dev) for k, v in ks.items()}}
    else:
        batch = {"lq": torch.rand(B, 3, 64, 64, device=dev), "gt": torch.rand(B, 3, 256, 256, device=dev)}

    def step(it: int) -> None:
        model.feed_data(batch)
        model.optimize_parameters(it)

    def barrier() -> None:
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    for _ in range(args.warmup):
        it += 1
        step(it)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        it += 1
        step(it)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = model.get_current_log().get("l_g_pix", model.get_current_log().get("l_g_total"))

    roofline = None
    # every rank runs the profiled steps (they contain the gradient all-reduce); rank 0 reports
    if not args.no_roofline and not args.arch.startswith(("swinir", "hat")):
        lib = _C.load()
        # per-kernel durations are only meaningful without overlap: the trunk's launch chains are put
        # back on one stream for this pass (the timed region above ran the default, two chains)
        prev_streams = lib.neosr_set_num_streams(1)
        lib.neosr_prof_enable(1)
        nprof = max(1, min(args.steps, 3))
        for _ in range(nprof):
            it += 1
            step(it)
        ms = (C.c_double * 6)()
        ln = (C.c_longlong * 6)()
        fl = (C.c_double * 6)()
        by = (C.c_double * 6)()
        _C.check(lib.neosr_prof_collect(ms, ln, fl, by), "neosr_prof_collect")
        lib.neosr_prof_enable(0)
        lib.neosr_set_num_streams(prev_streams)
        # classes 0 / 1 are the two uses of ONE kernel symbol (the gather-form backward-data is a forward-shaped
        # launch); 4 / 5 are the few staged / thin launches outside the RDB trunk
        names = ["conv3x3_glds_kernel (forward launches)", "conv3x3_glds_kernel (backward-data launches)",
                 "conv3x3_wgrad_multi_kernel", "conv3x3_wgrad_reduce_kernel",
                 "staged + thin conv kernels (forward)", "staged + thin conv kernels (backward-data)"]
        kern = {}
        for i, nm in enumerate(names):
            if ln[i]:
                kern[nm] = {"launches": int(ln[i]), "avg_us": round(1e3 * ms[i] / ln[i], 2),
                            "total_ms": round(ms[i], 3),
                            "tflops": round(fl[i] / (ms[i] * 1e9), 2) if ms[i] > 0 else None,
                            "algo_GBps": round(by[i] / (ms[i] * 1e6), 1) if ms[i] > 0 else None}
        dom = max(range(3), key=lambda i: ms[i])
        ach = fl[dom] / (ms[dom] * 1e9) if ms[dom] > 0 else 0.0
        allms = sum(ms[i] for i in (0, 1, 2, 4, 5))
        allfl = sum(fl[i] for i in (0, 1, 2, 4, 5))
        glds_avg = 1e3 * (ms[0] + ms[1]) / max(1, ln[0] + ln[1])
        tr = pmc_traffic(names[dom]) if args.arch == "esrgan" and B == 16 else None
        roofline = {"bound": "mfma", "kernel": names[dom], "achieved": round(ach, 2),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": tr["bytes_per_launch"] if tr else None, "traffic_detail": tr,
                    "algo_bytes_per_launch": round(by[dom] / max(1, ln[dom])),
                    "avg_launch_us": round(1e3 * ms[dom] / max(1, ln[dom]), 2),
                    # all launches of the symbol, directly comparable with the rocprofv3 row of conv3x3_glds_kernel
                    "conv3x3_glds_kernel_avg_us": round(glds_avg, 2),
                    "all_conv_tflops": round(allfl / (allms * 1e9), 2) if allms > 0 else None,
                    # whole step as timed (two launch chains, loss + optimizer included)
                    "step_tflops": round(allfl / nprof / (elapsed / args.steps * 1e12), 2),
                    "step_frac": round(allfl / nprof / (elapsed / args.steps * 1e12) / PEAK_F32_MFMA_TFLOPS, 4),
                    "hbm_algo_frac_of_8TBps": round((by[dom] / (ms[dom] * 1e6)) / PEAK_HBM_GBS, 4) if ms[dom] > 0 else None,
                    "kernels": kern,
                    "method": "HIP events around every launch of the class on the launch stream, separate "
                              "profiled pass after the timed region with the trunk on ONE stream "
                              "(neosr_set_num_streams(1)); profiles/r01_bench_kernel_stats.csv is rocprofv3 of "
                              "`NEOSR_AMD_STREAMS=1 python bench.py`, ..._2chains.csv of the default run"}

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
    if rank != 0:
        return
    patches = world * B * args.steps
    value = patches / elapsed
    out = {
        "metric": "LR-patches/sec (64x64 -> 256x256 x4) fwd+bwd+optimizer step",
        "value": round(value, 3), "unit": "LR-patches/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.arch} RRDB x4, paired 64x64 LR synthetic (U[0,1)), L1 only, "
                               f"AdamW + grad-clip + EMA, batch={B}/GPU (BASELINE configs[1])"
                   if args.arch == "esrgan" else f"{args.arch} x4 L1 batch={B}/GPU (not the headline config)",
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "gflop_per_patch": GFLOP_PER_PATCH if args.arch == "esrgan" else None},
        "whole_step_tflops": round(value * GFLOP_PER_PATCH / 1e3, 2) if args.arch == "esrgan" else None,
        "final_l_g_pix": loss,
        "roofline": roofline,
    }
    if args.workload == "otf_gan":
        cfg = "configs[4]" if args.arch.startswith("hat") else "configs[2]"
        out["config"]["workload"] = (f"{args.arch} x4 + unet-SN D + VGG19 perceptual (random weights) + GAN, "
                                     f"otf degradation from 512x512 GT, adan_sf x2 (template), batch={B}/GPU (BASELINE {cfg})")
    if args.workload == "swinir_percep":
        out["config"]["workload"] = (f"{args.arch} x4, paired 64x64 LR synthetic, L1 + VGG19 perceptual (random weights), "
                                     f"window-attention path, adan_sf (template) + grad-clip + EMA, batch={B}/GPU (BASELINE configs[3])")
    if world == 1 and args.cpu_budget > 0 and args.workload == "paired_l1" and not args.arch.startswith(("swinir", "hat")):
        out["cpu_baseline"] = cpu_baseline(args.arch, args.cpu_budget)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
