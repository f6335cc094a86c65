"""DDP rehearsal on a 1-GPU lease: what does a collective that holds CUs on the exchange stream cost the
headline step, with the RRDB trunk as chain launches (every workgroup must be resident) vs one launch per convolution?

A single-rank RCCL group runs the real data-parallel code path (`opt["dist"] = True`: gradient marks recorded inside the
RRDB backward plan, three buckets + the arena head issued on the exchange stream by GradSync); every `dist.all_reduce`
is followed on the same stream by `cu_hog` (tools/micro/cu_hog.hip): k workgroups that each occupy a CU for t
microseconds — the footprint of an 8-GPU ring all-reduce of that bucket over xGMI (RCCL: up to 32-64 channels = CUs,
~22 MB per bucket at ~50-100 GB/s algorithmic -> 0.3-0.6 ms; 1-2 ms is the pessimistic end).

    python tools/ddp_contention.py            -> one JSON line per (chain, k, t) to stdout
"""
import ctypes as C
import json
import os
import socket
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import logging

import torch
import torch.distributed as dist

import bench
from neosr_amd import _C
from neosr_amd.utils.dist_util import init_dist

logging.getLogger("neosr").setLevel(logging.WARNING)
init_dist("pytorch")
hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libcu_hog.so"))
hog.cu_hog.argtypes = [C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
sink = torch.zeros(16, device="cuda", dtype=torch.int32)
HOG = {"k": 0, "us": 0, "calls": 0}
_all_reduce = dist.all_reduce


def all_reduce_with_hog(t, *a, **kw):
    w = _all_reduce(t, *a, **kw)
    if HOG["k"] and t.numel() > 1_000_000:   # gradient buckets only (not the loss scalars)
        hog.cu_hog(HOG["k"], HOG["us"], sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        HOG["calls"] += 1
    return w


dist.all_reduce = all_reduce_with_hog
lib = _C.load()
config = sys.argv[1] if len(sys.argv) > 1 else "bench_esrgan"
args = types.SimpleNamespace(config=config, batch=0, arch=None, template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
opt["dist"] = True
from neosr_amd.models import build_model

torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
B = opt["datasets"]["train"]["batch_size"]


def timed(steps=20, warm=4):
    it = [0]

    def step():
        it[0] += 1
        model.feed_data(batch)
        model.optimize_parameters(it[0])

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    HOG["calls"] = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, HOG["calls"] / steps


for chain in (1, 0):
    lib.neosr_set_conv_chain(chain)
    base = None
    for k, us in ((0, 0), (16, 500), (32, 500), (64, 500), (32, 1000), (64, 1000), (64, 2000), (128, 1000)):
        HOG["k"], HOG["us"] = k, us
        ms, calls = timed()
        base = ms if base is None else base
        print(json.dumps({"config": config, "chain": chain, "hog_cus": k, "hog_us": us, "hog_launches_per_step": calls,
                          "ms_per_step": round(ms, 3), "patches_per_s": round(B / ms * 1e3, 1),
                          "loss_vs_no_hog_pct": round(100 * (ms / base - 1), 2),
                          "chain_status": lib.neosr_conv_chain_status()}), flush=True)
dist.destroy_process_group()
