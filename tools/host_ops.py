"""Host time per autograd Function / ATen op of one training step (torch.profiler, CPU side, all threads — the backward
pass runs on autograd's device thread, which cProfile (tools/host_profile.py) does not see)."""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

args = types.SimpleNamespace(config=(sys.argv[1] if len(sys.argv) > 1 else "bench_hat_l_otf_gan"), batch=0, arch=None,
                             template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
from neosr_amd.models import build_model
import logging
logging.getLogger("neosr").setLevel(logging.WARNING)
torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
for it in range(1, 5):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for it in range(5, 5 + N):
        model.feed_data(batch); model.optimize_parameters(it)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
tot = sum(e.self_cpu_time_total for e in rows)
print(f"total self CPU time per step: {tot / N / 1e3:.1f} ms")
for e in rows[:45]:
    print(f"{e.self_cpu_time_total / N / 1e3:8.2f} ms  {e.count / N:8.1f} calls  {e.key[:90]}")
