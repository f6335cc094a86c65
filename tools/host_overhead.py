"""How long the HOST needs to enqueue one training step vs how long the GPU needs to run it (headline config)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

args = types.SimpleNamespace(config=(sys.argv[1] if len(sys.argv) > 1 else "bench_esrgan"), batch=0, arch=None, template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
if len(sys.argv) > 2:  # smaller patches: the same launches with a fraction of the device work -> the host's own cost shows
    opt["datasets"]["train"]["patch_size"] = int(sys.argv[2])
from neosr_amd.models import build_model
import logging
logging.getLogger("neosr").setLevel(logging.WARNING)
torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
for it in range(1, 6):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
host, total = [], []
for it in range(6, 16):
    t0 = time.perf_counter()
    model.feed_data(batch); model.optimize_parameters(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print("host enqueue ms/step: %.2f   enqueue + drain ms/step: %.2f" % (1e3 * sum(host) / len(host), 1e3 * sum(total) / len(total)))
if os.environ.get("HOST_PROFILE"):   # where the host time goes: cProfile of 3 steps, cumulative time by function (our modules)
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for it in range(16, 19):
        model.feed_data(batch); model.optimize_parameters(it)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(r"neosr_amd|run_backward", 45)
    st.sort_stats("tottime").print_stats(25)
