"""cProfile of the host side of a training step: which Python / C-ABI calls the enqueue time of a step goes to.
usage: python tools/host_profile.py <config> [steps]"""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "bench_compact"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
args = types.SimpleNamespace(config=cfg, batch=0, arch=None, template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
from neosr_amd.models import build_model
import logging
logging.getLogger("neosr").setLevel(logging.WARNING)
torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
for it in range(1, 11):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(11, 11 + steps):
    model.feed_data(batch); model.optimize_parameters(it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
