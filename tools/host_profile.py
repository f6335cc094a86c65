"""cProfile of the host side of one training step (where does the Python dispatch time go?)."""
import cProfile, os, pstats, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

args = types.SimpleNamespace(config=(sys.argv[1] if len(sys.argv) > 1 else "bench_hat_l_otf_gan"), batch=0, arch=None,
                             template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
from neosr_amd.models import build_model
import logging
logging.getLogger("neosr").setLevel(logging.WARNING)
torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
for it in range(1, 4):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(4, 7):
    model.feed_data(batch); model.optimize_parameters(it)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.print_callers("_named_members")
st.print_callers("_repack_stale")
