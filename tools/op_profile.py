"""ATen-level op counts of one training step (which torch ops still launch kernels / memcpys around our C calls)."""
import os, sys, types, collections
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
for it in range(1, 4):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    model.feed_data(batch); model.optimize_parameters(4)
    torch.cuda.synchronize()
agg = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::zeros", "aten::zero_", "aten::fill_", "aten::sum", "aten::div_", "aten::bernoulli_", "aten::to", "aten::_to_copy"):
        shp = str(e.input_shapes)[:60]
        st = ""
        for fr in (e.stack or [])[:12]:
            if "neosr_amd" in fr or "bench.py" in fr:
                st = fr.split("neosr_amd/")[-1][:60]; break
        agg[(e.name, shp, st)] += 1
for k, v in agg.most_common(40):
    print(v, k)
