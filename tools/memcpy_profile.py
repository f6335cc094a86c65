"""Which host calls issue device copies / memsets in one training step (torch.profiler, runtime + kernel level)."""
import os, sys, types, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

args = types.SimpleNamespace(config=(sys.argv[1] if len(sys.argv) > 1 else "bench_esrgan"), batch=0, arch=None,
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.feed_data(batch); model.optimize_parameters(4)
    torch.cuda.synchronize()
names = collections.Counter()
allk = collections.Counter()
for e in prof.events():
    n = e.name
    allk[(n[:60], str(e.device_type)[-4:])] += 1
    if "emcpy" in n.lower() or "emset" in n.lower() or "copy" in n.lower() or "fill" in n.lower():
        st = ""
        for fr in (e.stack or [])[:14]:
            if "neosr_amd" in fr or "bench.py" in fr:
                st = fr.split("neosr_amd/")[-1][:70]; break
        names[(n[:50], str(e.device_type)[-4:], st)] += 1
print("-- copy-like events")
for k, v in names.most_common(25):
    print(v, k)
print("-- most frequent events")
for k, v in allk.most_common(25):
    print(v, k)
