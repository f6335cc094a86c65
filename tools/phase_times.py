"""Serial GPU time of the phases of one training step (device drained between phases): generator forward, loss branches,
generator backward, discriminator phase, optimizer steps.  python tools/phase_times.py <config> (GPU box)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["NEOSR_AMD_D_OVERLAP"] = "0"
import torch
import bench
import logging

args = types.SimpleNamespace(config=(sys.argv[1] if len(sys.argv) > 1 else "bench_hat_l_otf_gan"), batch=0, arch=None,
                             template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
from neosr_amd.models import build_model
logging.getLogger("neosr").setLevel(logging.WARNING)
torch.manual_seed(1024)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
for it in range(1, 4):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
model.net_g.forward = timed("G forward", model.net_g.forward)
if model.cri_perceptual: model.cri_perceptual.forward = timed("perceptual forward (VGG x2)", model.cri_perceptual.forward)
if model.net_d is not None: model.net_d.forward = timed("D forward (x3)", model.net_d.forward)
model.optimizer_g.step = timed("optimizer G", model.optimizer_g.step)
if model.net_d is not None: model.optimizer_d.step = timed("optimizer D", model.optimizer_d.step)
model.feed_data = timed("feed_data", model.feed_data)
orig_bw = torch.Tensor.backward
n_bw = [0]
def bw(self, *a, **k):
    n_bw[0] += 1
    name = "G backward (incl. VGG / D data gradients)" if n_bw[0] % 3 == 1 or model.net_d is None else "D backward (x2)"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_bw(self, *a, **k)
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r
torch.Tensor.backward = bw
N = 5
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(4, 4 + N):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
print(f"{args.config}: serial step {tot * 1e3:.2f} ms")
for k, v in T.items():
    print(f"  {k:48s} {v / N * 1e3:8.2f} ms")
print(f"  {'(sum)':48s} {sum(T.values()) / N * 1e3:8.2f} ms")
