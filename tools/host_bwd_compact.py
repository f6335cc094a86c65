"""Where the host time of compact's backward pass goes (the config is host-bound): segments of CompactFunction.backward
timed with perf_counter, and the autograd engine's own share (run_backward total - the function body)."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from neosr_amd.hip import nets

args = types.SimpleNamespace(config="bench_compact", batch=0, arch=None, template_losses=False, augment=False)
opt = bench.load_opt(args, 1, 0)
from neosr_amd.models import build_model
import logging
logging.getLogger("neosr").setLevel(logging.WARNING)
model = build_model(opt)
batch = bench.make_batch(opt, torch.device("cuda"), 0)
T = {"alloc": 0.0, "body": 0.0, "n": 0}
orig_alloc = nets._alloc_flat_grads
def alloc(params):
    t0 = time.perf_counter(); r = orig_alloc(params); T["alloc"] += time.perf_counter() - t0; return r
nets._alloc_flat_grads = alloc
orig_bwd = nets.CompactFunction.backward
def bwd(ctx, gy):
    t0 = time.perf_counter(); r = orig_bwd(ctx, gy); T["body"] += time.perf_counter() - t0; T["n"] += 1; return r
nets.CompactFunction.backward = staticmethod(bwd)
for it in range(1, 21):
    model.feed_data(batch); model.optimize_parameters(it)
torch.cuda.synchronize()
for k in T: T[k] = 0
orig_tb = torch.Tensor.backward
tb = {"t": 0.0}
def tbackward(self, *a, **k):
    t0 = time.perf_counter(); r = orig_tb(self, *a, **k); tb["t"] += time.perf_counter() - t0; return r
torch.Tensor.backward = tbackward
N = 500
t0 = time.perf_counter()
for it in range(21, 21 + N):
    model.feed_data(batch); model.optimize_parameters(it)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("step host us %.1f | loss.backward() %.1f | CompactFunction.backward body %.1f (of which grad views %.1f)" % (
    1e6 * (t1 - t0) / N, 1e6 * tb["t"] / N, 1e6 * T["body"] / N, 1e6 * T["alloc"] / N))
