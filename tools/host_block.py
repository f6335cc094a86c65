"""Host cost of ONE transformer-block plan call (hat_l HAB geometry): the ctypes call into libneosr_amd alone vs the
whole autograd.Function, with the device drained before every call (no queue back-pressure in the numbers)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd import _C
from neosr_amd.archs.hat_arch import HAB
from neosr_amd.hip import transformer as tr

lib = _C.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
blk = HAB(180, (64, 64), 6, window_size=16, shift_size=8, compress_ratio=3, squeeze_factor=30, conv_scale=0.01,
          mlp_ratio=2.0).cuda().train()
from neosr_amd.hip.nets import flatten_parameters_
flatten_parameters_(blk)
x = torch.randn(B, 64, 64, 180, device="cuda", requires_grad=True)
gy = torch.randn(B, 64, 64, 180, device="cuda")
T = {"fwd_c": 0.0, "bwd_c": 0.0}
f0, b0 = lib.neosr_tblock_forward, lib.neosr_tblock_backward


def wrap(fn, key):
    def inner(*a):
        t = time.perf_counter()
        r = fn(*a)
        T[key] += time.perf_counter() - t
        return r
    return inner


lib.neosr_tblock_forward, lib.neosr_tblock_backward = wrap(f0, "fwd_c"), wrap(b0, "bwd_c")
for plans in (True, False):
    tr.BLOCK_PLANS = plans
    for k in T:
        T[k] = 0.0
    tf = tb = 0.0
    n = 40
    for it in range(n + 5):
        if it == 5:
            tf = tb = 0.0
            for k in T:
                T[k] = 0.0
        blk.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = blk(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        y.backward(gy)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        tf += t1 - t0
        tb += t3 - t2
        gf, gb = t2 - t0, t4 - t2
    print(f"plans={plans}: forward host {tf / n * 1e6:7.1f} us (library call {T['fwd_c'] / n * 1e6:7.1f} us), "
          f"backward host {tb / n * 1e6:7.1f} us (library call {T['bwd_c'] / n * 1e6:7.1f} us); "
          f"last fwd+drain {gf * 1e6:7.1f} us, bwd+drain {gb * 1e6:7.1f} us")
