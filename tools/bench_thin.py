"""Timing of the thin-layer conv kernels (first / last layers) at HR size. usage: python tools/bench_thin.py [B] [H]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x64 = torch.randn(B, H, W, 64, device=dev)
x4 = torch.randn(B, H, W, 4, device=dev)
w_last = torch.randn(3, 64, 3, 3, device=dev) * 0.05   # conv_last: 64 -> 3
w_first = torch.randn(64, 3, 3, 3, device=dev) * 0.05  # conv_first-like: 3 -> 64
o4 = torch.empty(B, H, W, 4, device=dev)
o64 = torch.empty(B, H, W, 64, device=dev)
mb = lambda *ts: sum(t.numel() for t in ts) * 4 / 1e6
for name, fn, traffic in [
    ("fwd  64->3 ", lambda: ops.conv3x3(x64, w_last, None, out=o4[..., :3]), mb(x64) + mb(o4) * 0.75),
    ("dgrad 3->64", lambda: ops.conv3x3(x4[..., :3], w_last, None, mode=ops.CONV_DGRAD, out=o64), mb(x4) + mb(o64)),
    ("fwd  3->64 ", lambda: ops.conv3x3(x4[..., :3], w_first, None, out=o64), mb(x4) + mb(o64)),
    ("dgrad 64->3", lambda: ops.conv3x3(x64, w_first, None, mode=ops.CONV_DGRAD, out=o4[..., :3], in_mask=x64, mask_slope=0.0), 2 * mb(x64) + mb(o4) * 0.75),
    ("wgrad N=3  ", lambda: ops.conv3x3_wgrad(x64, x4[..., :3], 3, 64), mb(x64) + mb(x4)),
    ("wgrad K=3  ", lambda: ops.conv3x3_wgrad(x4[..., :3], x64, 64, 3), mb(x64) + mb(x4)),
]:
    t = timeit(fn)
    print(f"{name}: {t:8.1f} us   {traffic / t * 1e6 / 1e6:7.1f} GB/s algorithmic")
