"""staged vs direct-to-LDS conv on arbitrary shapes: python tools/bench_conv_shapes.py B H K,N [K,N ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

B, H = int(sys.argv[1]), int(sys.argv[2])
W = H
dev = "cuda"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for kn in sys.argv[3:]:
    K, N = map(int, kn.split(","))
    x = torch.randn(B, H, W, K, device=dev)
    out = torch.empty(B, H, W, N, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    fl = 2.0 * B * H * W * K * N * 9
    t0 = timeit(lambda: ops.conv3x3(x, w, bias, out=out))
    pack = ops.conv3x3_pack_weights(w)
    t1 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack))
    t2 = timeit(lambda: ops.conv3x3_pack_weights(w))
    print(f"B={B} {H}x{W} K={K:3d} N={N:3d}  staged {t0:7.1f} us {fl / t0 / 1e6:6.1f} TF   glds {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF   pack {t2:5.1f} us")
