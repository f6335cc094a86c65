"""Per-shape timing of the 3x3 kernels on the transformer generators' convolution shapes (CAB 180 -> 60 -> 180,
180 -> 180 residual convs) at the token counts of configs[3] / configs[4].  usage: python tools/bench_conv_hat.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = W = 64
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


for K, N in [(180, 60), (60, 180), (180, 180)]:
    x = torch.randn(B, H, W, K, device=dev)
    out = torch.empty(B, H, W, N, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    pack = ops.conv3x3_pack_weights(w)
    wino = ops.conv3x3_pack_wino(w)
    fl = 2.0 * B * H * W * K * N * 9
    t1 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack))
    t2 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack, w_wino=wino))
    w4 = ops.conv3x3_pack_wino4(w)
    t5 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack, w_wino=wino, w_wino4=w4))
    g = torch.randn(B, H, W, N, device=dev)
    gx = torch.empty(B, H, W, K, device=dev)
    pd, wd = ops.conv3x3_pack_weights(w, ops.CONV_DGRAD), ops.conv3x3_pack_wino(w, ops.CONV_DGRAD)
    t3 = timeit(lambda: ops.conv3x3(g, w, None, mode=ops.CONV_DGRAD, out=gx, w_pack=pd, w_wino=wd))
    t4 = timeit(lambda: ops.conv3x3_wgrad(x, g, N, K))
    print(f"B={B} K={K:3d} N={N:3d}: fwd glds {t1:6.1f} us {fl / t1 / 1e6:6.1f} TF | fwd wino {t2:6.1f} us {fl / t2 / 1e6:6.1f} | fwd wino4 {t5:6.1f} us | "
          f"dgrad wino {t3:6.1f} us {fl / t3 / 1e6:6.1f} | wgrad {t4:6.1f} us {fl / t4 / 1e6:6.1f} TF-eq")
