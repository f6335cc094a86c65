"""Per-shape timing of the 3x3 kernels (register-staged, direct-to-LDS, Winograd F(2x2,3x3) / F(4x4,3x3)) on the RDB shapes.
usage: python tools/bench_conv.py [B] [H]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = "cuda"
buf = torch.randn(B, H, W, 192, device=dev)
out = torch.empty(B, H, W, 192, device=dev)


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


for K, N in ([(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)] if H <= 64 else [(64, 64)]):
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    pack = ops.conv3x3_pack_weights(w)
    fl = 2.0 * B * H * W * K * N * 9
    t0 = timeit(lambda: ops.conv3x3(buf[..., :K], w, bias, out=out[..., :N], act=ops.ACT_LRELU, slope=0.2))
    t1 = timeit(lambda: ops.conv3x3(buf[..., :K], w, bias, out=out[..., :N], act=ops.ACT_LRELU, slope=0.2, w_pack=pack))
    wino = ops.conv3x3_pack_wino(w)
    t2 = timeit(lambda: ops.conv3x3(buf[..., :K], w, bias, out=out[..., :N], act=ops.ACT_LRELU, slope=0.2, w_pack=pack, w_wino=wino))
    wino4 = ops.conv3x3_pack_wino4(w)
    t3 = timeit(lambda: ops.conv3x3(buf[..., :K], w, bias, out=out[..., :N], act=ops.ACT_LRELU, slope=0.2, w_pack=pack, w_wino4=wino4))
    if os.environ.get("BENCH_DGRAD"):  # gather-form backward-data shape: K rows of gradient -> N channels, mask epilogue
        wd = torch.randn(K, N, 3, 3, device=dev) * 0.05
        pk = ops.conv3x3_pack_weights(wd, ops.CONV_DGRAD)
        w2, w4 = ops.conv3x3_pack_wino(wd, ops.CONV_DGRAD), ops.conv3x3_pack_wino4(wd, ops.CONV_DGRAD)
        kw = dict(mode=ops.CONV_DGRAD, out=out[..., :N], out_mask=buf[..., :N], out_mask_slope=0.2, w_pack=pk)
        td2 = timeit(lambda: ops.conv3x3(buf[..., :K], wd, None, w_wino=w2, **kw))
        td4 = timeit(lambda: ops.conv3x3(buf[..., :K], wd, None, w_wino4=w4, **kw))
        print(f"   dgrad K={K:3d} N={N:2d}  F(2x2) {td2:7.1f} us   F(4x4) {td4:7.1f} us")
    print(f"K={K:3d} N={N:2d}  F(4x4,3x3) {t3:7.1f} us {fl / t3 / 1e6:6.1f} TF(direct-equivalent) {fl / t3 / 1e6 / 4:6.1f} TF executed")
    print(f"K={K:3d} N={N:2d}  staged {t0:7.1f} us {fl / t0 / 1e6:6.1f} TF   glds {t1:7.1f} us {fl / t1 / 1e6:6.1f} TF   "
          f"winograd {t2:7.1f} us {fl / t2 / 1e6:6.1f} TF(direct-equivalent)")
