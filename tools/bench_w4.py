"""F(4x4,3x3) kernel only, RDB shapes (A/B runs of experiment builds: NEOSR_AMD_LIB=... python tools/bench_w4.py [B])."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = W = 64
dev = "cuda"
buf = torch.randn(B, H, W, 192, device=dev)
out = torch.empty(B, H, W, 192, device=dev)


def timeit(fn, n=50):
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


row = []
for K, N in [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64), (64, 64)]:
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    pack, wino4 = ops.conv3x3_pack_weights(w), ops.conv3x3_pack_wino4(w)
    t = timeit(lambda: ops.conv3x3(buf[..., :K], w, bias, out=out[..., :N], act=ops.ACT_LRELU, slope=0.2, w_pack=pack, w_wino4=wino4))
    row.append(f"K{K}N{N} {t:6.1f}")
print((os.environ.get("NEOSR_AMD_LIB", "x/default/y").split("/")[-2] + " " * 16)[:16], " ".join(row))
