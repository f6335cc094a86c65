"""Small convolutions of HAT's CAB branch (180 -> 60 -> 180 at 64 x 64, B = 4: 64 pixel tiles of 16 x 16) under the three
algorithms: F(4x4,3x3) (default), F(2x2,3x3) (neosr_set_winograd(1)), direct (0); forward and backward-data."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd import _C
from neosr_amd.hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = W = 64
dev = "cuda"
lib = _C.load()


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for K, N in [(180, 60), (60, 180), (180, 180), (64, 64), (64, 256)]:
    x = torch.randn(B, H, W, K, device=dev)
    out = torch.empty(B, H, W, N, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    pack, wino, wino4 = ops.conv3x3_pack_weights(w), ops.conv3x3_pack_wino(w), ops.conv3x3_pack_wino4(w)
    row = []
    for mode, name in ((2, "F4"), (1, "F2"), (0, "direct")):
        prev = lib.neosr_set_winograd(mode)
        for n64 in ((-1, 0, 1) if mode == 2 else (-1,)):
            p64 = lib.neosr_set_wino4_n64(n64)
            t = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack, w_wino=wino, w_wino4=wino4))
            lib.neosr_set_wino4_n64(p64)
            row.append(f"{name}{'' if mode != 2 else '/n64=' + str(n64)} {t:6.1f}")
        lib.neosr_set_winograd(prev)
    print(f"B{B} K{K} N{N}: " + "  ".join(row))
