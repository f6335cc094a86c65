"""F(2x2,3x3) vs F(4x4,3x3) on the conv shapes of the layer-composed networks (VGG19, U-Net, SwinIR / HAT convs, esrgan tail):
python tools/bench_wino_shapes.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

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


SHAPES = [  # (tag, B, H, K, N)
    ("vgg1_2", 32, 256, 64, 64), ("vgg2_1", 32, 128, 64, 128), ("vgg2_2", 32, 128, 128, 128), ("vgg3_1", 32, 64, 128, 256),
    ("vgg3_x", 32, 64, 256, 256), ("vgg4_1", 32, 32, 256, 512), ("vgg4_x", 32, 32, 512, 512), ("vgg5_x", 32, 16, 512, 512),
    ("unet4", 32, 64, 512, 256), ("unet5", 32, 128, 256, 128), ("unet6", 32, 256, 128, 64), ("unet7", 32, 256, 64, 64),
    ("vgg1_2 b8", 8, 256, 64, 64), ("vgg3_x b8", 8, 64, 256, 256), ("vgg4_x b8", 8, 32, 512, 512), ("vgg5_x b8", 8, 16, 512, 512),
    ("swin conv", 8, 64, 180, 180), ("swin up0", 8, 64, 64, 256), ("swin up2", 8, 128, 64, 256), ("swin cbu", 8, 64, 180, 64),
    ("hat conv", 4, 64, 180, 180), ("hat cab a", 4, 64, 180, 60), ("hat cab b", 4, 64, 60, 180),
    ("esr hr b16", 16, 256, 64, 64), ("esr body b16", 16, 64, 64, 64), ("esr hr b32", 32, 256, 64, 64),
]
for tag, B, H, K, N in SHAPES:
    x = torch.randn(B, H, H, K, device=dev)
    out = torch.empty(B, H, H, N, device=dev)
    w = torch.randn(N, K, 3, 3, device=dev) * 0.05
    bias = torch.randn(N, device=dev)
    fl = 2.0 * B * H * H * K * N * 9
    pack, w2, w4 = ops.conv3x3_pack_weights(w), ops.conv3x3_pack_wino(w), ops.conv3x3_pack_wino4(w)
    t2 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack, w_wino=w2))
    t4 = timeit(lambda: ops.conv3x3(x, w, bias, out=out, w_pack=pack, w_wino4=w4))
    print(f"{tag:12s} B={B:2d} {H:3d}x{H:3d} K={K:3d} N={N:3d}  F(2x2) {t2:8.1f} us {fl / t2 / 1e6:6.1f} TF-eq   F(4x4) {t4:8.1f} us {fl / t4 / 1e6:6.1f} TF-eq   ratio {t2 / t4:5.2f}")
    del x, out
