"""Does splitting the batch into two independent conv chains on two streams hide launch gaps /
prologues?  usage: python tools/bench_chain.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neosr_amd.hip import ops

H = W = 64
dev = "cuda"
shapes = [(64, 32), (96, 32), (128, 32), (160, 32), (192, 64)]
ws = [(torch.randn(N, K, 3, 3, device=dev) * 0.02) for K, N in shapes]
bs = [torch.zeros(N, device=dev) for K, N in shapes]
packs = [ops.conv3x3_pack_weights(w) for w in ws]


def rdb(buf, nxt, use_pack):
    for i, (K, N) in enumerate(shapes):
        o = buf[..., K:K + N] if i < 4 else nxt[..., :64]
        ops.conv3x3(buf[..., :K], ws[i], bs[i], out=o, act=ops.ACT_LRELU if i < 4 else ops.ACT_NONE, slope=0.2,
                    w_pack=packs[i] if use_pack else None)


def chain(bufs, use_pack, n=12):
    for j in range(n):
        rdb(bufs[j & 1], bufs[(j + 1) & 1], use_pack)


def run(nstreams, use_pack, B=16):
    per = B // nstreams
    bufs = [[torch.randn(per, H, W, 192, device=dev) * 0.1 for _ in range(2)] for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def go():
        for s, bb in zip(streams, bufs):
            with torch.cuda.stream(s):
                chain(bb, use_pack)
    go(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    fl = sum(2.0 * B * H * W * K * N * 9 for K, N in shapes) * 12
    print(f"streams={nstreams} pack={use_pack}: {dt * 1e3:7.3f} ms  {fl / dt / 1e12:6.1f} TF")


for ns in (1, 2, 4):
    for up in (False, True):
        run(ns, up)
