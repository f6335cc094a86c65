#!/usr/bin/env python
"""Rounding of the split-bf16 tiers of the F(4x4,3x3) products against a float64 convolution (numpy, CPU; K = 192, N = 64, the RDB
conv5 shape): fp32 F(4x4) as the kernels compute it, bf16x3 with 6 / 9 cross products (fp32-faithful), two
pieces per operand (the `fast_matmul` tier, neosr_set_fast_matmul) and plain bf16.  Round-5 numbers in DESIGN.md §0.1 / §3."""
import numpy as np
rng = np.random.default_rng(0)
K, N, H, W = 192, 64, 32, 32
x = rng.standard_normal((K, H + 2, W + 2)).astype(np.float32)
# leaky-relu-like activations are positive-skewed; keep gaussian
w = (rng.standard_normal((N, K, 3, 3)) / np.sqrt(9 * K)).astype(np.float32)

def bf16_trunc(a):
    return (a.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
def bf16_rne(a):
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)
def split(a, n, rne_last=True):
    pieces = []; r = a.copy()
    for i in range(n):
        p = bf16_rne(r) if (rne_last and i == n - 1) else bf16_trunc(r)
        pieces.append(p); r = (r - p).astype(np.float32)
    return pieces

def direct64(x, w):
    y = np.zeros((N, H, W))
    for a in range(3):
        for b in range(3):
            y += np.einsum("nk,khw->nhw", w[:, :, a, b].astype(np.float64), x[:, a:a + H, b:b + W].astype(np.float64))
    return y

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],[0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],[1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)
U = np.einsum("ia,nkab,jb->nkij", G, w.astype(np.float64), G).astype(np.float32)
BT32, AT32 = BT.astype(np.float32), AT.astype(np.float32)

def wino(mode):
    y = np.zeros((N, H, W), np.float32)
    if mode != "f32":
        na, nw, terms = mode
        Us = split(U, nw)
    for ty in range(0, H, 4):
        for tx in range(0, W, 4):
            d = x[:, ty:ty + 6, tx:tx + 6]
            V = np.einsum("ia,kab,jb->kij", BT32, d, BT32).astype(np.float32)
            if mode == "f32":
                Mm = np.einsum("nkij,kij->nij", U, V).astype(np.float32)
            else:
                Vs = split(V, na)
                Mm = np.zeros((N, 6, 6), np.float32)
                for (i, j) in terms:
                    # bf16 products exact in fp32; accumulate in fp32 (einsum in f32)
                    Mm += np.einsum("nkij,kij->nij", Us[j], Vs[i]).astype(np.float32)
            y[:, ty:ty + 4, tx:tx + 4] = np.einsum("ia,nab,jb->nij", AT32, Mm, AT32)
    return y

ref = direct64(x, w); scale = np.abs(ref).max()
def rep(name, y): print(f"{name:44s} max|err|/max|y| = {np.abs(y-ref).max()/scale:.2e}  relL2 = {np.linalg.norm(y-ref)/np.linalg.norm(ref):.2e}")
rep("F(4x4) fp32", wino("f32"))
rep("bf16x3, 6 products", wino((3, 3, [(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)])))
rep("bf16x3, 9 products", wino((3, 3, [(i,j) for i in range(3) for j in range(3)])))
rep("bf16x2 both, 4 products", wino((2, 2, [(0,0),(0,1),(1,0),(1,1)])))
rep("bf16x2 both, 3 products", wino((2, 2, [(0,0),(0,1),(1,0)])))
rep("act x3, weights x2, 6 products", wino((3, 2, [(0,0),(0,1),(1,0),(1,1),(2,0),(2,1)])))
rep("bf16 x1 (plain bf16)", wino((1, 1, [(0,0)])))
