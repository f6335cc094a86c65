"""GPU: whole transformer blocks on the C++ plans of csrc/blocks.hip (`neosr_tblock_forward/backward`: one library call
per block and direction) against the op-by-op composition of the same kernels from Python (rounds 1-3): outputs, input
gradients and every parameter gradient must agree BIT FOR BIT — the plans build the same descriptors in the same order.
References: neosr/archs/swinir_arch.py:231-392 (SwinTransformerBlock), neosr/archs/hat_arch.py:218-350 (HAB), :393-515
(OCAB)."""

from __future__ import annotations

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(net, x, gy, plans: bool, seed: int):
    from neosr_amd.hip import transformer as tr

    prev = tr.BLOCK_PLANS
    tr.BLOCK_PLANS = plans
    try:
        if net.training and not getattr(net, "_warm", False):
            # (the first train-mode forward of a network RECORDS its DropPath sites with one draw per site; every later one
            # takes all sites from one batched draw — another use of the RNG stream: compare like with like)
            net(x)
            net._warm = True
        torch.manual_seed(seed)   # DropPath draws
        net.zero_grad(set_to_none=True)
        xd = x.clone().requires_grad_(True)
        y = net(xd)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach().clone(), xd.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
    finally:
        tr.BLOCK_PLANS = prev


def _same(a, b):
    ya, gxa, ga = a
    yb, gxb, gb = b
    assert torch.equal(ya, yb)
    assert torch.equal(gxa, gxb)
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, bad[:6]


@pytest.mark.parametrize("train,drop", [(True, 0.1), (True, 0.0), (False, 0.1)])
def test_swinir_block_plan_is_bit_identical_to_op_by_op(train, drop):
    from neosr_amd.archs import swinir_arch as A

    torch.manual_seed(3)
    net = A.swinir_small(upscale=4, drop_path_rate=drop).to(DEV)
    net.train(train)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(2, 3, 32, 48, generator=g).to(DEV)
    gy = torch.randn(2, 3, 128, 192, generator=g).to(DEV)
    _same(_run(net, x, gy, True, 11), _run(net, x, gy, False, 11))


def test_swinir_block_plan_under_no_grad_and_twice():
    """inference (no autograd graph) and a second training pass re-using the parameters: same bits again"""
    from neosr_amd.archs import swinir_arch as A
    from neosr_amd.hip import transformer as tr

    torch.manual_seed(4)
    net = A.swinir_small(upscale=4, drop_path_rate=0.0).to(DEV).eval()
    x = torch.rand(1, 3, 24, 40, device=DEV)
    outs = []
    for plans in (True, False, True):
        tr.BLOCK_PLANS = plans
        with torch.no_grad():
            outs.append(net(x))
    tr.BLOCK_PLANS = True
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_network_copies_after_a_forward():
    """copy.deepcopy / torch.save of a network whose blocks have cached their ctypes descriptors: the copy
    drops the caches, rebuilds them on its own parameters and gives the same bits"""
    import copy
    import io
    from neosr_amd.archs import swinir_arch as A

    torch.manual_seed(5)
    net = A.swinir_small(upscale=4, drop_path_rate=0.0).to(DEV).train()
    x = torch.rand(1, 3, 24, 40, device=DEV)
    y = net(x)
    y.sum().backward()
    twin = copy.deepcopy(net)
    buf = io.BytesIO()
    torch.save(net, buf)
    with torch.no_grad():
        assert torch.equal(twin(x), net(x))


@pytest.mark.parametrize("ws,res,drop,cr", [(16, 32, 0.1, 3), (8, 16, 0.0, 3), (16, 64, 0.0, 3), (16, 32, 0.1, 4)])
def test_hat_block_plans_are_bit_identical_to_op_by_op(ws, res, drop, cr):
    """HAB (CAB branch, both of its convolutions' gradients, the channel gate, shifted windows) and OCAB; compress_ratio 4
    gives the CAB 6 inner channels (hat_s: 144 / 24) — not a multiple of 4, the convolutions take their generic kernels"""
    from neosr_amd.archs.hat_arch import hat

    torch.manual_seed(5)
    net = hat(img_size=res, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=ws, compress_ratio=cr,
              squeeze_factor=6, mlp_ratio=2, drop_path_rate=drop, upsampler="pixelshuffle", upscale=4).to(DEV).train()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, res, res, generator=g).to(DEV)
    gy = torch.randn(2, 3, 4 * res, 4 * res, generator=g).to(DEV)
    _same(_run(net, x, gy, True, 13), _run(net, x, gy, False, 13))


def test_hat_l_block_plans_full_width():
    """hat_l geometry (dim 180, 6 heads, windows of 16, CAB 180 -> 60 -> 180 with F(4x4) images) at B = 1, two groups"""
    from neosr_amd.archs.hat_arch import hat

    torch.manual_seed(6)
    net = hat(img_size=64, embed_dim=180, depths=(2, 2), num_heads=(6, 6), window_size=16, compress_ratio=3,
              squeeze_factor=30, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle", upscale=4).to(DEV).train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 3, 64, 64, generator=g).to(DEV)
    gy = torch.randn(1, 3, 256, 256, generator=g).to(DEV)
    _same(_run(net, x, gy, True, 17), _run(net, x, gy, False, 17))


def test_block_backward_side_stream_does_not_change_results():
    """`neosr_set_tblock_streams`: weight gradients (2) or the CAB branch (3) on the library's side stream (fork / join by
    events inside the call) vs everything on the caller's stream — bit-identical, run after run (a missing event wait would show up here)"""
    from neosr_amd import _C
    from neosr_amd.archs.hat_arch import hat

    lib = _C.load()
    torch.manual_seed(8)
    net = hat(img_size=32, embed_dim=60, depths=(2, 2), num_heads=(6, 6), window_size=16, compress_ratio=3,
              squeeze_factor=30, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle", upscale=4).to(DEV).train()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(4, 3, 32, 32, generator=g).to(DEV)
    gy = torch.randn(4, 3, 128, 128, generator=g).to(DEV)
    prev = lib.neosr_set_tblock_streams(1)
    try:
        ref = _run(net, x, gy, True, 19)
        n0 = lib.neosr_tblock_side_forks()
        assert n0 >= 0
        for mode in (2, 3):   # 2: weight gradients on the side stream; 3: the CAB branch (forward and backward) on it
            lib.neosr_set_tblock_streams(mode)
            before = lib.neosr_tblock_side_forks()
            for _ in range(4):
                _same(_run(net, x, gy, True, 19), ref)
            # the path was really taken (a silently inactive side stream would also be "bit-identical"): at least one fork per
            # HAB and direction and run
            assert lib.neosr_tblock_side_forks() - before >= 4 * 2 * 4, mode
    finally:
        lib.neosr_set_tblock_streams(prev)


def test_grouped_tn_gemm_equals_single_launches_bitwise():
    """`neosr_gemm_tn_group`: the four weight-gradient GEMMs of a block (fc2 / fc1 / proj / qkv shapes, two of them with a
    DropPath row scale) in one launch against four `neosr_gemm` launches: identical partial matrices, row counts included"""
    import ctypes as C

    from neosr_amd import _C

    lib = _C.load()
    g = torch.Generator().manual_seed(21)
    M = 2 * 64 * 64
    rs = torch.tensor([1.25, 0.0], device=DEV)
    shapes = [(180, 360, True), (360, 180, False), (180, 180, True), (540, 180, False)]   # (N_out, K_in, row scale)
    descs, keep = (_C.GemmDesc * 4)(), []
    singles = []
    for i, (n, k, scaled) in enumerate(shapes):
        dy, x = torch.randn(M, n, generator=g).to(DEV), torch.randn(M, k, generator=g).to(DEV)
        d = _C.GemmDesc(A=dy.data_ptr(), B=x.data_ptr(), M=n, N=k, K=M, lda=n, ldb=k, ldc=k, ldres=k, ldaux=k,
                        mode=_C.GEMM_TN, accumulate=2, rows_per_scale=64 * 64 if scaled else 0,
                        row_scale=rs.data_ptr() if scaled else None)
        nws = lib.neosr_gemm_workspace_bytes(d) // 4
        out = torch.zeros(n * k + n, device=DEV)
        ws1, ws2 = torch.zeros(nws, device=DEV), torch.zeros(nws, device=DEV)
        d.C, d.colsum_a, d.workspace = out.data_ptr(), out.data_ptr() + 4 * n * k, ws1.data_ptr()
        rc = lib.neosr_gemm(d, None)
        assert rc < 0, lib.neosr_last_error()
        singles.append((-rc, ws1))
        d.workspace = ws2.data_ptr()
        descs[i] = d
        keep.append((dy, x, out, ws2))
    ns = (C.c_int32 * 4)()
    _C.check(lib.neosr_gemm_tn_group(descs, 4, ns, None), "neosr_gemm_tn_group")
    torch.cuda.synchronize()
    for i, (n, k, _s) in enumerate(shapes):
        rows, ws1 = singles[i]
        assert ns[i] == rows
        slab = n * k + n
        assert torch.equal(keep[i][3][: rows * slab], ws1[: rows * slab]), i


def test_prelu_dslope_many_equals_single_calls_bitwise():
    import ctypes as C

    from neosr_amd import _C

    lib = _C.load()
    g = torch.Generator().manual_seed(22)
    npix, Cc, n = 2 * 64 * 64, 64, 5
    dA = [torch.randn(npix, Cc, generator=g).to(DEV) for _ in range(n)]
    z = [torch.randn(npix, Cc, generator=g).to(DEV) for _ in range(n)]
    wsb = lib.neosr_prelu_dslope_workspace_bytes(npix, Cc) // 4
    one = [torch.empty(Cc, device=DEV) for _ in range(n)]
    ws = torch.empty(wsb + 64, device=DEV)
    for i in range(n):
        _C.check(lib.neosr_prelu_dslope(dA[i].data_ptr(), z[i].data_ptr(), one[i].data_ptr(), ws.data_ptr(), npix, Cc, Cc,
                                        Cc, 0, None), "neosr_prelu_dslope")
    many = [torch.empty(Cc, device=DEV) for _ in range(n)]
    items = (_C.DslopeItem * n)(*[_C.DslopeItem(dA=dA[i].data_ptr(), z=z[i].data_ptr(), dslope=many[i].data_ptr())
                                  for i in range(n)])
    wsm = torch.empty(n * wsb + 64, device=DEV)
    _C.check(lib.neosr_prelu_dslope_many(items, n, wsm.data_ptr(), npix, Cc, Cc, Cc, None), "neosr_prelu_dslope_many")
    torch.cuda.synchronize()
    for a, b in zip(one, many):
        assert torch.equal(a, b)
    ref = (dA[0].double() * z[0].double().clamp(max=0)).sum(0)
    assert (many[0].double() - ref).abs().max() < 1e-3 * ref.abs().max()


def test_backward_tail_on_the_library_stream_is_bit_identical_and_joined():
    """Round 6: a block's weight-gradient tail (grouped TN GEMMs + batched column sums) runs on a library stream and
    `neosr_tblock_backward` returns with it in flight (include/neosr_amd.h: neosr_tblock_tail_join).  Same kernels on the
    same operands: gradients bit-identical to the tail on the caller's stream — read WITHOUT a device synchronisation in
    between (the end-of-backward join orders the caller's stream) —, also when a second backward accumulates into existing
    gradients, and the buffers of a call survive the allocator's reuse (a churn of allocations right behind backward)."""
    from neosr_amd import _C
    from neosr_amd.archs import swinir_arch as A
    from neosr_amd.archs.hat_arch import hat
    from neosr_amd.hip import transformer as tr

    lib = _C.load()
    nets = []
    torch.manual_seed(7)
    nets.append((A.swinir_small(upscale=4, drop_path_rate=0.0).to(DEV).train(), 32, 48))
    nets.append((hat(img_size=32, embed_dim=24, depths=(2, 2), num_heads=(2, 2), window_size=16, compress_ratio=3,
                     squeeze_factor=6, mlp_ratio=2, drop_path_rate=0.0, upsampler="pixelshuffle", upscale=4).to(DEV).train(),
                 32, 32))
    for net, H, W in nets:
        g = torch.Generator().manual_seed(3)
        x = torch.rand(2, 3, H, W, generator=g).to(DEV)
        gy = torch.randn(2, 3, 4 * H, 4 * W, generator=g).to(DEV)

        def grads(tail: bool, passes: int):
            prev = lib.neosr_set_tblock_tail(1 if tail else 0)
            try:
                net.zero_grad(set_to_none=True)
                t0 = lib.neosr_tblock_tails()
                for _ in range(passes):
                    net(x).backward(gy)
                    junk = [torch.full((1 << 20,), 3.0, device=DEV) for _ in range(8)]   # allocator churn on this stream
                    del junk
                issued = lib.neosr_tblock_tails() - t0
                out = [p.grad.clone() for p in net.parameters()]   # (clone on the caller's stream: no synchronise before)
                assert lib.neosr_tblock_tail_join(_C.stream_ptr()) == 0, "the backward pass must have joined its tails"
                assert not tr._TAIL_KEEP
                torch.cuda.synchronize()
                return out, issued
            finally:
                lib.neosr_set_tblock_tail(prev)

        for passes in (1, 2):
            a, na = grads(True, passes)
            b, nb = grads(False, passes)
            assert na > 0 and nb == 0, (na, nb)
            bad = [i for i, (u, v) in enumerate(zip(a, b)) if not torch.equal(u, v)]
            assert not bad, (passes, bad[:6])
