"""Differentiable layer fronts over the C ABI for networks that are composed op-by-op in Python
(U-Net-SN discriminator, VGG19 feature extractor) and the GAN / perceptual losses.

Every `autograd.Function` here moves channels-last `(B,H,W,C)` fp32 HBM tensors through HIP
kernels (`neosr_conv3x3[_wgrad]`, `neosr_space_to_depth2`, `neosr_bilinear_up2`, `neosr_maxpool2`,
`neosr_leaky_relu`, `neosr_spectral_norm_*`, `neosr_chc_loss_*`, `neosr_bce_logits_*`, …).  torch is
used for allocation, the autograd graph, and index plumbing on small weight tensors only.
"""

from __future__ import annotations

import weakref

import torch

from neosr_amd import _C
from neosr_amd.hip import ops

ACT_NONE, ACT_LRELU, ACT_RELU = _C.ACT_NONE, _C.ACT_LRELU, _C.ACT_RELU


def _st():
    return _C.stream_ptr()


# --------------------------------------------------------------------------------------------
# layout
# --------------------------------------------------------------------------------------------
class ToNHWC(torch.autograd.Function):
    """(B,C,H,W) -> (B,H,W,cs) channels-last (cs >= C, zero padded)."""

    @staticmethod
    def forward(ctx, x, cs):
        ctx.c = x.shape[1]
        return ops.nchw_to_nhwc(_C.require_device(x, "x"), cs)

    @staticmethod
    def backward(ctx, g):
        return ops.nhwc_to_nchw(g.contiguous(), ctx.c), None


class ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        ctx.cs = x.shape[3]
        return ops.nhwc_to_nchw(_C.require_device(x, "x").contiguous(), c)

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g.contiguous(), ctx.cs), None


class VGGInput(torch.autograd.Function):
    """(x - mean) / std fused with NCHW -> NHWC (vgg_arch.py:190-191)."""

    @staticmethod
    def forward(ctx, x, mean, std, cs):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, C_, H, W = x.shape
        out = torch.zeros(B, H, W, cs, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_norm_nchw_nhwc(x.data_ptr(), out.data_ptr(), B, C_, H, W, cs, mean, std, 0, _st()),
                 "neosr_norm_nchw_nhwc")
        ctx.meta = (B, C_, H, W, cs, mean, std)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        B, C_, H, W, cs, mean, std = ctx.meta
        g = g.contiguous()
        out = torch.empty(B, C_, H, W, device=g.device, dtype=torch.float32)
        _C.check(lib.neosr_norm_nchw_nhwc(g.data_ptr(), out.data_ptr(), B, C_, H, W, cs, mean, std, 1, _st()),
                 "neosr_norm_nchw_nhwc")
        return out, None, None, None


# --------------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------------
# Packed images (direct-to-LDS image = kind 0, Winograd image = kind 1) are cached ON the weight tensor object and
# rebuilt when the weights changed: torch in-place ops move `w._version`, the fused optimizer kernels (raw pointers)
# move `_C.WEIGHTS_EPOCH`, re-homing (flatten_parameters_) moves `data_ptr()`.  Weights marked frozen
# (`w._neosr_frozen = True`: the VGG feature extractor, never touched by an optimizer or an EMA) ignore the epoch and
# are packed once.  Temporaries (spectral-normalised weights) carry no cache.  Thin layers have their own kernels
# and are not packed.
#
# A cache miss re-packs EVERY registered image that is stale on that device in one `neosr_conv3x3_pack_many` call
# (ceil(n / 24) launches per kind) instead of one launch per (layer, mode, kind): after an optimizer step the first
# convolution of the next forward pass refreshes the whole network.
_PACKS: dict[tuple[int, int, int], "weakref.ref"] = {}  # (id(w), kind, mode) -> weakref(w)


# Set by neosr_amd.utils.graph before a hipGraph capture: the first packed-image request made while the stream is
# capturing re-packs every registered trainable weight (whatever its cache key says), so the launch becomes a node
# of the graph and every replay starts from the weights of that moment.
FORCE_REPACK_IN_CAPTURE = False
_S2D_PREMASK = __import__('os').environ.get('NEOSR_AMD_S2D_PREMASK', '1') != '0'  # A/B switch


def _pack_key(w):
    return (w._version, 0 if getattr(w, "_neosr_frozen", False) else _C.WEIGHTS_EPOCH, w.data_ptr(), _C.FAST_MATMUL)


def _image_floats(w, kind, mode):
    lib = _C.load()
    cout, cin = w.shape[0], w.shape[1]
    N, K = (cout, cin) if mode == ops.CONV_FWD else (cin, cout)
    fn = (lib.neosr_conv3x3_pack_bytes, lib.neosr_conv3x3_pack_wino_bytes, lib.neosr_conv3x3_pack_wino4_bytes)[kind]
    return fn(N, K) // 4


def _pack_now(w, mode, kind):
    return (ops.conv3x3_pack_weights, ops.conv3x3_pack_wino, ops.conv3x3_pack_wino4)[kind](w, mode)


def _packed(w: torch.Tensor, mode: int, kind: int):
    if min(w.shape[0], w.shape[1]) <= 4 or w.shape[0] % 4 or w.shape[1] % 4:
        return None
    if not hasattr(w, "__dict__"):
        return _pack_now(w, mode, kind)
    global FORCE_REPACK_IN_CAPTURE
    if FORCE_REPACK_IN_CAPTURE and torch.cuda.is_current_stream_capturing():
        FORCE_REPACK_IN_CAPTURE = False
        _repack_stale(w.device, force=True)
    cache = w.__dict__.setdefault("_neosr_packs", {})
    hit = cache.get((kind, mode))
    if hit is not None and hit[0] == _pack_key(w):
        return hit[1]
    if not isinstance(w, torch.nn.Parameter):
        # temporaries (spectral-normalised weights): a fresh tensor every forward, nothing to cache or to batch with
        return _pack_now(w, mode, kind)
    _PACKS[(id(w), kind, mode)] = weakref.ref(w)
    _repack_stale(w.device)
    return cache[(kind, mode)][1]


_FRESH: dict = {}   # device -> (weights epoch, fast_matmul mode) of the last completed _repack_stale


def images_fresh(device) -> bool:
    """every registered packed image of `device` was refreshed for the current weights epoch and arithmetic mode (callers that
    hold on to image tensors — the HAB block plans — skip their per-image lookups then; the tensors are re-packed in place)"""
    return _FRESH.get(device) == (_C.WEIGHTS_EPOCH, _C.FAST_MATMUL)


def _repack_stale(device, force: bool = False) -> None:
    lib = _C.load()
    todo = []
    for (wid, kind, mode), ref in list(_PACKS.items()):
        w = ref()
        if w is None or id(w) != wid:
            del _PACKS[(wid, kind, mode)]
            continue
        if w.device != device:
            continue
        cache = w.__dict__.setdefault("_neosr_packs", {})
        key = _pack_key(w)
        hit = cache.get((kind, mode))
        if hit is not None and hit[0] == key and not (force and w.requires_grad):
            continue
        assert w.is_contiguous() and w.shape[2:] == (3, 3)
        if hit is not None and hit[1].device == device:   # (a parameter's shape is fixed: the image keeps its size)
            dst = hit[1]
        else:
            dst = torch.empty(_image_floats(w, kind, mode), device=device, dtype=torch.float32)
        cache[(kind, mode)] = (key, dst)
        todo.append(_C.PackItem(w=w.data_ptr(), dst=dst.data_ptr(), w_cout=w.shape[0], w_cin=w.shape[1], mode=mode,
                                kind=kind))
    if todo:
        arr = (_C.PackItem * len(todo))(*todo)
        _C.check(lib.neosr_conv3x3_pack_many(arr, len(todo), _st()), "neosr_conv3x3_pack_many")
    if not torch.cuda.is_current_stream_capturing():
        _FRESH[device] = (_C.WEIGHTS_EPOCH, _C.FAST_MATMUL)


def packed_weights(w: torch.Tensor, mode: int):
    """Packed image of `w` for the direct-to-LDS kernel (`neosr_conv3x3_pack_weights`); see the note above."""
    return _packed(w, mode, 0)


def packed_wino(w: torch.Tensor, mode: int):
    """Winograd F(2x2,3x3) image of `w` (`neosr_conv3x3_pack_wino`), cached like `packed_weights`."""
    return _packed(w, mode, 1)


def packed_wino4(w: torch.Tensor, mode: int):
    """Winograd F(4x4,3x3) image of `w` (`neosr_conv3x3_pack_wino4`), cached like `packed_weights`."""
    return _packed(w, mode, 2)


WINO4_MIN_WGS = 64  # include/neosr_amd.h: NEOSR_WINO4_MIN_WGS


def wino_images(w: torch.Tensor, mode: int, B: int, H: int, W: int, n_out: int, ok: bool = True) -> dict:
    """The ONE Winograd image a plain 3x3 launch of this geometry will use, as keyword arguments of `ops.conv3x3`:
    F(4x4,3x3) under `neosr_set_winograd(2)` when the launch has enough 16 x 16-pixel x 32-cout workgroups to fill the
    chip, else F(2x2,3x3) (the library applies the same rule; packing only the image it will pick halves the pack work)."""
    if not ok:
        return {}
    wgs = B * -(-H // 16) * -(-W // 16) * -(-n_out // 32)
    if _C.load().neosr_get_winograd() == 2 and wgs >= WINO4_MIN_WGS:
        return {"w_wino4": packed_wino4(w, mode)}
    return {"w_wino": packed_wino(w, mode)}


class Conv3x3(torch.autograd.Function):
    """y = act(conv3x3(x'[..., :K], w) + b) (+ res), fused bias/activation/residual; x' = x or its
    nearest x2 upsampling (`ups`, folded into the conv loader).  backward = MFMA dgrad (activation
    derivative applied on load; 2x2 sum-pool after it for `ups`) + multi-conv wgrad kernel."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, ups, res, s2d_c=0, sole_consumer_is_conv=False):
        _C.require_device(x, "x")
        w = _C.require_device(w, "weight").contiguous()
        # Winograd kernel when eligible (a nearest-upsampled input is gathered by the kernels' DMA addresses)
        up = 2 if ups else 1
        wino = wino_images(w, ops.CONV_FWD, x.shape[0], up * x.shape[1], up * x.shape[2], w.shape[0], s2d_c == 0)
        y = ops.conv3x3(x, w, b, act=act, slope=slope, k_in=w.shape[1], ups=ups, res1=res,
                        w_pack=packed_weights(w, ops.CONV_FWD), s2d_c=s2d_c, **wino)
        ctx.s2d_c = s2d_c
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.leaves = (w, b)
        ctx.act, ctx.slope, ctx.has_bias, ctx.ups, ctx.has_res = act, slope, b is not None, ups, res is not None
        if act != ACT_NONE and res is not None:
            raise _C.NeosrAmdError("Conv3x3: activation + residual cannot be differentiated from the output")
        # conv -> act -> conv chains: the consumer's backward-data epilogue can apply THIS layer's activation derivative
        # (mask = own input > 0), which saves the producer-side elementwise pass over the gradient; see backward.
        # ReLU (VGG19) is always offered: its 0 / 1 mask is idempotent, so a gradient that was accumulated with other
        # contributions is simply masked again.  LeakyReLU is only offered when the caller promises that the next
        # conv3x3 is the ONLY consumer of this output (`sole_consumer_is_conv`, the U-Net's conv7 -> conv8 -> conv9).
        fold = getattr(x, "_neosr_act_fold", None)
        ctx.x_fold_slope = fold if fold is not None and not ups and s2d_c == 0 and x.shape[3] == w.shape[1] else None
        if act == ACT_RELU:
            y._neosr_act_fold = 0.0
        elif act == ACT_LRELU and sole_consumer_is_conv:
            y._neosr_act_fold = float(slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        g = g.contiguous()
        slope = ctx.slope if ctx.act == ACT_LRELU else 0.0
        # (the tag is only honoured while the tensor is exactly what the consumer wrote: autograd's input buffer may add
        # further contributions IN PLACE into a tagged tensor it holds the last reference to — that bumps `_version`)
        if y is not None and getattr(g, "_neosr_act_masked", None) == (y.data_ptr(), g._version):
            y = None  # the consumer's backward-data epilogue already multiplied g by act'(y) (below)
        if y is not None and (ctx.s2d_c == 0 or _S2D_PREMASK):
            # producer-side activation derivative: g <- g * act'(y) in ONE elementwise pass, so that neither the
            # backward-data nor the weight-gradient launch masks on load -> both are plain (packed / Winograd kernels)
            lib = _C.load()
            gm = torch.empty_like(g)
            _C.check(lib.neosr_leaky_relu(y.data_ptr(), g.data_ptr(), slope, gm.data_ptr(), g.numel(), _st()),
                     "neosr_leaky_relu")
            g, y = gm, None
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            plain = y is None  # no activation derivative to apply on load -> packed / Winograd kernels
            # x is the activation output of the layer below: fold ITS derivative into this epilogue and tell it so.  The
            # tag only survives when this gradient is the sole contribution to x (autograd hands the same tensor on); if
            # x has other consumers the sum arrives untagged and the layer below masks it itself (ReLU: idempotent).
            fold = plain and ctx.x_fold_slope is not None
            gx = ops.conv3x3(g, w, None, mode=ops.CONV_DGRAD, in_mask=y, mask_slope=slope,
                             w_pack=packed_weights(w, ops.CONV_DGRAD) if plain else None, s2d_c=ctx.s2d_c,
                             out_mask=x if fold else None, out_mask_slope=ctx.x_fold_slope if fold else 1.0,
                             **wino_images(w, ops.CONV_DGRAD, g.shape[0], g.shape[1], g.shape[2], w.shape[1],
                                           plain and ctx.s2d_c == 0))
            if fold:
                gx._neosr_act_masked = (x.data_ptr(), gx._version)
            if ctx.ups:
                gx = ops.pool2x2_sum(gx)
            if x.shape[3] > gx.shape[3]:  # conv read a channel prefix of a wider buffer
                pad = torch.zeros(*x.shape[:3], x.shape[3] - gx.shape[3], device=g.device)
                gx = torch.cat((gx, pad), 3)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            # data-parallel runs: straight into the parameters' slices of the exchange arena (transformer.GRAD_SLOT)
            from neosr_amd.hip import transformer as _tr

            wl, bl = ctx.leaves
            dw = _tr.GRAD_SLOT(wl) if _tr.GRAD_SLOT is not None else None
            db = _tr.GRAD_SLOT(bl) if _tr.GRAD_SLOT is not None and ctx.has_bias and bl is not None else None
            gw, gb = ops.conv3x3_wgrad(x, g, w.shape[0], w.shape[1], g_mask=y, mask_slope=slope,
                                       want_bias=ctx.has_bias, ups=ctx.ups, s2d_c=ctx.s2d_c, dw=dw, db=db)
        return gx, gw, gb, None, None, None, (g if ctx.has_res else None), None, None


def conv3x3(x, w, b=None, act=ACT_NONE, slope=0.0, ups=False, res=None, s2d_c=0, sole_consumer_is_conv=False):
    return Conv3x3.apply(x, w, b, act, slope, ups, res, s2d_c, sole_consumer_is_conv)


class SpaceToDepth2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, H, W, C_ = x.shape
        if H % 2 or W % 2:
            raise _C.NeosrAmdError("4x4/s2 conv needs even H, W")
        out = torch.empty(B, H // 2, W // 2, 4 * C_, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_space_to_depth2(x.data_ptr(), out.data_ptr(), B, H // 2, W // 2, C_, 0, _st()),
                 "neosr_space_to_depth2")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        g = g.contiguous()
        B, Hl, Wl, C4 = g.shape
        out = torch.empty(B, 2 * Hl, 2 * Wl, C4 // 4, device=g.device, dtype=torch.float32)
        _C.check(lib.neosr_space_to_depth2(g.data_ptr(), out.data_ptr(), B, Hl, Wl, C4 // 4, 1, _st()),
                 "neosr_space_to_depth2")
        return out


class SpaceToDepth2Skip(torch.autograd.Function):
    """(space_to_depth(x), x) for a LeakyReLU(slope) layer output x whose ONLY consumers are a 4x4 / stride-2 convolution
    (through the first output) and a skip addition (through the second) — the U-Net's x0 / x1 / x2.  backward: the
    depth-to-space of the convolution's input gradient, the skip gradient autograd would add in a pass of its own and the
    LeakyReLU derivative the producing layer would apply in another, in ONE pass (neosr_depth_to_space2_fused); the
    result is tagged so that the producer skips its own mask (see Conv3x3.backward).  Bit-identical to the separate
    passes."""

    @staticmethod
    def forward(ctx, x, slope):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, H, W, C_ = x.shape
        if H % 2 or W % 2:
            raise _C.NeosrAmdError("4x4/s2 conv needs even H, W")
        out = torch.empty(B, H // 2, W // 2, 4 * C_, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_space_to_depth2(x.data_ptr(), out.data_ptr(), B, H // 2, W // 2, C_, 0, _st()),
                 "neosr_space_to_depth2")
        ctx.save_for_backward(x)
        ctx.slope = float(slope)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, g, gskip):
        (x,) = ctx.saved_tensors
        lib = _C.load()
        B, H, W, C_ = x.shape
        if g is None:
            g = torch.zeros(B, H // 2, W // 2, 4 * C_, device=x.device, dtype=torch.float32)
        g = g.contiguous()
        gskip = None if gskip is None else gskip.contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_depth_to_space2_fused(g.data_ptr(), None if gskip is None else gskip.data_ptr(), x.data_ptr(),
                                                 ctx.slope, out.data_ptr(), B, H // 2, W // 2, C_, _st()),
                 "neosr_depth_to_space2_fused")
        out._neosr_act_masked = (x.data_ptr(), out._version)
        return out, None


_S2D_INDEX: dict[tuple, torch.Tensor] = {}


def expand_4x4s2_weight(w: torch.Tensor) -> torch.Tensor:
    """(N, C, 4, 4) stride-2/pad-1 kernel -> the (N, 4C, 3, 3) kernel acting on the space-to-depth
    tensor: tap ky -> (block row, sub row) = (0,1), (1,0), (1,1), (2,0); same for kx.  A single
    differentiable gather (index plumbing on the weight, 20 of 36 entries are structural zeros)."""
    return _Expand4x4s2.apply(w)


def _s2d_indices(N: int, C_: int, device) -> tuple[torch.Tensor, torch.Tensor]:
    """(fwd, inv): expanded[i] = cat(w, 0)[fwd[i]];  w_grad[j] = expanded_grad[inv[j]]"""
    key = (N, C_, device)
    hit = _S2D_INDEX.get(key)
    if hit is None:
        m = [(0, 1), (1, 0), (1, 1), (2, 0)]
        src = torch.full((N, 4, C_, 3, 3), N * C_ * 16, dtype=torch.long)  # -> appended zero
        base = torch.arange(N * C_ * 16).view(N, C_, 4, 4)
        dst = torch.arange(N * 4 * C_ * 9).view(N, 4, C_, 3, 3)
        inv = torch.empty(N, C_, 4, 4, dtype=torch.long)
        for ky, (by, dy) in enumerate(m):
            for kx, (bx, dx) in enumerate(m):
                src[:, dy * 2 + dx, :, by, bx] = base[:, :, ky, kx]
                inv[:, :, ky, kx] = dst[:, dy * 2 + dx, :, by, bx]
        hit = (src.view(-1).to(device), inv.view(-1).to(device))
        _S2D_INDEX[key] = hit
    return hit


class _Expand4x4s2(torch.autograd.Function):
    """weight scatter expressed as a gather in BOTH directions (torch's generic index backward is a
    serialised scatter-add: 19 ms per call for the 512x256x4x4 kernel)."""

    @staticmethod
    def forward(ctx, w):
        N, C_ = w.shape[:2]
        fwd, inv = _s2d_indices(N, C_, w.device)
        ctx.inv, ctx.shape = inv, w.shape
        flat = torch.cat((w.reshape(-1), w.new_zeros(1)))
        return flat.index_select(0, fwd).view(N, 4 * C_, 3, 3)

    @staticmethod
    def backward(ctx, g):
        return g.reshape(-1).index_select(0, ctx.inv).view(ctx.shape)


# The 3x3 expansion of a 4x4 / stride-2 kernel has 16 live (tap, sub-pixel) blocks of 36.  The direct kernels skip the 20
# zero blocks (`s2d_c`: 16 C multiplications per output and output channel); the Winograd F(4x4,3x3) kernel cannot skip
# anything (the transformed weights are dense) but needs only 36 / 16 = 2.25 per input channel of the 4 C — 9 C: 1.78x
# fewer.  So under Winograd mode 2 the strided layers of the U-Net discriminator (unet_arch.py:20-22: 64 -> 128 -> 256 -> 512)
# run as PLAIN 3x3 convolutions over the space-to-depth tensor (forward, backward-data and weight gradient; round 4,
# NEOSR_AMD_S2D_WINO=0 keeps the tap-skipping direct kernels).
_S2D_WINO = __import__('os').environ.get('NEOSR_AMD_S2D_WINO', '1') != '0'


def _s2d_c(c: int) -> int:
    return 0 if _S2D_WINO and _C.load().neosr_get_winograd() == 2 else c


def conv4x4s2(x, w, b=None, act=ACT_NONE, slope=0.0):
    """nn.Conv2d(C, N, 4, 2, 1) on channels-last x, as space-to-depth + the MFMA 3x3 kernel."""
    return conv3x3(SpaceToDepth2.apply(x), expand_4x4s2_weight(w), b, act, slope, s2d_c=_s2d_c(x.shape[3]))


def conv4x4s2_skip(x, x_slope, w, b=None, act=ACT_NONE, slope=0.0):
    """conv4x4s2(x, ...) for a LeakyReLU(x_slope) output x that also feeds a skip addition: returns (y, x') with x' to be
    used by the skip INSTEAD of x (see SpaceToDepth2Skip: x must have no other consumer)."""
    t, xs = SpaceToDepth2Skip.apply(x, x_slope)
    return conv3x3(t, expand_4x4s2_weight(w), b, act, slope, s2d_c=_s2d_c(x.shape[3])), xs


# --------------------------------------------------------------------------------------------
# resampling / pooling / pointwise
# --------------------------------------------------------------------------------------------
class BilinearUp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, H, W, C_ = x.shape
        out = torch.empty(B, 2 * H, 2 * W, C_, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_bilinear_up2(x.data_ptr(), out.data_ptr(), B, H, W, C_, 0, _st()), "neosr_bilinear_up2")
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        g = g.contiguous()
        B, H2, W2, C_ = g.shape
        out = torch.empty(B, H2 // 2, W2 // 2, C_, device=g.device, dtype=torch.float32)
        _C.check(lib.neosr_bilinear_up2(g.data_ptr(), out.data_ptr(), B, H2 // 2, W2 // 2, C_, 1, _st()),
                 "neosr_bilinear_up2")
        return out


class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        B, H, W, C_ = x.shape
        out = torch.empty(B, H // 2, W // 2, C_, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_maxpool2(x.data_ptr(), None, out.data_ptr(), B, H // 2, W // 2, C_, _st()), "neosr_maxpool2")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, C_ = x.shape
        gx = torch.zeros_like(x) if (H % 2 or W % 2) else torch.empty_like(x)
        _C.check(lib.neosr_maxpool2(x.data_ptr(), g.data_ptr(), gx.data_ptr(), B, H // 2, W // 2, C_, _st()),
                 "neosr_maxpool2")
        return gx


class Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _C.load()
        a = _C.require_device(a, "a").contiguous()
        b = _C.require_device(b, "b").contiguous()
        out = torch.empty_like(a)
        _C.check(lib.neosr_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _st()), "neosr_add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class LeakyReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        lib = _C.load()
        x = _C.require_device(x, "x").contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_leaky_relu(x.data_ptr(), None, slope, out.data_ptr(), x.numel(), _st()), "neosr_leaky_relu")
        ctx.save_for_backward(x)
        ctx.slope = slope
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        (x,) = ctx.saved_tensors
        g = g.contiguous()
        out = torch.empty_like(x)
        _C.check(lib.neosr_leaky_relu(x.data_ptr(), g.data_ptr(), ctx.slope, out.data_ptr(), x.numel(), _st()),
                 "neosr_leaky_relu")
        return out, None


# --------------------------------------------------------------------------------------------
# spectral norm
# --------------------------------------------------------------------------------------------
class SpectralNorm(torch.autograd.Function):
    """w = W / sigma(W; u, v) with one in-place power iteration of the (u, v) buffers in train mode
    (torch.nn.utils.spectral_norm semantics: u, v are constants for the gradient)."""

    @staticmethod
    def forward(ctx, w_orig, u, v, training, eps):
        lib = _C.load()
        w_orig = _C.require_device(w_orig, "weight_orig").contiguous()
        rows, cols = w_orig.shape[0], w_orig[0].numel()
        w = torch.empty_like(w_orig)
        sigma = torch.empty(1, device=w.device, dtype=torch.float32)
        scratch = torch.empty(rows, device=w.device, dtype=torch.float32)
        _C.check(lib.neosr_spectral_norm_fwd(w_orig.data_ptr(), u.data_ptr(), v.data_ptr(), w.data_ptr(),
                                             sigma.data_ptr(), scratch.data_ptr(), rows, cols, int(training),
                                             eps, _st()), "neosr_spectral_norm_fwd")
        # u, v advance with every train-mode forward: keep this call's values for the backward
        ctx.save_for_backward(w, u.clone(), v.clone(), sigma)
        return w

    @staticmethod
    def backward(ctx, gw):
        lib = _C.load()
        w, u, v, sigma = ctx.saved_tensors
        gw = gw.contiguous()
        rows, cols = w.shape[0], w[0].numel()
        out = torch.empty_like(w)
        ws = torch.empty(1032, device=w.device, dtype=torch.float32)
        _C.check(lib.neosr_spectral_norm_bwd(gw.data_ptr(), w.data_ptr(), u.data_ptr(), v.data_ptr(),
                                             sigma.data_ptr(), out.data_ptr(), ws.data_ptr(), rows, cols, _st()),
                 "neosr_spectral_norm_bwd")
        return out, None, None, None, None


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
class ChcLoss(torch.autograd.Function):
    """w * mean(clamp(charbonnier|l1((a-b)*pre), lo, hi)) (chc_loss with loss_lambda = 0)."""

    @staticmethod
    def forward(ctx, a, b, pre, huber, lo, hi, weight):
        lib = _C.load()
        a = _C.require_device(a, "pred").contiguous()
        b = _C.require_device(b, "target").contiguous()
        if a.shape != b.shape:
            raise _C.NeosrAmdError(f"chc_loss: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = torch.empty(1024, device=a.device, dtype=torch.float32)
        _C.check(lib.neosr_chc_loss_fwd(a.data_ptr(), b.data_ptr(), a.numel(), pre, int(huber), lo, hi, weight,
                                        out.data_ptr(), ws.data_ptr(), _st()), "neosr_chc_loss_fwd")
        ctx.save_for_backward(a, b)
        ctx.cfg = (pre, int(huber), lo, hi, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _C.load()
        a, b = ctx.saved_tensors
        pre, huber, lo, hi, weight = ctx.cfg
        g = g.contiguous().float()
        ga = torch.empty_like(a)
        _C.check(lib.neosr_chc_loss_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), a.numel(), pre, huber, lo, hi,
                                        weight, ga.data_ptr(), 0, _st()), "neosr_chc_loss_bwd")
        gb = -ga if ctx.needs_input_grad[1] else None
        return ga, gb, None, None, None, None, None


class ChcCosLoss(torch.autograd.Function):
    """chc_loss with loss_lambda != 0 (basic_loss.py:192-219) on NCHW tensors: the mean cosine distance over the
    channel vectors shifts every element before the clamp; `neosr_chc_cos_loss_fwd/bwd`."""

    COS_EPS = 1e-20  # nn.CosineSimilarity(dim=1, eps=1e-20)

    @staticmethod
    def forward(ctx, a, b, huber, lo, hi, lam, weight):
        lib = _C.load()
        a = _C.require_device(a, "pred").contiguous()
        b = _C.require_device(b, "target").contiguous()
        if a.shape != b.shape or a.dim() < 2:
            raise _C.NeosrAmdError(f"chc_loss: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        n, c = a.shape[0], a.shape[1]
        hw = a[0, 0].numel()
        out = torch.empty((), device=a.device, dtype=torch.float32)
        aux = torch.empty(2, device=a.device, dtype=torch.float32)
        ws = torch.empty(3072, device=a.device, dtype=torch.float32)
        _C.check(lib.neosr_chc_cos_loss_fwd(a.data_ptr(), b.data_ptr(), n, c, hw, int(huber), lo, hi, lam, weight,
                                            ChcCosLoss.COS_EPS, out.data_ptr(), aux.data_ptr(), ws.data_ptr(), _st()),
                 "neosr_chc_cos_loss_fwd")
        ctx.save_for_backward(a, b, aux)
        ctx.cfg = (n, c, hw, int(huber), lo, hi, lam, weight)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[1]:
            raise _C.NeosrAmdError("chc_loss (loss_lambda != 0): gradient w.r.t. the target is not implemented")
        lib = _C.load()
        a, b, aux = ctx.saved_tensors
        n, c, hw, huber, lo, hi, lam, weight = ctx.cfg
        g = g.contiguous().float()
        ga = torch.empty_like(a)
        _C.check(lib.neosr_chc_cos_loss_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), n, c, hw, huber, lo, hi, lam,
                                            weight, ChcCosLoss.COS_EPS, aux.data_ptr(), ga.data_ptr(), _st()),
                 "neosr_chc_cos_loss_bwd")
        return ga, None, None, None, None, None, None


class BceLogits(torch.autograd.Function):
    """weight * BCEWithLogits(x, constant target); also returns mean(x)."""

    @staticmethod
    def forward(ctx, x, target, weight):
        lib = _C.load()
        x = _C.require_device(x, "logits").contiguous()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        mean = torch.empty((), device=x.device, dtype=torch.float32)
        ws = torch.empty(2048, device=x.device, dtype=torch.float32)
        _C.check(lib.neosr_bce_logits_fwd(x.data_ptr(), x.numel(), target, weight, out.data_ptr(),
                                          mean.data_ptr(), ws.data_ptr(), _st()), "neosr_bce_logits_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (target, weight)
        ctx.mark_non_differentiable(mean)
        return out, mean

    @staticmethod
    def backward(ctx, g, _gmean):
        lib = _C.load()
        (x,) = ctx.saved_tensors
        target, weight = ctx.cfg
        g = g.contiguous().float()
        gx = torch.empty_like(x)
        _C.check(lib.neosr_bce_logits_bwd(x.data_ptr(), g.data_ptr(), x.numel(), target, weight, gx.data_ptr(),
                                          _st()), "neosr_bce_logits_bwd")
        return gx, None, None
