"""ctypes binding of ``libneosr_amd.so`` (the C ABI declared in ``include/neosr_amd.h``).

This is the *only* door between the Python host layer and the HIP kernels: plain pointers and
sizes go in, nothing torch-typed crosses the boundary.  There is deliberately no fallback: if the
library is missing, or a tensor is not on a HIP device, the caller gets an exception.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "lib" / "libneosr_amd.so"

c_float_p = C.POINTER(C.c_float)
c_void_pp = C.POINTER(C.c_void_p)


class ConvDesc(C.Structure):
    """neosr_conv_desc"""

    _fields_ = [
        ("in_", C.c_void_p),
        ("in_mask", C.c_void_p),
        ("mask_slopes", C.c_void_p),
        ("in_prelu", C.c_void_p),
        ("w", C.c_void_p),
        ("bias", C.c_void_p),
        ("prelu", C.c_void_p),
        ("res1", C.c_void_p),
        ("res2", C.c_void_p),
        ("out", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("K", C.c_int32),
        ("N", C.c_int32),
        ("w_cout", C.c_int32),
        ("w_cin", C.c_int32),
        ("in_cs", C.c_int32),
        ("mask_cs", C.c_int32),
        ("out_cs", C.c_int32),
        ("res1_cs", C.c_int32),
        ("res2_cs", C.c_int32),
        ("res1_nch", C.c_int32),
        ("res2_nch", C.c_int32),
        ("mode", C.c_int32),
        ("ups", C.c_int32),
        ("act", C.c_int32),
        ("accumulate", C.c_int32),
        ("mask_slope", C.c_float),
        ("slope", C.c_float),
        ("alpha", C.c_float),
        ("alpha2", C.c_float),
        ("w_pack", C.c_void_p),
        ("out_mask", C.c_void_p),
        ("out_mask_cs", C.c_int32),
        ("out_mask_slope", C.c_float),
        ("s2d_c", C.c_int32),
        ("reserved0", C.c_int32),
        ("w_wino", C.c_void_p),
        ("w_wino4", C.c_void_p),
        ("out_mask_slopes", C.c_void_p),
        ("out2", C.c_void_p),
        ("out2_cs", C.c_int32),
        ("out_mask_gelu", C.c_int32),
    ]


class WgradDesc(C.Structure):
    """neosr_wgrad_desc"""

    _fields_ = [
        ("in_", C.c_void_p),
        ("in_prelu", C.c_void_p),
        ("g", C.c_void_p),
        ("g_mask", C.c_void_p),
        ("mask_slopes", C.c_void_p),
        ("dw", C.c_void_p),
        ("db", C.c_void_p),
        ("workspace", C.c_void_p),
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("K", C.c_int32),
        ("N", C.c_int32),
        ("in_cs", C.c_int32),
        ("g_cs", C.c_int32),
        ("mask_cs", C.c_int32),
        ("ups", C.c_int32),
        ("accumulate", C.c_int32),
        ("mask_slope", C.c_float),
        ("scale", C.c_float),
        ("s2d_c", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class AdamWDesc(C.Structure):
    """neosr_adamw_desc"""

    _fields_ = [
        ("param", C.c_void_p),
        ("grad", C.c_void_p),
        ("exp_avg", C.c_void_p),
        ("exp_avg_sq", C.c_void_p),
        ("ema", C.c_void_p),
        ("norm_ws", C.c_void_p),
        ("n", C.c_int64),
        ("lr", C.c_float),
        ("beta1", C.c_float),
        ("beta2", C.c_float),
        ("eps", C.c_float),
        ("weight_decay", C.c_float),
        ("max_norm", C.c_float),
        ("ema_decay", C.c_float),
        ("grad_scale", C.c_float),
        ("step", C.c_int32),
    ]


class AdanDesc(C.Structure):
    """neosr_adan_desc"""

    _fields_ = (
        [(n, C.c_void_p) for n in ("param", "grad", "exp_avg", "exp_avg_sq", "exp_avg_diff", "z", "neg_pre_grad",
                                   "ema", "norm_ws")]
        + [("n", C.c_int64)]
        + [(n, C.c_float) for n in ("lr", "beta1", "beta2", "beta3", "eps", "weight_decay", "ckp1", "max_norm",
                                    "ema_decay", "grad_scale")]
        + [(n, C.c_int32) for n in ("step", "first_step", "schedule_free")]
    )


class OptimDesc(C.Structure):
    """neosr_optim_desc"""

    _fields_ = (
        [(n, C.c_void_p) for n in ("param", "grad", "s0", "s1", "s2", "s3", "ema", "norm_ws")]
        + [("n", C.c_int64), ("c", C.c_float * 12)]
        + [(n, C.c_float) for n in ("max_norm", "ema_decay", "grad_scale")]
        + [("kind", C.c_int32), ("flags", C.c_int32)]
    )


class FsamDesc(C.Structure):
    """neosr_fsam_desc"""

    _fields_ = (
        [(n, C.c_void_p) for n in ("param", "grad", "momentum", "old_p", "norm_ws")]
        + [("n", C.c_int64)]
        + [(n, C.c_float) for n in ("rho", "sigma", "lmbda", "grad_scale")]
        + [("first", C.c_int32), ("adaptive", C.c_int32)]
    )


OPT_ADAM, OPT_NADAM, OPT_ADAN, OPT_ADAMW_SF, OPT_ADAMW_WIN = 1, 2, 3, 4, 5


class RRDBNetCfg(C.Structure):
    """neosr_rrdbnet_cfg"""

    _fields_ = [
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("num_in_ch", C.c_int32),
        ("num_out_ch", C.c_int32),
        ("num_feat", C.c_int32),
        ("num_block", C.c_int32),
        ("num_grow_ch", C.c_int32),
        ("training", C.c_int32),
    ]


class CompactCfg(C.Structure):
    """neosr_compact_cfg"""

    _fields_ = [
        ("B", C.c_int32),
        ("H", C.c_int32),
        ("W", C.c_int32),
        ("num_in_ch", C.c_int32),
        ("num_out_ch", C.c_int32),
        ("num_feat", C.c_int32),
        ("num_conv", C.c_int32),
        ("upscale", C.c_int32),
        ("act_type", C.c_int32),
        ("training", C.c_int32),
    ]


class PackItem(C.Structure):
    """neosr_pack_item"""

    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("w_cout", C.c_int32), ("w_cin", C.c_int32),
                ("mode", C.c_int32), ("kind", C.c_int32)]


class ColsumItem(C.Structure):
    """neosr_colsum_item"""

    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32), ("ld", C.c_int32),
                ("accumulate", C.c_int32)]


class GemmDesc(C.Structure):
    """neosr_gemm_desc"""

    _fields_ = [(n, C.c_void_p) for n in ("A", "B", "C", "bias", "res", "aux_in", "aux_out", "row_scale", "workspace")] + [
        (n, C.c_int32)
        for n in ("M", "N", "K", "lda", "ldb", "ldc", "ldres", "ldaux", "rows_per_scale", "mode", "gelu", "accumulate")
    ] + [("colsum_a", C.c_void_p)]


class WattnDesc(C.Structure):
    """neosr_wattn_desc"""

    _fields_ = (
        [(n, C.c_void_p) for n in ("qkv", "rpb_table", "out", "lse", "dout", "dqkv", "d_rpb_table", "workspace")]
        + [(n, C.c_int32) for n in ("B", "H", "W", "C", "heads", "ws", "shift", "accumulate_rpb")]
        + [("scale", C.c_float)]
    )


class FattnDesc(C.Structure):
    """neosr_fattn_desc"""

    _fields_ = (
        [(n, C.c_void_p) for n in ("qkv", "rpb_table", "out", "lse", "dout", "dqkv", "d_rpb_table", "workspace")]
        + [(n, C.c_int32) for n in ("B", "H", "W", "C", "heads", "ws", "ks", "shift", "accumulate_rpb")]
        + [("scale", C.c_float)]
    )


TBLOCK_PARAMS = ("n1_w", "n1_b", "rpb", "qkv_w", "qkv_b", "proj_w", "proj_b", "n2_w", "n2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")
TBLOCK_CAB_PARAMS = ("c0_w", "c0_b", "c2_w", "c2_b", "ca1_w", "ca1_b", "ca2_w", "ca2_b")
TBLOCK_IMAGES = tuple(f"{c}_{k}_{m}" for c in ("c0", "c2") for m in ("f", "d") for k in ("pack", "wino", "wino4"))


class TBlockDesc(C.Structure):
    """neosr_tblock_desc"""

    _fields_ = (
        [(n, C.c_int32) for n in ("B", "H", "W", "C", "heads", "ws", "ks", "shift", "hidden", "attn", "cab_mid", "cab_sq")]
        + [(n, C.c_float) for n in ("scale", "eps1", "eps2", "conv_scale")]
        + [(n, C.c_void_p) for n in TBLOCK_PARAMS + TBLOCK_CAB_PARAMS]
        + [(n, C.c_void_p) for n in ("c0_pack_f", "c0_wino_f", "c0_wino4_f", "c0_pack_d", "c0_wino_d", "c0_wino4_d",
                                     "c2_pack_f", "c2_wino_f", "c2_wino4_f", "c2_pack_d", "c2_wino_d", "c2_wino4_d")]
        + [("drop_scale", C.c_void_p), ("drop_scale2", C.c_void_p)]
    )


class TBlockGrads(C.Structure):
    """neosr_tblock_grads"""

    _fields_ = [(n, C.c_void_p) for n in TBLOCK_PARAMS + TBLOCK_CAB_PARAMS]


class DslopeItem(C.Structure):
    """neosr_dslope_item"""

    _fields_ = [("dA", C.c_void_p), ("z", C.c_void_p), ("dslope", C.c_void_p)]


class MsssimDesc(C.Structure):
    """neosr_msssim_desc"""

    _fields_ = [("partial", C.c_void_p * 5), ("npix", C.c_int64 * 5), ("nblk", C.c_int32 * 5), ("weights", C.c_float * 5),
                ("loss_weight", C.c_float), ("nscales", C.c_int32), ("loss", C.c_void_p), ("gscal", C.c_void_p)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_PRELU, ACT_GELU = 0, 1, 2, 3, 4
CONV_FWD, CONV_DGRAD = 0, 1

# name -> (restype, argtypes); mirrors include/neosr_amd.h one to one
_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
SIGNATURES: dict[str, tuple] = {
    "neosr_last_error": (C.c_char_p, []),
    "neosr_build_info": (C.c_char_p, []),
    "neosr_abi_version": (C.c_int, []),
    "neosr_conv3x3": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "neosr_conv3x3_pack_bytes": (_i64, [_i32, _i32]),
    "neosr_conv3x3_pack_weights": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "neosr_debug_set_timeline": (C.c_int, [_vp]),
    "neosr_set_num_streams": (C.c_int, [C.c_int]),
    "neosr_set_xcd_aware": (C.c_int, [C.c_int]),
    "neosr_set_winograd": (C.c_int, [C.c_int]),
    "neosr_get_winograd": (C.c_int, []),
    "neosr_set_wino4_n64": (C.c_int, [C.c_int]),
    "neosr_set_wgrad4": (C.c_int, [C.c_int]),
    "neosr_set_conv_chain": (C.c_int, [C.c_int]),
    "neosr_set_fast_matmul": (C.c_int, [C.c_int]),
    "neosr_set_gemm_x3": (C.c_int, [C.c_int]),
    "neosr_set_wgrad_rrdb": (C.c_int, [C.c_int]),
    "neosr_set_conv_chain_sync": (C.c_int, [C.c_int]),
    "neosr_conv_chain_status": (C.c_int, []),
    "neosr_conv_chain_health": (C.c_int, [C.c_void_p, C.c_void_p]),
    "neosr_conv_chain_ack": (C.c_int, [C.c_void_p]),
    "neosr_debug_chain_mark_slow": (C.c_int, [C.c_void_p]),
    "neosr_conv3x3_pack_wino_bytes": (_i64, [_i32, _i32]),
    "neosr_conv3x3_pack_wino": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "neosr_conv3x3_pack_wino4_bytes": (_i64, [_i32, _i32]),
    "neosr_conv3x3_pack_wino4": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "neosr_conv3x3_pack_many": (C.c_int, [C.POINTER(PackItem), _i32, _vp]),
    "neosr_conv3x3_wgrad_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "neosr_conv3x3_wgrad": (C.c_int, [C.POINTER(WgradDesc), _vp]),
    "neosr_conv3x3_wgrad_multi_workspace_bytes": (_i64, [C.POINTER(WgradDesc), _i32]),
    "neosr_conv3x3_wgrad_multi": (C.c_int, [C.POINTER(WgradDesc), _i32, _vp, _vp]),
    "neosr_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_nhwc_to_nchw": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_pool2x2_sum": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_pool2x2_sum_masked": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "neosr_pixel_shuffle_nhwc_to_nchw": (
        C.c_int,
        [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    ),
    "neosr_pixel_unshuffle_nchw_to_nhwc": (
        C.c_int,
        [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    ),
    "neosr_prelu_dslope": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "neosr_prelu_dslope_workspace_bytes": (_i64, [_i64, _i32]),
    "neosr_fill": (C.c_int, [_vp, _i64, _f32, _vp]),
    "neosr_axpy_slice": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _vp]),
    "neosr_l1_loss_fwd": (C.c_int, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "neosr_l1_loss_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _vp, _vp]),
    "neosr_grad_norm": (C.c_int, [_vp, _i64, _f32, _vp, _vp]),
    "neosr_adamw_step": (C.c_int, [C.POINTER(AdamWDesc), _vp]),
    "neosr_filter2d": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_resize": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    "neosr_gaussian_noise": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "neosr_poisson_rate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "neosr_poisson_noise": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "neosr_poisson_sample": (C.c_int, [_vp, _vp, _i64, C.c_uint64, C.c_uint64, _vp]),
    "neosr_diffjpeg": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "neosr_quantize_u8": (C.c_int, [_vp, _vp, _i64, _vp]),
    "neosr_clamp01": (C.c_int, [_vp, _vp, _i64, _vp]),
    "neosr_crop": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_gather_rows": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp]),
    "neosr_normal_sample": (C.c_int, [_vp, _i64, C.c_uint64, C.c_uint64, _vp]),
    "neosr_blur_kernels": (C.c_int, [_vp, _i32, _vp, _vp]),
    "neosr_space_to_depth2": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_depth_to_space2_fused": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp, _i32, _i32, _i32, _i32, _vp]),
    "neosr_bilinear_up2": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_maxpool2": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "neosr_add": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "neosr_leaky_relu": (C.c_int, [_vp, _vp, _f32, _vp, _i64, _vp]),
    "neosr_norm_nchw_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "neosr_chc_loss_fwd": (C.c_int, [_vp, _vp, _i64, _f32, _i32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "neosr_chc_loss_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _i32, _f32, _f32, _f32, _vp, _i32, _vp]),
    "neosr_chc_cos_loss_fwd": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp, _vp]),
    "neosr_chc_cos_loss_bwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _vp]),
    "neosr_bce_logits_fwd": (C.c_int, [_vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp]),
    "neosr_bce_logits_bwd": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "neosr_spectral_norm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_spectral_norm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "neosr_gemm_workspace_bytes": (_i64, [C.POINTER(GemmDesc)]),
    "neosr_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "neosr_colsum": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "neosr_colsum_many_workspace_floats": (C.c_int64, [C.POINTER(ColsumItem), _i32]),
    "neosr_colsum_many": (C.c_int, [C.POINTER(ColsumItem), _i32, _vp, _vp]),
    "neosr_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "neosr_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "neosr_layernorm_bwd_res": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "neosr_window_attention_fwd": (C.c_int, [C.POINTER(WattnDesc), _vp]),
    "neosr_window_attention_bwd": (C.c_int, [C.POINTER(WattnDesc), _vp]),
    "neosr_pointwise_loss_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp, _vp]),
    "neosr_pointwise_loss_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp, _vp]),
    "neosr_resize_aa": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 12 + [_vp]),
    "neosr_box_blend": (C.c_int, [_vp, _vp, _vp, _vp] + [_i32] * 8 + [_f32, _vp]),
    "neosr_ssim_tiles": (_i64, [_i32, _i32, _i32]),
    "neosr_ssim_fwd": (C.c_int, [_vp, _vp, c_float_p, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    "neosr_ssim_bwd": (C.c_int, [_vp, _vp, _vp, c_float_p, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "neosr_avgpool2_planes": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "neosr_msssim_finalize": (C.c_int, [C.POINTER(MsssimDesc), _vp]),
    "neosr_clamp": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    "neosr_gaussian_blur_reflect": (C.c_int, [_vp, _vp, _vp, c_float_p, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_rgb_to_luma": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_rgb_to_oklab_chroma": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_cosine_dist_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _vp]),
    "neosr_cosine_dist_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _f32, _vp]),
    "neosr_gelu": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "neosr_batched_colsum": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_channel_attention_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "neosr_channel_attention_bwd": (C.c_int, [_vp] * 11 + [_i32, _i32, _i32, _vp]),
    "neosr_scale_channels_add": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_scale_channels_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "neosr_flash_window_attention_workspace_bytes": (_i64, [C.POINTER(FattnDesc)]),
    "neosr_flash_window_attention_fwd": (C.c_int, [C.POINTER(FattnDesc), _vp]),
    "neosr_flash_window_attention_bwd": (C.c_int, [C.POINTER(FattnDesc), _vp]),
    "neosr_set_fattn_fused": (C.c_int, [C.c_int]),
    "neosr_tblock_save_floats": (_i64, [C.POINTER(TBlockDesc)]),
    "neosr_tblock_bwd_workspace_floats": (_i64, [C.POINTER(TBlockDesc)]),
    "neosr_tblock_forward": (C.c_int, [C.POINTER(TBlockDesc), _vp, _vp, _vp, _vp]),
    "neosr_tblock_backward": (C.c_int, [C.POINTER(TBlockDesc), _vp, _vp, _vp, _vp, C.POINTER(TBlockGrads), _vp, _vp]),
    "neosr_set_tblock_streams": (C.c_int, [C.c_int]),
    "neosr_tblock_side_forks": (C.c_int64, []),
    "neosr_set_tblock_tail": (C.c_int, [C.c_int]),
    "neosr_tblock_tail_join": (C.c_int64, [_vp]),
    "neosr_tblock_tails": (C.c_int64, []),
    "neosr_tblock_tail_done": (C.c_int, [C.c_int64]),
    "neosr_gemm_tn_group": (C.c_int, [C.POINTER(GemmDesc), _i32, C.POINTER(C.c_int32), _vp]),
    "neosr_prelu_dslope_many": (C.c_int, [C.POINTER(DslopeItem), _i32, _vp, _i64, _i32, _i32, _i32, _vp]),
    "neosr_pixel_shuffle_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "neosr_affine": (C.c_int, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "neosr_row_scale": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "neosr_adan_sf_step": (C.c_int, [C.POINTER(AdanDesc), _vp]),
    "neosr_lerp": (C.c_int, [_vp, _vp, _i64, _f32, _vp]),
    "neosr_optim_step": (C.c_int, [C.POINTER(OptimDesc), _vp]),
    "neosr_fsam_first_step": (C.c_int, [C.POINTER(FsamDesc), _vp]),
    "neosr_prof_num_classes": (C.c_int, []),
    "neosr_prof_enable": (C.c_int, [C.c_int]),
    "neosr_prof_collect": (
        C.c_int,
        [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    ),
    "neosr_prof_collect_exec": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "neosr_prof_collect_chain": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "neosr_rrdbnet_workspace_bytes": (_i64, [C.POINTER(RRDBNetCfg)]),
    "neosr_rrdbnet_num_params": (_i32, [C.POINTER(RRDBNetCfg)]),
    "neosr_rrdbnet_forward": (C.c_int, [C.POINTER(RRDBNetCfg), c_void_pp, _vp, _vp, _vp, _vp]),
    "neosr_rrdbnet_backward": (
        C.c_int,
        [C.POINTER(RRDBNetCfg), c_void_pp, c_void_pp, _vp, _vp, _vp, _vp],
    ),
    "neosr_rrdbnet_backward_marked": (
        C.c_int,
        [C.POINTER(RRDBNetCfg), c_void_pp, c_void_pp, _vp, _vp, _vp, _vp, _i32, C.POINTER(C.c_int32), c_void_pp],
    ),
    "neosr_compact_workspace_bytes": (_i64, [C.POINTER(CompactCfg)]),
    "neosr_compact_num_params": (_i32, [C.POINTER(CompactCfg)]),
    "neosr_compact_forward": (C.c_int, [C.POINTER(CompactCfg), c_void_pp, _vp, _vp, _vp, _vp]),
    "neosr_compact_backward": (
        C.c_int,
        [C.POINTER(CompactCfg), c_void_pp, c_void_pp, _vp, _vp, _vp, _vp],
    ),
}

_lib = None

# Bumped by everything that rewrites parameters through raw device pointers (the fused optimizer steps,
# schedule-free eval()/train() lerps, SAM's climb / restore): derived per-weight data cached on the Python
# side (packed convolution weights, neosr_amd/hip/layers.py) is keyed on it.  In-place torch ops are
# covered by the tensors' own `_version`.
WEIGHTS_EPOCH = 0


def params_changed() -> None:
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


# The `fast_matmul` tier of the F(4x4,3x3) kernels (include/neosr_amd.h: neosr_set_fast_matmul).  The weight images are
# packed FOR the mode, so the mode is part of every packed-image cache key (hip/layers.py: also for frozen weights, which
# ignore the epoch) and a switch invalidates whatever was packed before it.
FAST_MATMUL = 0


def set_fast_matmul(on: bool) -> bool:
    """Switch the tier on / off for this process; returns the previous setting."""
    global FAST_MATMUL
    prev = bool(load().neosr_set_fast_matmul(1 if on else 0))
    FAST_MATMUL = 1 if on else 0
    params_changed()
    return prev


class NeosrAmdError(RuntimeError):
    pass


def lib_path() -> Path:
    return Path(os.environ.get("NEOSR_AMD_LIB", str(_LIB_PATH)))


def _sync_fast_matmul(lib) -> None:
    """(an environment default, NEOSR_AMD_FAST_MATMUL=1, is read by the library: mirror it)"""
    global FAST_MATMUL
    prev = lib.neosr_set_fast_matmul(0)
    lib.neosr_set_fast_matmul(prev)
    FAST_MATMUL = int(prev)


def load():
    """Load the shared library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        msg = (
            f"{path} not found. Build the HIP extension first: "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `bash neosr_amd/csrc/build.sh`. "
            "neosr_amd has no CPU/PyTorch fallback for its kernels."
        )
        raise NeosrAmdError(msg)
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _sync_fast_matmul(lib)
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        err = load().neosr_last_error().decode()
        raise NeosrAmdError(f"{what or 'libneosr_amd'} failed (rc={rc}): {err}")


_RAW_STREAM = None


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device.  Called once per kernel launch: it goes through
    torch's raw-stream query (~0.3 us) instead of building a `torch.cuda.Stream` object (~9 us, which was 14 ms of host
    time per hat_l step)."""
    global _RAW_STREAM
    import torch

    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)  # noqa: SLF001
    if _RAW_STREAM:
        return _RAW_STREAM(torch._C._cuda_getDevice())  # noqa: SLF001
    return torch.cuda.current_stream().cuda_stream


def require_device(t, name: str = "tensor"):
    """The product path never silently runs on the CPU."""
    import torch

    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        msg = f"neosr_amd: {name} must live on a HIP device (got {getattr(t, 'device', type(t))})"
        raise NeosrAmdError(msg)
    if t.dtype != torch.float32:
        raise NeosrAmdError(f"neosr_amd: {name} must be float32 (got {t.dtype})")
    return t


def ptr_table(tensors) -> C.Array:
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
