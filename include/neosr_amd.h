/*
 * neosr_amd.h — C ABI of libneosr_amd.so, the MI355X (gfx950 / CDNA4) kernel library behind
 * the neosr training hot path (`feed_data()` -> `optimize_parameters()`).
 *
 * The reference (muslll/neosr, 100 % Python on PyTorch) has no native layer: every device op on
 * the path is an ATen call made from a torch.nn module.  Each entry point below therefore names
 * the reference *call site* whose ATen op(s) it replaces (paths relative to /root/reference).
 * All pointers are device pointers (HBM) unless stated; `stream` is a hipStream_t passed as
 * void* (NULL = default stream).  Every function is asynchronous on `stream`, never
 * synchronises the device, never allocates, and returns 0 on success; on failure it returns
 * non-zero and neosr_last_error() describes why.
 *
 * Activation tensors are channels-last: element (b, y, x, c) of a tensor with channel stride
 * `cs` lives at ((b*H + y)*W + x)*cs + c.  A tensor pointer may point into the middle of a
 * wider pixel vector (that is how the RDB dense concatenation is expressed without torch.cat).
 * Weights stay in PyTorch's canonical (Cout, Cin, 3, 3) fp32 layout so state-dicts round-trip.
 */
#ifndef NEOSR_AMD_H
#define NEOSR_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NEOSR_ACT_NONE 0
#define NEOSR_ACT_LRELU 1 /* slope given per call (0.2 esrgan/unet, 0.01 swinir) */
#define NEOSR_ACT_RELU 2
#define NEOSR_ACT_PRELU 3 /* per-channel learnable slope (compact) */
#define NEOSR_ACT_GELU 4  /* nn.GELU() of HAT's CAB (hat_arch.py:62-66), F(4x4,3x3) kernel only: see out2 / out_mask_gelu */

#define NEOSR_CONV_FWD 0
#define NEOSR_CONV_DGRAD 1

/* library / build info ------------------------------------------------------------------ */
const char* neosr_last_error(void);
const char* neosr_build_info(void); /* "gfx950 <date> ..." */
int neosr_abi_version(void);

/*
 * 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on fp32-input MFMA
 * (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
 *
 * Replaces: nn.Conv2d(.,.,3,1,1) + LeakyReLU + residual scale-add + torch.cat in
 *   neosr/archs/esrgan_arch.py:109-116 (RDB), :137-142 (RRDB), :196-214 (esrgan.forward,
 *   incl. F.interpolate(nearest, x2) feeding conv_up1/conv_up2 via `ups`),
 *   neosr/archs/compact_arch.py:76-79 (conv + PReLU chain, PReLU applied on load via in_prelu),
 *   and their autograd backward-data (mode = NEOSR_CONV_DGRAD: the transposed/flipped kernel).
 *
 *   mode FWD  : out[p, n] = sum_{k<K, tap} in'[p+tap, k] * w[n0w+n, k, tap]         (w is (w_cout, w_cin,3,3), K == w_cin)
 *   mode DGRAD: out[p, n] = sum_{k<K, tap} in'[p-tap, k] * w[k, n, tap]             (K == w_cout, n < N <= w_cin)
 *   in' = in (optionally nearest-upsampled x2, PReLU'd on load, or multiplied by the
 *         activation derivative mask: in * (mask > 0 ? 1 : mask_slope[_c]))
 *   epilogue: v = act(acc + bias[n]);  v = v*alpha + res1 (n < res1_nch);
 *             v = v*alpha2 + res2 (n < res2_nch);  v = accumulate ? out + v : v;
 *             out = out_mask ? v * (out_mask[p, n] > 0 ? 1 : out_mask_slope) : v
 *             (out_mask = the forward activation whose derivative gates this gradient slice: lets a
 *              gather-form backward store d(loss)/d(pre-activation) directly)
 *
 * w_pack (optional): the same weights re-laid by neosr_conv3x3_pack_weights() for `mode`.  When given
 * (and in/out/res are 16-byte aligned, K % 4 == 0, no in_mask / in_prelu) the launch takes the
 * direct-to-LDS kernel: chunks of 16 reduction channels go global -> LDS with
 * global_load_lds_dwordx4 into two buffers, one barrier per chunk, 16-byte LDS fragment reads.  With
 * w_pack the reduction runs over the first K channels of the PACKED image, whose rows may concatenate
 * several convolutions (see neosr_rrdbnet_backward), so `w`, w_cout, w_cin are not consulted.
 */
typedef struct neosr_conv_desc {
  const float* in;          /* (B, Hin, Win, in_cs); Hin = ups ? H/2 : H */
  const float* in_mask;     /* optional, same geometry as `in` (never with ups) */
  const float* mask_slopes; /* optional per-k slope for the mask (PReLU); else mask_slope */
  const float* in_prelu;    /* optional per-k PReLU slope applied to `in` on load */
  const float* w;           /* canonical (w_cout, w_cin, 3, 3) */
  const float* bias;        /* optional (N) */
  const float* prelu;       /* per-n slope when act == PRELU */
  const float* res1;        /* optional */
  const float* res2;        /* optional */
  float* out;               /* (B, H, W, out_cs) */
  int32_t B, H, W;          /* output spatial size */
  int32_t K, N;
  int32_t w_cout, w_cin;
  int32_t in_cs, mask_cs, out_cs, res1_cs, res2_cs;
  int32_t res1_nch, res2_nch;
  int32_t mode, ups, act, accumulate;
  float mask_slope, slope, alpha, alpha2;
  const float* w_pack;      /* optional packed weights, see above */
  const float* out_mask;    /* optional (B, H, W, out_mask_cs), first N channels */
  int32_t out_mask_cs;
  float out_mask_slope;
  int32_t s2d_c;            /* > 0: the input (FWD: K side, DGRAD: N side) is a space-to-depth tensor with
                               s2d_c channels per sub-pixel and `w` is the 3x3 expansion of a 4x4 / stride-2
                               / pad-1 kernel (nn.Conv2d(C, N, 4, 2, 1), unet_arch.py:20-22): 20 of its 36
                               (tap, sub-pixel) blocks are structurally zero and are skipped (4 taps per
                               sub-pixel instead of 9) */
  int32_t reserved0;
  const float* w_wino;      /* optional Winograd F(2x2,3x3) image of the same weights (neosr_conv3x3_pack_wino):
                               launches that qualify for w_pack and have no ups / s2d_c / PReLU take the Winograd
                               kernel (16/36 of the multiplications; same epilogue); see neosr_set_winograd */
  const float* w_wino4;     /* optional Winograd F(4x4,3x3) image of the same weights (neosr_conv3x3_pack_wino4): under
                               neosr_set_winograd(2) (the default) launches that would take w_wino
                               take the F(4x4,3x3) kernel instead (36 multiplications per 4x4 output tile and channel
                               pair instead of 144 direct / 64 with F(2x2,3x3); same epilogue) */
  /* F(4x4,3x3) kernel only (round 4: lets conv + PReLU chains — SRVGGNetCompact, compact_arch.py:49-85 — run as plain
   * Winograd launches; a launch that sets them and does not qualify for that kernel is an error): */
  const float* out_mask_slopes; /* optional per-n slope of the out_mask derivative (PReLU'), instead of out_mask_slope */
  float* out2;                  /* optional second output (B, H, W, out2_cs): conv + bias BEFORE activation / residuals /
                                   mask — the pre-activation a PReLU layer keeps for its backward pass (forward), the
                                   unmasked gradient its slope gradient needs (backward-data) */
  int32_t out2_cs;
  int32_t out_mask_gelu;        /* 1: the out_mask derivative is GELU'(out_mask) instead of the leaky step (backward-data of the
                                   convolution BEHIND a GELU: HAT's CAB, hat_arch.py:62-66) */
} neosr_conv_desc;

int neosr_conv3x3(const neosr_conv_desc* d, void* stream);
/* Packed weight image for the direct-to-LDS kernel: [ceil(N/32)][ceil(K/16)][tap 9][k quad 4][n 32][4]
 * floats (each (n-block, chunk) slab is the 18 KB LDS image, so staging is eighteen contiguous 1 KB
 * wave loads), zero padded.  mode FWD: N = w_cout, K = w_cin, element = w[n, k, tap];
 * mode DGRAD: N = w_cin, K = w_cout, element = w[k, n, 8 - tap]. */
int64_t neosr_conv3x3_pack_bytes(int32_t N, int32_t K);
int neosr_conv3x3_pack_weights(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode,
                               float* dst, void* stream);
/* Winograd F(2x2, 3x3) image: [ceil(N/32)][ceil(K/16)][pos 16][k quad 4][n 32][4] floats, element (G g G^T)[pos] with g
 * the 3x3 kernel of (n, k) exactly as the direct image addresses it (mode FWD / DGRAD), zero padded.  The kernel
 * (conv_wino.hip) transforms the input patches in registers, so only the weights are pre-transformed.  Exact fp32
 * arithmetic in a different summation order: results differ from the direct kernel by ~1e-6 relative.
 * neosr_set_winograd(0) (env NEOSR_AMD_WINOGRAD=0) makes every launch ignore w_wino; returns the previous setting. */
int64_t neosr_conv3x3_pack_wino_bytes(int32_t N, int32_t K);
int neosr_conv3x3_pack_wino(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode, float* dst, void* stream);
int neosr_set_winograd(int on);
int neosr_get_winograd(void);   /* the current mode (0 / 1 / 2), see below */
/* Winograd F(4x4, 3x3) image (conv_wino4.hip): [ceil(N/32)][ceil(K/32)][pos 36][k parity 2][cout block 2][k quad 4]
 * [cout 16][4] floats, element (G g G^T)[pos] (6x6, points 0, +-1, +-2, inf; evaluated in float64 and rounded once) of the
 * (n, k) pair the direct image addresses (mode FWD / DGRAD), zero padded.  fp32 products and sums; the larger transforms
 * amplify rounding ~10x more than F(2x2,3x3): ~5e-6 of the output scale against a float64 convolution.
 * neosr_set_winograd: 0 = direct kernels, 1 = F(2x2,3x3) wherever w_wino is given, 2 (default; env NEOSR_AMD_WINOGRAD)
 * = F(4x4,3x3) wherever w_wino4 is given and the launch has at least NEOSR_WINO4_MIN_WGS workgroups (16 x 16-pixel tiles x
 * 32-cout blocks; below that a 256-CU chip is better filled by the 8 x 16-pixel tiles of F(2x2,3x3)) or no w_wino was
 * given, else as 1. */
#define NEOSR_WINO4_MIN_WGS 64
/* The F(4x4,3x3) kernel has two workgroup shapes, 32 or 64 output channels (launches with N > 32), chosen per launch by a
 * fill estimate over the 256 CUs.  They sum the channel chunks in different orders (results differ by rounding, ~1e-6), so a
 * network's output bits can depend on the batch size the way a vendor library's algorithm choice does.
 * neosr_set_wino4_n64: -1 = by the estimate (default), 0 = always 32, 1 = 64 whenever N > 32; returns the previous mode. */
int neosr_set_wino4_n64(int mode);
/* Weight gradient of the 32-channel-tile path: 1 (default; env NEOSR_AMD_WGRAD4) = the F(4x4-tile) Winograd form when
 * neosr_get_winograd() == 2, 0 = the F(2x2) form.  Returns the previous setting. */
int neosr_set_wgrad4(int on);
/* The RRDB trunk (neosr/archs/esrgan_arch.py:82-142 and its backward-data pass) as ONE launch per RRDB and direction
 * (conv_wino4_chain.hip: fifteen F(4x4,3x3) layers, one persistent workgroup per 16 x 16-pixel tile, tile-to-tile hand-off
 * through flag words) instead of one launch per convolution: 1 (default; env NEOSR_AMD_CHAIN) = wherever the trunk takes
 * the F(4x4,3x3) kernel, the batch has at most one tile per CU and neosr_set_wino4_n64 is not 0; 0 = never.  Same
 * arithmetic per layer: bit-identical to the per-convolution launches of the same workgroup shapes.  Returns the previous
 * setting.  neosr_conv_chain_status: 0 while no flag wait ever ran into its spin bound on this device (it synchronises;
 * a non-zero value means a chain launch did not get all its workgroups resident: its
 * results are invalid, and from this read on the library uses one launch per convolution in this process).
 * neosr_set_conv_chain_sync(0) skips the flag waits (timing experiments only: results are then racy); default 1.
 * neosr_conv_chain_health(dst, stream): enqueues a copy of the device's two health words into dst[0..1] (floats, device
 * memory) without synchronising — dst[0] = 1 when a flag wait lasted about a millisecond or more since the last neosr_conv_chain_ack
 * (results valid; the mark stays until neosr_conv_chain_ack), dst[1] = the sticky abort word neosr_conv_chain_status returns.  The
 * models reduce them over the ranks with their loss scalars (neosr/models/base.py:498-526) so that all ranks leave the
 * chain launches — or stop — at the same iteration. */
int neosr_set_conv_chain(int on);
/* The `fast_matmul` tier of the F(4x4,3x3) forward / backward-data kernels (reference: neosr/train.py:168-173 enables TF32
 * convolutions and "medium" matmul precision; neosr/models/image.py:117-127 autocast): 1 = every fp32 operand of the
 * Winograd-domain products as two bf16 pieces (hi + lo: 16 significant bits; TF32 has 11), all four cross products on
 * v_mfma_f32_16x16x32_bf16, fp32 accumulation; transforms, epilogues, weight gradients, every other kernel and all storage
 * stay fp32.  ~1e-4 of the output scale per layer against the float64 convolution (fp32 path: ~5e-6) — a labelled
 * reduced-precision tier, never the default (0; env NEOSR_AMD_FAST_MATMUL=1).  The weight images are packed for the mode
 * (same size): switch BEFORE packing / re-pack after a switch.  The bf16x3 nn.Linear GEMMs (neosr_gemm, neosr_gemm_tn_group)
 * follow the same switch: they keep the three leading cross terms p0q0 + p0q1 + p1q0 of their six (a product good to ~2e-5
 * instead of 2^-24), half the matrix work.  Returns the previous setting. */
int neosr_set_fast_matmul(int on);
/* nn.Linear GEMMs (neosr/archs/swinir_arch.py:15-38, 139-143; hat_arch.py): 1 (default; env NEOSR_AMD_GEMM_X3) = products on
 * v_mfma_f32_32x32x16_bf16 from bf16x3 operands (each fp32 value as three bf16 pieces, the six leading cross products, fp32
 * accumulation: fp32-faithful, ~2^-24 per product); 0 = v_mfma_f32_32x32x2_f32.  Returns the previous setting. */
int neosr_set_gemm_x3(int on);
/* Behind a chain launch the fifteen weight gradients of an RRDB run as ONE neosr_conv3x3_wgrad_multi launch (1, default;
 * env NEOSR_AMD_WGRAD_RRDB) or as one launch per RDB (0): another split of the pixel range, i.e. another summation
 * order (~1e-7 relative).  Returns the previous setting. */
int neosr_set_wgrad_rrdb(int on);
int neosr_set_conv_chain_sync(int mode);
int neosr_conv_chain_status(void);
int neosr_conv_chain_health(float* dst, void* stream);
int neosr_conv_chain_ack(void* stream);
int neosr_debug_chain_mark_slow(void* stream);   /* tests: the mark a slow flag wait leaves */
int64_t neosr_conv3x3_pack_wino4_bytes(int32_t N, int32_t K);
int neosr_conv3x3_pack_wino4(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode, float* dst, void* stream);
/* Both images of MANY weight tensors in ceil(n / 24) launches per image kind (the per-layer calls above cost one launch
 * each: a transformer generator with ~160 convolutions re-packs 4 images per layer after every optimizer step).
 * `items` is a HOST array; kind 0 = direct image (neosr_conv3x3_pack_weights), 1 = Winograd F(2x2,3x3) image,
 * 2 = Winograd F(4x4,3x3) image. */
typedef struct neosr_pack_item {
  const float* w;
  float* dst;
  int32_t w_cout, w_cin, mode, kind;
} neosr_pack_item;
int neosr_conv3x3_pack_many(const neosr_pack_item* items, int32_t n, void* stream);
/* Debug aid: device buffer (4 x 64 uint64) that NEOSR_TIMELINE builds of the conv kernel fill with
 * per-wave clock stamps of workgroup 0; NULL (default) disables.  No effect in normal builds. */
int neosr_debug_set_timeline(void* dev_buf);

/*
 * Weight/bias gradient of the same convolution (autograd's convolution_backward, weight part):
 *   dw[n, k, tap] (+)= scale * sum_p g'[p, n] * in'[p+tap, k],  db[n] (+)= scale * sum_p g'[p, n]
 * g' = g * (mask > 0 ? 1 : slope) when g_mask is given (LeakyReLU/PReLU derivative).
 * Two-stage, fixed-order reduction (no float atomics): run-to-run deterministic.
 * `workspace` must hold neosr_conv3x3_wgrad_workspace_bytes() bytes.
 * Replaces the backward of the call sites listed for neosr_conv3x3.
 */
typedef struct neosr_wgrad_desc {
  const float* in;          /* (B, Hin, Win, in_cs) forward input of the conv */
  const float* in_prelu;    /* optional per-k PReLU slope applied to `in` on load */
  const float* g;           /* (B, H, W, g_cs) gradient wrt the conv output (post-activation if g_mask) */
  const float* g_mask;      /* optional activation tensor for the derivative mask */
  const float* mask_slopes; /* optional per-n slopes */
  float* dw;                /* canonical (N, K, 3, 3) */
  float* db;                /* optional (N) */
  float* workspace;
  int32_t B, H, W, K, N;
  int32_t in_cs, g_cs, mask_cs;
  int32_t ups, accumulate;
  float mask_slope, scale;
  int32_t s2d_c;            /* as in neosr_conv_desc (K side): only the 4 live taps per sub-pixel are computed */
  int32_t reserved0;
} neosr_wgrad_desc;

int64_t neosr_conv3x3_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t K, int32_t N);
int neosr_conv3x3_wgrad(const neosr_wgrad_desc* d, void* stream);
/* Up to NEOSR_WGRAD_MAX convolutions that share (B, H, W, ups) in ONE launch — e.g. the five
 * convs of a Residual Dense Block (esrgan_arch.py:109-116), or the fifteen of an RRDB — so every workgroup gets a long
 * pixel strip and the chip is filled by a single resident round.  `workspace` (shared) must
 * hold neosr_conv3x3_wgrad_multi_workspace_bytes(); the per-descriptor workspace field is unused. */
#define NEOSR_WGRAD_MAX 16
int64_t neosr_conv3x3_wgrad_multi_workspace_bytes(const neosr_wgrad_desc* descs, int32_t n);
int neosr_conv3x3_wgrad_multi(const neosr_wgrad_desc* descs, int32_t n, float* workspace,
                              void* stream);

/* layout / index kernels ----------------------------------------------------------------- */
/* (B,C,H,W) planar -> channels-last slice; replaces the implicit NCHW contract of every arch
 * forward (esrgan_arch.py:196, compact_arch.py:76). */
int neosr_nchw_to_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                       int32_t out_cs, void* stream);
int neosr_nhwc_to_nchw(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                       int32_t in_cs, void* stream);
/* backward of F.interpolate(nearest, x2): 2x2 sum pool (esrgan_arch.py:207-212). in is
 * (B,2H,2W,in_cs), out (B,H,W,out_cs); out = accumulate ? out + pooled : pooled. */
int neosr_pool2x2_sum(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                      int32_t in_cs, int32_t out_cs, int32_t accumulate, void* stream);
/* the same pool followed by the derivative of the LeakyReLU that produced the upsampled map: out = pooled *
 * (mask > 0 ? 1 : mask_slope), mask (B,H,W,mask_cs) = that activation's output (esrgan_arch.py:207-212 backward:
 * lets the gradient leave as dL/d(pre-activation), so the conv below needs no mask on load). */
int neosr_pool2x2_sum_masked(const float* in, float* out, const float* mask, int32_t B, int32_t H, int32_t W,
                             int32_t C, int32_t in_cs, int32_t out_cs, int32_t mask_cs, float mask_slope,
                             void* stream);
/* nn.PixelShuffle(r) fused with compact's `out += F.interpolate(x, nearest, r)`
 * (compact_arch.py:81-85): out[b,c,h*r+i,w*r+j] = in[b,h,w,c*r*r+i*r+j] (+ base[b,c,h,w]).
 * in channels-last (B,H,W,in_cs), out/base planar NCHW.  Pure index math: bit-exact. */
int neosr_pixel_shuffle_nhwc_to_nchw(const float* in, const float* base, float* out, int32_t B,
                                     int32_t C, int32_t H, int32_t W, int32_t r, int32_t in_cs,
                                     void* stream);
/* its adjoint: gin[b,h,w,c*r*r+i*r+j] = gout[b,c,h*r+i,w*r+j]. */
int neosr_pixel_unshuffle_nchw_to_nhwc(const float* gout, float* gin, int32_t B, int32_t C,
                                       int32_t H, int32_t W, int32_t r, int32_t gin_cs,
                                       void* stream);
/* PReLU slope gradient: dslope[c] (+)= sum_p dA[p,c] * min(z[p,c], 0)  (compact_arch.py:58-69). */
int neosr_prelu_dslope(const float* dA, const float* z, float* dslope, float* workspace,
                       int64_t npix, int32_t C, int32_t da_cs, int32_t z_cs, int32_t accumulate,
                       void* stream);
int64_t neosr_prelu_dslope_workspace_bytes(int64_t npix, int32_t C);
/* The same for n (dA, z, dslope) triples of one geometry in two launches (the PReLU layers of SRVGGNetCompact,
 * compact_arch.py:49-72, at the end of its backward pass); workspace >= n * neosr_prelu_dslope_workspace_bytes bytes.
 * Same sums in the same order as n calls of neosr_prelu_dslope (accumulate = 0). */
typedef struct neosr_dslope_item {
  const float* dA;
  const float* z;
  float* dslope;
} neosr_dslope_item;
int neosr_prelu_dslope_many(const neosr_dslope_item* items, int32_t n, float* workspace, int64_t npix, int32_t C,
                            int32_t da_cs, int32_t z_cs, void* stream);

/* p[0..n) = v ; out[p,c] += alpha*in[p,c] over a channels-last slice (skip-connection grads:
 * esrgan_arch.py:205 `feat = feat + body_feat`). */
int neosr_fill(float* p, int64_t n, float v, void* stream);
int neosr_axpy_slice(float* out, const float* in, int64_t npix, int32_t C, int32_t out_cs,
                     int32_t in_cs, float alpha, void* stream);

/* losses ---------------------------------------------------------------------------------- */
/* L1Loss(reduction="mean") * loss_weight  (neosr/losses/basic_loss.py:10-11,24-53).
 * Two-stage fixed-order reduction; loss_out is one device float.  workspace >= 4096 floats. */
int neosr_l1_loss_fwd(const float* pred, const float* target, int64_t n, float loss_weight,
                      float* loss_out, float* workspace, void* stream);
/* d loss / d pred = sign(pred-target) * loss_weight / n * (*grad_out)  (grad_out: device scalar) */
int neosr_l1_loss_bwd(const float* pred, const float* target, const float* grad_out, int64_t n,
                      float loss_weight, float* grad_pred, void* stream);

/* optimizer step ---------------------------------------------------------------------------- */
/* clip_grad_norm_(params, max_norm) (neosr/models/image.py:533-544) + torch.optim.AdamW.step
 * (base.py:151-172 "adamw") + AveragedModel EMA update (image.py:661-662), fused over a flat
 * fp32 parameter arena: one norm pass, one update pass.
 *   norm_ws: >= 4100 floats.  ws[0] receives the total grad norm.
 *   max_norm <= 0 disables clipping.  ema may be NULL.  ema_decay<0: copy (first update).
 * lr / step are passed by value: hyper-parameters live on the host in the reference too. */
typedef struct neosr_adamw_desc {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* ema; /* optional */
  float* norm_ws;
  int64_t n;
  float lr, beta1, beta2, eps, weight_decay;
  float max_norm;
  float ema_decay;
  float grad_scale; /* multiply grads (e.g. 1/world_size after a sum all-reduce) */
  int32_t step;     /* 1-based step count for bias correction */
} neosr_adamw_desc;
int neosr_grad_norm(const float* grad, int64_t n, float grad_scale, float* norm_ws, void* stream);
int neosr_adamw_step(const neosr_adamw_desc* d, void* stream);

/* on-the-fly degradation bank (otf.feed_data) ------------------------------------------------
 * Images are planar (B, C, H, W) fp32 in [0,1], exactly the tensors neosr/models/otf.py:92-283
 * moves through its 2nd-order Real-ESRGAN pipeline.  Random DRAWS (noise fields, Poisson counts,
 * qualities, scales, modes, offsets, permutations) are inputs: the host draws them (device RNG in
 * production, captured reference draws in the parity tests) and the kernels are the deterministic
 * functions of (image, draws). */
#define NEOSR_RESIZE_AREA 0
#define NEOSR_RESIZE_BILINEAR 1
#define NEOSR_RESIZE_BICUBIC 2
/* filter2D (neosr/utils/diffjpeg.py:558-584): reflect-pad k//2 then cross-correlate every plane
 * of sample b with kernel[b] (k x k, k odd <= 21); kernel_batched = 0 uses kernel[0] for all. */
int neosr_filter2d(const float* img, const float* kernel, float* out, int32_t B, int32_t C,
                   int32_t H, int32_t W, int32_t k, int32_t kernel_batched, void* stream);
/* F.interpolate(mode = area | bilinear | bicubic, align_corners=False, antialias=False)
 * (otf.py:126,179-186,222-226,243-247).  rs_h / rs_w = source-coordinate scale: 1/scale_factor
 * for the scale_factor form, in/out for the size form (ignored by area = adaptive_avg_pool2d). */
int neosr_resize(const float* in, float* out, int32_t planes, int32_t Hin, int32_t Win,
                 int32_t Hout, int32_t Wout, int32_t mode, float rs_h, float rs_w, void* stream);
/* add_gaussian_noise_pt (degradations.py:569-624), clip=True rounds=False:
 * out = clamp(img + n*(sigma_b/255)*(1-g_b) + n_gray*(sigma_b/255)*g_b, 0, 1); n (B,C,H,W) and the
 * single (H,W) field n_gray (may be NULL) are standard normal draws; sigma, gray are (B). */
int neosr_gaussian_noise(const float* img, const float* noise, const float* noise_gray,
                         const float* sigma, const float* gray, float* out, int32_t B, int32_t C,
                         int32_t H, int32_t W, void* stream);
/* generate_poisson_noise_pt part 1 (degradations.py:762-781): per-sample count of distinct 8-bit
 * levels (256-bit presence bitmap, integer atomics: deterministic, no sort, no host sync) ->
 * vals[b] = 2^ceil(log2(count)); rate = clamp(round(x*255),0,255)/255 * vals[b] where x is the
 * image (gray=0, rate (B,3,H,W)) or rgb_to_grayscale(image) (gray=1, rate (B,1,H,W)).
 * levels_ws: B*8 uint32 scratch. */
int neosr_poisson_rate(const float* img, uint32_t* levels_ws, float* vals, float* rate, int32_t B,
                       int32_t H, int32_t W, int32_t gray, void* stream);
/* part 2 (degradations.py:768-786,820-828): given P ~ Poisson(rate) (and P_gray, optional):
 * out = clamp(img + ((P/vals - q(img))*(1-g) + (P_gray/vals_gray - q(gray(img)))*g)*scale, 0, 1). */
int neosr_poisson_noise(const float* img, const float* P, const float* vals, const float* P_gray,
                        const float* vals_gray, const float* scale, const float* gray, float* out,
                        int32_t B, int32_t H, int32_t W, void* stream);
/* P ~ Poisson(rate) elementwise — the `torch.poisson(rate)` draw of generate_poisson_noise_pt
 * (degradations.py:782-785) — from a counter-based Philox4x32-10 stream keyed by (seed, offset, element):
 * Knuth's product method below rate 10, Hoermann's PTRS rejection above (exact in distribution). */
int neosr_poisson_sample(const float* rate, float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* DiffJPEG(differentiable=False)(x, quality) (neosr/utils/diffjpeg.py:514-555): pad to x16 with
 * zeros, RGB*255 -> YCbCr, 2x2 chroma mean, 8x8 DCT, quantise with the (transposed-as-stored)
 * tables * quality_to_factor(quality[b]) and round-half-even, dequantise, IDCT, chroma repeat,
 * -> RGB, clamp, /255, crop.  One wavefront per 16x16 MCU; one read + one write of the image. */
int neosr_diffjpeg(const float* img, const float* quality, float* out, int32_t B, int32_t H,
                   int32_t W, void* stream);
/* clamp((x*255).round(), 0, 255)/255 (otf.py:251) and clamp(x, 0, 1) (otf.py:153). */
int neosr_quantize_u8(const float* in, float* out, int64_t n, void* stream);
int neosr_clamp01(const float* in, float* out, int64_t n, void* stream);
/* paired_random_crop's tensor branch (neosr/data/transforms.py:38-131): one window for the batch. */
int neosr_crop(const float* in, float* out, int32_t planes, int32_t H, int32_t W, int32_t top,
               int32_t left, int32_t h, int32_t w, void* stream);
/* training-pair pool shuffle `queue = queue[randperm]` (otf.py:70-72): dst[i] = src[idx[i]]. */
int neosr_gather_rows(const float* src, const int64_t* idx, float* dst, int32_t nrows,
                      int64_t row_elems, void* stream);
/* Standard normal field, a pure function of (seed, offset): Philox4x32-10 + Box-Muller.  Replaces `torch.randn` in the
 * live draw stream of the degradation bank (neosr/data/degradations.py:593-598, 781-785); a replayed draw stream
 * (tests) bypasses it. */
int neosr_normal_sample(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* Blur / sinc kernels of the OTF dataset (neosr/data/otf_dataset.py:189-246; generators neosr/data/degradations.py:24-512)
 * evaluated on the device for a whole batch: params (n, 8) float64 on the device =
 * {type, k, sig_x, sig_y, theta, beta | cutoff, isotropic, -}, type 0 Gaussian, 1 generalized Gaussian, 2 plateau,
 * 3 circular low-pass (sinc), 4 pulse; out (n, 21, 21) float32, each k x k kernel normalised and zero-padded to 21.
 * The random parameters themselves are drawn on the host in the reference's order (KernelSampler). */
int neosr_blur_kernels(const double* params, int32_t n, float* out, void* stream);

/* GAN / perceptual branch: non-conv layers and losses (channels-last activations) ---------------
 * U-Net-SN discriminator (neosr/archs/unet_arch.py:9-67), VGG19 feature extractor
 * (neosr/archs/vgg_arch.py:76-199), chc and BCE losses (neosr/losses/basic_loss.py:132-219,
 * neosr/losses/gan_loss.py:45-82). */
/* space-to-depth by 2: out[b,Y,X,(dy*2+dx)*C+c] = in[b,2Y+dy,2X+dx,c] (inverse=1: the adjoint).
 * A 4x4/stride-2/pad-1 convolution (unet conv1-3) is a 3x3/s1/p1 convolution of this tensor with
 * the 16 taps scattered into a (N, 4C, 3, 3) weight, so it runs on neosr_conv3x3[_wgrad]. */
int neosr_space_to_depth2(const float* in, float* out, int32_t B, int32_t Hlo, int32_t Wlo,
                          int32_t C, int32_t inverse, void* stream);
/* Backward of neosr_space_to_depth2 fused with the two passes that follow it in the U-Net discriminator's backward
 * (unet_arch.py:36-60: x0 / x1 / x2 feed a 4x4 / stride-2 convolution AND a skip addition): out (B, 2Hlo, 2Wlo, C) =
 * (depth_to_space(g) + skip) * (y > 0 ? 1 : slope); skip and y optional (NULL: no addend / no mask).  Same expressions as
 * the separate passes (autograd's sum, then neosr_leaky_relu's derivative form): bit-identical. */
int neosr_depth_to_space2_fused(const float* g, const float* skip, const float* y, float slope, float* out,
                                int32_t B, int32_t Hlo, int32_t Wlo, int32_t C, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) (unet_arch.py:45,51,57);
 * backward=1: its adjoint in gather form (in = grad (B,2H,2W,C), out = (B,H,W,C)). */
int neosr_bilinear_up2(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t backward, void* stream);
/* nn.MaxPool2d(2, 2) (vgg_arch.py:140-146). gout == NULL: out (B,Ho,Wo,C) = pool(in (B,2Ho,2Wo,C));
 * else out (B,2Ho,2Wo,C) = gradient routed to the first maximum of each window. */
int neosr_maxpool2(const float* in, const float* gout, float* out, int32_t B, int32_t Ho, int32_t Wo,
                   int32_t C, void* stream);
int neosr_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* g == NULL: out = leaky_relu(x, slope) (slope 0 = ReLU); else out = x > 0 ? g : g*slope. */
int neosr_leaky_relu(const float* x, const float* g, float slope, float* out, int64_t n, void* stream);
/* VGG input normalisation fused with the layout change: out_nhwc = (in_nchw - mean) / std
 * (vgg_arch.py:159-173, 190-191); backward=1: in = NHWC grad, out = NCHW grad / std. */
int neosr_norm_nchw_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                         int32_t cs, float mean, float std, int32_t backward, void* stream);
/* chc_loss with loss_lambda = 0 (basic_loss.py:192-219): d = (a - b)*pre,
 * loss = w * mean(clamp(huber ? sqrt(d^2+1e-12) : |d|, clip_min, clip_max)); workspace >= 1024 floats.
 * bwd: grad_a (+)= w/n * 1[clip_min <= t <= clip_max] * dt/dd * pre * (*grad_out). */
int neosr_chc_loss_fwd(const float* a, const float* b, int64_t n, float pre, int32_t huber,
                       float clip_min, float clip_max, float loss_weight, float* loss_out,
                       float* workspace, void* stream);
int neosr_chc_loss_bwd(const float* a, const float* b, const float* grad_out, int64_t n, float pre,
                       int32_t huber, float clip_min, float clip_max, float loss_weight, float* grad_a,
                       int32_t accumulate, void* stream);
/* chc_loss with its cosine-similarity term (neosr/losses/basic_loss.py:192-219, loss_lambda != 0) on NCHW tensors
 * (N, C, hw = H*W):  c = mean over pixels of (1 - cos_sim over the C channels, nn.CosineSimilarity(dim=1, eps)),
 * loss = loss_weight * mean(clamp(t + loss_lambda * c, clip_min, clip_max)), t = |a-b| or sqrt((a-b)^2 + 1e-12).
 * aux (2 floats, device): c and the number of elements inside the clamp range, kept for the backward; workspace
 * 3072 floats.  bwd writes d loss / d a (the elementwise term where the clamp passes + the cosine term's share). */
int neosr_chc_cos_loss_fwd(const float* a, const float* b, int32_t N, int32_t C, int64_t hw, int32_t huber,
                           float clip_min, float clip_max, float loss_lambda, float loss_weight, float cos_eps,
                           float* loss_out, float* aux, float* workspace, void* stream);
int neosr_chc_cos_loss_bwd(const float* a, const float* b, const float* grad_out, int32_t N, int32_t C, int64_t hw,
                           int32_t huber, float clip_min, float clip_max, float loss_lambda, float loss_weight,
                           float cos_eps, const float* aux, float* grad_a, void* stream);
/* nn.BCEWithLogitsLoss()(x, full_like(x, target)) * loss_weight (gan_loss.py:59-82);
 * mean_out (optional) receives mean(x) (`out_d_real/out_d_fake`, image.py:566,579).
 * workspace >= 2048 floats. */
int neosr_bce_logits_fwd(const float* x, int64_t n, float target, float loss_weight, float* loss_out,
                         float* mean_out, float* workspace, void* stream);
int neosr_bce_logits_bwd(const float* x, const float* grad_out, int64_t n, float target,
                         float loss_weight, float* grad_x, void* stream);
/* torch.nn.utils.spectral_norm, n_power_iterations = 1 (unet_arch.py:21-34): W is (rows, cols)
 * = (Cout, Cin*k*k).  update_uv (train mode): v = normalize(W^T u), u = normalize(W v) in place;
 * sigma = u.(W v); w_out = W / sigma.  scratch_rows: rows floats.
 * bwd: g_orig = (gw - <gw, w> u v^T) / sigma (u, v constants); workspace >= 1028 floats. */
int neosr_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_out, float* sigma,
                            float* scratch_rows, int32_t rows, int32_t cols, int32_t update_uv,
                            float eps, void* stream);
int neosr_spectral_norm_bwd(const float* gw, const float* w, const float* u, const float* v,
                            const float* sigma, float* g_orig, float* workspace, int32_t rows,
                            int32_t cols, void* stream);

/* transformer generators (SwinIR) ---------------------------------------------------------------
 * Tokens are channels-last pixels: the (B, H*W, C) token matrix of neosr/archs/swinir_arch.py IS the
 * (B, H, W, C) activation buffer, so PatchEmbed/PatchUnEmbed (swinir_arch.py:666-765) cost nothing. */
#define NEOSR_GEMM_NT 0 /* C[M,N] = A[M,K] B[N,K]^T   nn.Linear forward  (B = weight)          */
#define NEOSR_GEMM_NN 1 /* C[M,N] = A[M,K] B[K,N]     nn.Linear backward-data (A = dY, B = W)   */
#define NEOSR_GEMM_TN 2 /* C[M,N] = A[K,M]^T B[K,N]   nn.Linear backward-weight (A = dY, B = X) */
/* fp32 MFMA GEMM for the Linear layers (swinir_arch.py:15-38 Mlp, :139-143,209 qkv/proj).
 * NT/NN epilogue, in this order: + bias[n]; aux_out = value (pre-activation kept for backward);
 * erf-form GELU (gelu=1; erf by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7, with __expf — not libm erff); * GELU'(aux_in[m,n]); * row_scale[m / rows_per_scale] (DropPath);
 * + res[m,n].  TN: fixed-order split-K through `workspace` (neosr_gemm_workspace_bytes), dense C,
 * C = accumulate ? C + result : result; row_scale[k / rows_per_scale] multiplies ROW k of A (the DropPath
 * scale of the incoming gradient dY — the weight and bias gradients of a dropped sample are zero) so that
 * no scaled copy of dY is needed: NN takes the same scale in its epilogue.  TN with accumulate == 2 skips the
 * split reduction: it returns -(number of splits) and leaves the partials [splits][M N (+ M)] in `workspace` for
 * neosr_colsum_many (colsum_a, if given, must be C + M N). */
typedef struct neosr_gemm_desc {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* res;
  const float* aux_in;
  float* aux_out;
  const float* row_scale;
  float* workspace;
  int32_t M, N, K, lda, ldb, ldc, ldres, ldaux, rows_per_scale, mode, gelu, accumulate;
  float* colsum_a; /* TN only, optional: colsum_a[m] (+)= sum_k A[k, m] — the bias gradient sum_rows dY,
                      taken from the A tiles the weight-gradient GEMM stages anyway */
} neosr_gemm_desc;
int64_t neosr_gemm_workspace_bytes(const neosr_gemm_desc* d);
int neosr_gemm(const neosr_gemm_desc* d, void* stream);
/* Up to 4 TN problems (weight gradients dW = dY^T X, each described as for neosr_gemm with its own `workspace`) in ONE
 * launch of the register-fed kernel: the four Linears of a transformer block once its data-gradient chain has run
 * (swinir_arch.py:15-38,139-143: fc2, fc1, proj, qkv).  The split reductions are left to the caller as with
 * accumulate == 2: nsplit_out[i] = rows of problem i's partial matrix [rows][M N (+ M)] for neosr_colsum_many.
 * Returns 0, an error code, or -1 when a problem does not qualify (launch them one by one then).  Same per-task
 * arithmetic as neosr_gemm: bit-identical partials. */
int neosr_gemm_tn_group(const neosr_gemm_desc* descs, int32_t n, int32_t* nsplit_out, void* stream);
/* out[c] (+)= sum_r x[r, c]  (bias gradients); workspace >= 256*cols floats. */
int neosr_colsum(const float* x, float* out, float* workspace, int32_t rows, int32_t cols, int32_t ld,
                 int32_t accumulate, void* stream);
/* The same reduction for MANY matrices in one (two-stage: two) launch per 32 jobs, each job cut into row slabs and
 * summed in exactly the order neosr_colsum uses.  None of the parameter-gradient sums of a transformer block
 * (LayerNorm dgamma / dbeta partials, relative-position-bias bins, bias gradients) is needed before the optimizer:
 * the backward pass queues them and runs this once at its end (the reference gets the same sums from ATen reductions
 * inside each op's backward: swinir_arch.py:284-297 LayerNorm, :85-137 bias table).
 * workspace >= neosr_colsum_many_workspace_floats(items, n) floats. */
typedef struct neosr_colsum_item {
  const float* x;
  float* out;
  int32_t rows, cols, ld, accumulate;
} neosr_colsum_item;
int64_t neosr_colsum_many_workspace_floats(const neosr_colsum_item* items, int32_t n);
int neosr_colsum_many(const neosr_colsum_item* items, int32_t n, float* workspace, void* stream);
/* nn.LayerNorm(C) over the last dim, eps 1e-5, biased variance (swinir_arch.py:284,297,960,1035).
 * fwd keeps (mean, rstd) per row in `stats` (2*rows floats).  bwd: dx, and dgamma/dbeta (+)= via a
 * fixed-order two-stage column reduction; workspace >= (2*1024 + 512)*C floats.  C <= 512.  One reduction launch
 * when dbeta == dgamma + C.  neosr_layernorm_bwd_res with dgamma == dbeta == NULL skips the reduction: it returns
 * -(number of partial rows) and leaves the partials [rows][2 C] (dgamma | dbeta) at the start of `workspace`. */
int neosr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                        int64_t rows, int32_t C, float eps, void* stream);
int neosr_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                        float* dx, float* dgamma, float* dbeta, float* workspace, int64_t rows,
                        int32_t C, int32_t accumulate, void* stream);
/* Same with dx += dres: the gradient that reaches x over the shortcut of a pre-norm residual block
 * (x -> x + f(norm(x)), swinir_arch.py:343-392) is added in the same pass (autograd would otherwise sum the two with an
 * elementwise kernel per norm).  dres may be NULL. */
int neosr_layernorm_bwd_res(const float* dy, const float* x, const float* stats, const float* gamma,
                            const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace,
                            int64_t rows, int32_t C, int32_t accumulate, void* stream);
/* (Shifted-)window multi-head self-attention (swinir_arch.py:150-212,343-392) on the fused
 * qkv matrix [B*H*W, 3*C] in IMAGE order: torch.roll, window_partition/reverse and the head split are
 * folded into addressing; relative-position bias is gathered from `rpb_table` ((2ws-1)^2, heads) by
 * the analytic index; the shifted-window mask (0 / -100) is computed analytically (the reference
 * rebuilds it 18x per forward).  One workgroup per (window, head); QK^T and PV on fp32 MFMA.
 * fwd: out [B*H*W, C] image order, lse [B*nW*heads*N] kept for backward.
 * bwd: dqkv [B*H*W, 3*C] (every element written once), d_rpb_table (+)= (fixed-order reduction
 * over windows; workspace >= (B*nW + 256)*heads*(2ws-1)^2 floats).  With `out` = the forward output also set, delta =
 * rowsum(dO . O) comes from it and dS is finished in the score tiles' registers (else from the P / dP tiles in a row pass).
 * The forward with head_dim <= 30 runs the wave-per-(window, head, query tile) kernel of attn_wave.hip. */
typedef struct neosr_wattn_desc {
  const float* qkv;
  const float* rpb_table;
  float* out;
  float* lse;
  const float* dout;   /* bwd */
  float* dqkv;         /* bwd */
  float* d_rpb_table;  /* bwd */
  float* workspace;    /* bwd */
  int32_t B, H, W, C, heads, ws, shift, accumulate_rpb;
  float scale;
} neosr_wattn_desc;
int neosr_window_attention_fwd(const neosr_wattn_desc* d, void* stream);
/* bwd with accumulate_rpb == 2 (here and in neosr_flash_window_attention_bwd, self-attention form): the bias-table
 * partials stay in the workspace — [B nW][bins heads] at its start here, [B nW ws^2/64][bins heads] at float offset
 * B nW heads ws^2 there — and the call returns -(rows) for neosr_colsum_many. */
int neosr_window_attention_bwd(const neosr_wattn_desc* d, void* stream);
/* Window attention of HAT on the fused qkv matrix [B*H*W, 3*C] in image order, windows of ws = 16 (hat_s / m / l) or 8:
 *   ks == ws:     (shifted-)window self-attention of HAB (hat_arch.py:168-216 inside :299-351), rel-pos
 *                 index (yi-yj+ws-1)*(2ws-1) + (xi-xj+ws-1), shift mask as in neosr_window_attention;
 *   ks == 1.5 ws: overlapping cross-attention of OCAB (hat_arch.py:445-516): queries = the ws x ws window,
 *                 keys/values = the zero-padded ks x ks window around it (nn.Unfold(ks, stride ws, pad ws/4)),
 *                 rel-pos index per calculate_rpi_oca (hat_arch.py:1035-1068) including its negative-index
 *                 wrap-around into the (ws+ks-1)^2-row table; shift must be 0.
 * Streams 64-key blocks past 64-query blocks with an online softmax (one workgroup per window, head and
 * query block).  fwd: out [B*H*W, C], lse [B*nW*heads*ws*ws].  bwd: dqkv [B*H*W, 3*C] fully written,
 * d_rpb_table (+)= ; needs `out` (forward result) and workspace of
 * neosr_flash_window_attention_workspace_bytes().  Deterministic (no float atomics). */
typedef struct neosr_fattn_desc {
  const float* qkv;
  const float* rpb_table; /* ((ws+ks-1)^2, heads) */
  float* out;
  float* lse;
  const float* dout;   /* bwd */
  float* dqkv;         /* bwd */
  float* d_rpb_table;  /* bwd */
  float* workspace;    /* bwd */
  int32_t B, H, W, C, heads, ws, ks, shift, accumulate_rpb;
  float scale;
} neosr_fattn_desc;
int64_t neosr_flash_window_attention_workspace_bytes(const neosr_fattn_desc* d);
int neosr_flash_window_attention_fwd(const neosr_fattn_desc* d, void* stream);
int neosr_flash_window_attention_bwd(const neosr_fattn_desc* d, void* stream);
/* Backward of the self-attention form (ks == ws): 1 (default; env NEOSR_AMD_FATTN_FUSED) = ONE pass per (window, head)
 * — every score tile is formed once and feeds dQ, dK and dV (5 products per tile) —, 0 = the two recompute passes
 * (dQ per query block, dK / dV per key block: 7 products) the overlapping form always takes.  Same accumulation order:
 * bit-identical results.  Returns the previous setting. */
int neosr_set_fattn_fused(int on);
/* MSELoss / HuberLoss with reduction = "mean" (neosr/losses/basic_loss.py:57-127: F.mse_loss,
 * F.huber_loss(delta)): loss = loss_weight * mean(term(pred - target)); workspace >= 1024 floats;
 * bwd: grad_pred = grad_out[0] * loss_weight / n * term'(pred - target). */
#define NEOSR_LOSS_MSE 1
#define NEOSR_LOSS_HUBER 2
int neosr_pointwise_loss_fwd(const float* pred, const float* target, int64_t n, int32_t kind, float delta,
                             float loss_weight, float* loss_out, float* workspace, void* stream);
int neosr_pointwise_loss_bwd(const float* pred, const float* target, const float* grad_out, int64_t n,
                             int32_t kind, float delta, float loss_weight, float* grad_pred, void* stream);

/* batch augmentations (neosr/data/augmentations.py) --------------------------------------
 * F.interpolate(mode = bilinear | bicubic, antialias=True) on planar NCHW data with ATen's window /
 * weights (support = interp/2 * max(scale, 1), bicubic a = -0.5, normalised), horizontal then vertical
 * pass through `tmp` (>= B*C*Hin*Wout floats); the result lands in the box (y0, x0) of an (Hfull, Wfull)
 * output image, optionally clamped to [0, 1] and optionally reading batch entry perm[b] (resizemix,
 * augmentations.py:150-170).  Used for the x scale up / down round trip of apply_augment (:258-308). */
int neosr_resize_aa(const float* in, float* out, float* tmp, const int32_t* perm, int32_t B, int32_t C,
                    int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t Hfull, int32_t Wfull, int32_t y0,
                    int32_t x0, int32_t mode, int32_t clamp01, void* stream);
/* out = inside rows [y0,y1) x cols [x0,x1) ? lam * x[b] + (1-lam) * src[perm[b]] : x[b]  (mixup: whole
 * image; cutmix / cutblur: lam = 0; perm optional).  NCHW. */
int neosr_box_blend(const float* x, const float* src, const int32_t* perm, float* out, int32_t B, int32_t C,
                    int32_t H, int32_t W, int32_t y0, int32_t y1, int32_t x0, int32_t x1, float lam, void* stream);

/* multi-scale SSIM loss (neosr/losses/ssim_loss.py:11-163) on planar NCHW data -------------
 * Per scale: neosr_ssim_fwd applies the separable 11-tap Gaussian window (host array `window11`, zero
 * padding 5) to x, y, x^2, y^2, xy in one pass and writes per-tile partial sums [tile][{cs, ssim}]
 * (neosr_ssim_tiles(P,H,W) tiles) and, when `dmaps` is given, the three derivative maps
 * d map/d(G*x), d map/d(G*x^2), d map/d(G*xy) (3*P*H*W floats; map = ssim if want_ssim else cs).
 * neosr_avgpool2_planes is the F.avg_pool2d(2, 2) between scales.  neosr_msssim_finalize reduces the
 * partials in a fixed order to loss = loss_weight * (1 - prod_i m_i^w_i) (m = mean cs, mean ssim on the last
 * scale) and the per-scale pixel gradients gscal[i] = -loss_weight w_i prod / m_i / npix_i (device scalars, no
 * host sync).  neosr_ssim_bwd: dx = gscal * gout * (G*D0 + 2 x G*D1 + y G*D2) + 0.25 * coarse[y/2, x/2]
 * (the gradient arriving through the average pool from the next scale; optional). */
#define NEOSR_MSSSIM_SCALES 5
typedef struct neosr_msssim_desc {
  const float* partial[NEOSR_MSSSIM_SCALES];
  int64_t npix[NEOSR_MSSSIM_SCALES];
  int32_t nblk[NEOSR_MSSSIM_SCALES];
  float weights[NEOSR_MSSSIM_SCALES];
  float loss_weight;
  int32_t nscales;
  float* loss;   /* 1 float */
  float* gscal;  /* nscales floats */
} neosr_msssim_desc;
int64_t neosr_ssim_tiles(int32_t P, int32_t H, int32_t W);
int neosr_ssim_fwd(const float* x, const float* y, const float* window11, float* dmaps, float* partial, int32_t P,
                   int32_t H, int32_t W, float C1, float C2, int32_t want_ssim, void* stream);
int neosr_ssim_bwd(const float* dmaps, const float* x, const float* y, const float* window11, const float* gscal,
                   const float* gout, const float* coarse, float* dx, int32_t P, int32_t H, int32_t W, void* stream);
int neosr_avgpool2_planes(const float* in, float* out, int32_t P, int32_t H, int32_t W, void* stream);
int neosr_msssim_finalize(const neosr_msssim_desc* d, void* stream);

/* consistency_loss pieces (neosr/losses/consistency_loss.py:14-192), planar NCHW ---------
 * out = g ? g * [lo <= x <= hi] : clamp(x, lo, hi)   (torch.clamp and its backward). */
int neosr_clamp(const float* x, const float* g, float* out, int64_t n, float lo, float hi, void* stream);
/* torchvision GaussianBlur: separable `taps` (host array, odd count <= 31) with reflect padding on P planes
 * of H x W (consistency_loss.py:39,144-145); adjoint = 1: the exact transpose (pad gradients fold back).
 * tmp >= P*H*W floats. */
#define NEOSR_BLUR_MAX_TAPS 31
int neosr_gaussian_blur_reflect(const float* in, float* out, float* tmp, const float* taps, int32_t ntaps,
                                int32_t P, int32_t H, int32_t W, int32_t adjoint, void* stream);
/* sRGB (B,3,H,W) -> clamp(CIE L* / 100, 0, 1) * mul as (B,H,W) (consistency_loss.py:106-133);
 * with g (B,H,W): out = gradient wrt rgb (B,3,H,W). */
int neosr_rgb_to_luma(const float* rgb, const float* g, float* out, int32_t B, int32_t H, int32_t W, float mul,
                      void* stream);
/* sRGB (B,3,H,W) -> clamp(Oklab (a, b) * mul + 0.5, 0, 1) as (B,2,H,W) (consistency_loss.py:63-104,159-165);
 * with g (B,2,H,W): out = gradient wrt rgb. */
int neosr_rgb_to_oklab_chroma(const float* rgb, const float* g, float* out, int32_t B, int32_t H, int32_t W,
                              float mul, void* stream);
/* out[0] = 1 - mean cos(a, b) over `groups * inner` vectors of length L and stride `inner`
 * (nn.CosineSimilarity(dim=1) on (groups, L, inner) data; consistency_loss.py:176-178).  stats: 3 floats per
 * vector kept for backward; partial >= 1024 floats.  bwd: da = gout[0] * d out / d a. */
int neosr_cosine_dist_fwd(const float* a, const float* b, float* stats, float* partial, float* out, int64_t groups,
                          int32_t L, int64_t inner, float eps, void* stream);
int neosr_cosine_dist_bwd(const float* a, const float* b, const float* stats, const float* gout, float* da,
                          int64_t groups, int32_t L, int64_t inner, float eps, void* stream);

/* HAT Channel Attention Block, non-conv parts (hat_arch.py:15-52) ------------------------------------
 * erf-form GELU (same erf approximation as neosr_gemm, |err| <= 1.5e-7) between the two convs: out = g ? g * GELU'(x) : GELU(x). */
int neosr_gelu(const float* x, const float* g, float* out, int64_t n, void* stream);
/* out[b, c] = scale * sum_r x[b, r, c] (* y[b, r, c] if y): AdaptiveAvgPool2d(1) on channels-last data
 * and the gate gradient; fixed-order two-stage (128 row slabs per sample); workspace >= B*128*cols floats. */
int neosr_batched_colsum(const float* x, const float* y, float* out, float* workspace, int32_t B, int32_t rows,
                         int32_t cols, float scale, void* stream);
/* squeeze-excite gate: hidden = relu(W1 pooled + b1) (Cs <= 16), attn = sigmoid(W2 hidden + b2);
 * W1 (Cs, C), W2 (C, Cs) = the 1x1 conv weights (hat_arch.py:27-33).  bwd: parameter gradients and
 * d pooled, samples summed in order (deterministic); C <= 512. */
int neosr_channel_attention_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                                const float* b2, float* hidden, float* attn, int32_t B, int32_t C, int32_t Cs,
                                void* stream);
int neosr_channel_attention_bwd(const float* dattn, const float* attn, const float* hidden, const float* pooled,
                                const float* w1, const float* w2, float* dpooled, float* dw1, float* db1, float* dw2,
                                float* db2, int32_t B, int32_t C, int32_t Cs, void* stream);
/* out = res + alpha * y * attn[b, c]  (ChannelAttention's x*y, hat_arch.py:36-37, fused with
 * `+ conv_x * conv_scale` of HAB.forward :347; res optional);  bwd: dy = alpha g attn + dpooled / rows. */
int neosr_scale_channels_add(const float* y, const float* attn, const float* res, float* out, int32_t B,
                             int32_t rows, int32_t C, float alpha, void* stream);
int neosr_scale_channels_bwd(const float* g, const float* attn, const float* dpooled, float* dy, int32_t B,
                             int32_t rows, int32_t C, float alpha, void* stream);
/* nn.PixelShuffle(r) on channels-last tensors (swinir_arch.py:782-783): in (B,H,W,C*r*r) ->
 * out (B,H*r,W*r,C); inverse=1: the adjoint. */
int neosr_pixel_shuffle_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                             int32_t r, int32_t inverse, void* stream);
/* out = (in + shift) * scale elementwise, NCHW<->NHWC not involved (swinir_arch.py:1041,1078). */
int neosr_affine(const float* in, float* out, int64_t n, float shift, float scale, void* stream);
/* DropPath (archs/arch_util.py:118-133): out[m, :] = in[m, :] * scale[m / rows_per_scale]. */
int neosr_row_scale(const float* in, const float* scale, float* out, int64_t rows, int32_t cols,
                    int32_t rows_per_scale, void* stream);

/* Schedule-Free Adan (neosr/optimizers/adan_sf.py:138-330, the template optimizer of
 * options/train_*_otf.toml) fused with the model-level clip_grad_norm_ (models/image.py:533-544) and
 * the EMA update (image.py:661-662): neosr_grad_norm + one sweep over flat arenas.
 *   g = grad * grad_scale * min(1, max_norm / (||grad|| + 1e-6));  npg += g  (npg := -g first at step 1)
 *   m = b1 m + (1-b1) g;  d = b2 d + (1-b2) npg;  npg = b2 npg + g;  n = b3 n + (1-b3) npg^2
 *   denom = sqrt(n)/sqrt(1-b3^t) + eps;  p *= 1 - lr wd
 *   schedule_free: p = lerp(p, z, ckp1); p -= lr (1-b1^t)(1-ckp1) m/denom; p -= lr b2/(1-b2^t) (1-ckp1) d/denom;
 *                  z -= lr g        else: p -= lr/(1-b1^t) m/denom; p -= lr b2/(1-b2^t) d/denom
 *   npg = -g;  ema as in neosr_adamw_step.  `ckp1` (the averaging weight) is host state. */
typedef struct neosr_adan_desc {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* exp_avg_diff;
  float* z;            /* schedule_free only */
  float* neg_pre_grad;
  float* ema;          /* optional */
  float* norm_ws;      /* >= 4200 floats when max_norm > 0 */
  int64_t n;
  float lr, beta1, beta2, beta3, eps, weight_decay, ckp1;
  float max_norm, ema_decay, grad_scale;
  int32_t step, first_step, schedule_free;
} neosr_adan_desc;
int neosr_adan_sf_step(const neosr_adan_desc* d, void* stream);
/* p = torch.lerp(p, end, weight) elementwise: adan_sf.train() / .eval() (adan_sf.py:112-136). */
int neosr_lerp(float* p, const float* end, int64_t n, float weight, void* stream);

/* The remaining optimizers of base.get_optimizer (neosr/models/base.py:151-172) as one fused elementwise
 * step on flat arenas, with the model-level clip (max_norm) and the EMA update fused as in
 * neosr_adamw_step.  The host computes every scalar coefficient in double (as the Python originals do):
 *   ADAM      torch.optim.Adam          s0 m, s1 v              c = b1, b2, eps, wd, lr/bc1, sqrt(bc2)
 *   NADAM     torch.optim.NAdam         s0 m, s1 v              c = b1, b2, eps, wd, lr(1-mu)/(1-muprod), sqrt(bc2),
 *                                                                   lr mu_next/(1-muprod mu_next)
 *   ADAN      optimizers/adan.py        s0 m, s1 n, s2 diff, s3 neg_pre_grad   c = b1, b2, b3, eps, lr wd,
 *                                       lr/bc1, lr b2/bc2, sqrt(bc3), no_prox;  flags bit 0 = first step
 *   ADAMW_SF  optimizers/adamw_sf.py    s0 exp_avg_sq, s1 z     c = b2, eps, decay, ckp1, lr_t, lr_t (b1 (1-ckp1) - 1)
 *   ADAMW_WIN optimizers/adamw_win.py   s0 m, s1 v, s2 x, s3 y  c = b1, b2, eps, wd, lr, bc1, sqrt(bc2), beta3, beta4;
 *                                       flags 0 = plain AdamW, 1 = win, 2 = win2 */
#define NEOSR_OPT_ADAM 1
#define NEOSR_OPT_NADAM 2
#define NEOSR_OPT_ADAN 3
#define NEOSR_OPT_ADAMW_SF 4
#define NEOSR_OPT_ADAMW_WIN 5
typedef struct neosr_optim_desc {
  float* param;
  const float* grad;
  float* s0;
  float* s1;
  float* s2;
  float* s3;
  float* ema;      /* optional */
  float* norm_ws;  /* >= 4200 floats when max_norm > 0 */
  int64_t n;
  float c[12];
  float max_norm, ema_decay, grad_scale;
  int32_t kind, flags;
} neosr_optim_desc;
int neosr_optim_step(const neosr_optim_desc* d, void* stream);

/* Friendly SAM, first half of the step (neosr/optimizers/fsam.py:36-66) on flat arenas, two launches:
 *   g' = first ? g : g - sigma * momentum;  momentum = first ? g : lmbda * momentum + (1 - lmbda) * g
 *   (g = grad * grad_scale; g' is written over grad);  norm = || (adaptive ? |w| : 1) * g' ||_2  -> norm_ws[0];
 *   old_p = w;  w += (adaptive ? w^2 : 1) * g' * rho / (norm + 1e-12).
 * The second half (fsam.py:68-79) is a copy old_p -> w followed by the base optimizer's step. */
typedef struct neosr_fsam_desc {
  float* param;
  float* grad;
  float* momentum;
  float* old_p;
  float* norm_ws; /* >= 4 + 1024 floats */
  int64_t n;
  float rho, sigma, lmbda, grad_scale;
  int32_t first, adaptive;
} neosr_fsam_desc;
int neosr_fsam_first_step(const neosr_fsam_desc* d, void* stream);

/* opt-in profiler ------------------------------------------------------------------------------
 * HIP events around every conv-class launch on the launch stream (classes: 0 / 1 forward / backward-data
 * launches of conv3x3_glds_kernel, 2 weight gradient, 3 its reduce, 4 / 5 forward / backward-data launches
 * of the staged and thin kernels, 6 / 7 / 8 nn.Linear forward / backward-data / backward-weight GEMMs, 9 / 10
 * window-attention forward / backward kernels).  Used by bench.py's roofline pass only.  collect()
 * synchronises the device and fills neosr_prof_num_classes()-element arrays. */
int neosr_prof_num_classes(void);
int neosr_prof_enable(int on);
int neosr_prof_collect(double* ms, long long* launches, double* flops, double* bytes);
/* executed[c]: FLOPs of the multiplications the class's launches really ran — a Winograd F(2x2,3x3) launch executes
 * 16/36 and an F(4x4,3x3) launch 36/144 of the direct form's (which is what flops[c] above counts);
 * by_algo[3 c + a]: launches in the direct (a = 0) / F(2x2,3x3) (1) / F(4x4,3x3) (2) form.  Call before
 * neosr_prof_collect. */
int neosr_prof_collect_exec(double* executed, long long* by_algo);
/* chain_launches[c] / chain_layers[c]: launches of the chain kernel (neosr_set_conv_chain) in class c and the layers they
 * ran; every layer counts as one launch in neosr_prof_collect's launches[c].  Call before neosr_prof_collect. */
int neosr_prof_collect_chain(long long* chain_launches, long long* chain_layers);

/* whole-network plans ------------------------------------------------------------------------ */
/*
 * RRDBNet ("esrgan", neosr/archs/esrgan_arch.py:145-214) forward and backward for scale 4 / 2 / 1
 * as one host-side plan that enqueues every kernel on `stream` (no per-layer Python, no cat).
 * params / grads: host arrays of device pointers in named_parameters() order
 *   (conv_first.weight, conv_first.bias, body.0.rdb1.conv1.weight, ..., conv_last.bias).
 * x: (B, num_in_ch, H, W) NCHW fp32 (after pixel_unshuffle for scale 1/2 the caller passes the
 * unshuffled tensor); y: (B, num_out_ch, 4H, 4W) NCHW (H, W for scale<4 handled by up_stages).
 * workspace: neosr_rrdbnet_workspace_bytes(); forward leaves the saved activations in it and
 * backward must be given the same workspace.
 */
typedef struct neosr_rrdbnet_cfg {
  int32_t B, H, W;
  int32_t num_in_ch, num_out_ch, num_feat, num_block, num_grow_ch;
  int32_t training; /* 0: inference (activations not kept) */
} neosr_rrdbnet_cfg;
int64_t neosr_rrdbnet_workspace_bytes(const neosr_rrdbnet_cfg* cfg);
int32_t neosr_rrdbnet_num_params(const neosr_rrdbnet_cfg* cfg);
int neosr_rrdbnet_forward(const neosr_rrdbnet_cfg* cfg, const float* const* params, const float* x,
                          float* y, void* workspace, void* stream);
/* gy: (B,num_out_ch,4H,4W) NCHW. grads[i] written (not accumulated). gx optional (B,C,H,W). */
int neosr_rrdbnet_backward(const neosr_rrdbnet_cfg* cfg, const float* const* params,
                           float* const* grads, const float* gy, float* gx, void* workspace,
                           void* stream);
/* The same backward with gradient marks for the data-parallel path (the reference gets this from
 * DistributedDataParallel's bucketed all-reduce during backward, neosr/models/base.py:140-146): backward walks
 * the parameters from the end of the arena to its start, so after RRDB `mark_block[i]` the gradients of
 * body.<mark_block[i]>.* and of every parameter behind it (later RRDBs, conv_body .. conv_last) are final.
 * mark_event[i] is a caller-owned hipEvent_t that the plan records at that point on the stream that writes the
 * weight gradients; the caller makes its communication stream wait on it and reduces that suffix of the flat
 * gradient arena while the earlier RRDBs are still in backward.  n_marks = 0 is neosr_rrdbnet_backward. */
int neosr_rrdbnet_backward_marked(const neosr_rrdbnet_cfg* cfg, const float* const* params,
                                  float* const* grads, const float* gy, float* gx, void* workspace,
                                  void* stream, int32_t n_marks, const int32_t* mark_block,
                                  void* const* mark_event);
/* The RRDB trunk runs the two halves of the batch as independent launch chains on the caller's stream
 * and one internal stream (fork / join by events; results do not depend on the setting).  n = 1
 * keeps everything on the caller's stream (used for per-kernel timing), n = 2 (default) .. 4 cuts the batch into n
 * groups of samples; returns the previous n. */
int neosr_set_num_streams(int n);
/* XCD-aware workgroup order of the 3x3 conv and weight-gradient kernels (each XCD = 32 CUs with a private L2 takes a
 * contiguous band of pixel tiles / whole pixel splits, so halo rows and the tiles shared by several (cout, cin)
 * pairs are fetched into ONE L2): 1 = on (default; env NEOSR_AMD_XCD=0 turns it off), returns the previous
 * setting.  A pure scheduling choice: results are bit-identical either way. */
int neosr_set_xcd_aware(int on);

/*
 * SRVGGNetCompact ("compact", neosr/archs/compact_arch.py:11-85), act_type prelu|relu|leakyrelu.
 * params order: body.0.weight, body.0.bias, body.1.weight (PReLU), body.2.weight, ...
 */
typedef struct neosr_compact_cfg {
  int32_t B, H, W;
  int32_t num_in_ch, num_out_ch, num_feat, num_conv, upscale;
  int32_t act_type; /* NEOSR_ACT_PRELU | NEOSR_ACT_RELU | NEOSR_ACT_LRELU(0.1) */
  int32_t training;
} neosr_compact_cfg;
int64_t neosr_compact_workspace_bytes(const neosr_compact_cfg* cfg);
int32_t neosr_compact_num_params(const neosr_compact_cfg* cfg);
int neosr_compact_forward(const neosr_compact_cfg* cfg, const float* const* params, const float* x,
                          float* y, void* workspace, void* stream);
int neosr_compact_backward(const neosr_compact_cfg* cfg, const float* const* params,
                           float* const* grads, const float* gy, float* gx, void* workspace,
                           void* stream);

/*
 * One transformer block of the SwinIR / HAT generators as ONE call per direction (csrc/blocks.hip):
 *   attn = 0: SwinTransformerBlock (neosr/archs/swinir_arch.py:231-392) — norm1, qkv, neosr_window_attention (ws = 8),
 *             proj + DropPath + shortcut, norm2, fc1 + GELU, fc2 + DropPath + shortcut;
 *   attn = 1, ks = 1.5 ws: OCAB (neosr/archs/hat_arch.py:393-515), the same chain around the overlapping cross-attention;
 *   attn = 1, ks = ws, cab_mid > 0: HAB (neosr/archs/hat_arch.py:218-350) — the chain around the (shifted-)window
 *             self-attention plus `conv_scale * CAB(norm1(x))` (conv3x3 -> GELU -> conv3x3 -> channel attention,
 *             hat_arch.py:15-52) added in front of norm2.
 * Replaces the ~40 ATen dispatches per block and direction of the reference (and the 6-9 / 12-20 per-op calls of this
 * library's Python fronts) — the plans enqueue the same kernels with the same descriptors, bit-identical results.
 * x, out, dout, dx: (B, H, W, C) channels-last = the (B, H W, C) token matrix.  `save` (neosr_tblock_save_floats floats,
 * written by forward, read by backward) keeps the activations; `workspace` (neosr_tblock_bwd_workspace_floats) holds
 * the temporaries of backward.  drop_scale / drop_scale2: per-sample DropPath scales (B floats: 0 or 1 / keep_prob) or NULL.
 * Parameters in canonical torch layouts.  HAB only: the four packed images of the two CAB convolutions the conv
 * kernels would be given (c0 / c2, forward and backward-data: w_pack always, exactly one of w_wino / w_wino4 as
 * neosr_conv3x3 would pick for the geometry).
 */
typedef struct neosr_tblock_desc {
  int32_t B, H, W, C, heads, ws, ks, shift, hidden;
  int32_t attn;              /* 0: neosr_window_attention; 1: neosr_flash_window_attention */
  int32_t cab_mid, cab_sq;   /* HAB: channels of the CAB's first conv / of the squeeze bottleneck; 0, 0: no CAB */
  float scale, eps1, eps2, conv_scale;
  const float *n1_w, *n1_b, *rpb, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2_w, *n2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  const float *c0_w, *c0_b, *c2_w, *c2_b, *ca1_w, *ca1_b, *ca2_w, *ca2_b;   /* CAB (HAB only) */
  const float *c0_pack_f, *c0_wino_f, *c0_wino4_f, *c0_pack_d, *c0_wino_d, *c0_wino4_d;
  const float *c2_pack_f, *c2_wino_f, *c2_wino4_f, *c2_pack_d, *c2_wino_d, *c2_wino4_d;
  const float* drop_scale;    /* DropPath scales of the attention branch (proj) ... */
  const float* drop_scale2;   /* ... and of the MLP branch: two independent draws in the reference (swinir_arch.py:387,390) */
} neosr_tblock_desc;
/* Gradient targets of backward.  Each (weight, bias) and (gamma, beta) pair must be contiguous — as they are in a
 * buffer laid out like the block's slice of the network's flat parameter arena (named_parameters order, tensor starts
 * rounded up to 4 floats): one fixed-order column-sum job then finishes both tensors of a pair. */
typedef struct neosr_tblock_grads {
  float *n1_w, *n1_b, *rpb, *qkv_w, *qkv_b, *proj_w, *proj_b, *n2_w, *n2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  float *c0_w, *c0_b, *c2_w, *c2_b, *ca1_w, *ca1_b, *ca2_w, *ca2_b;
} neosr_tblock_grads;
int64_t neosr_tblock_save_floats(const neosr_tblock_desc* d);
int64_t neosr_tblock_bwd_workspace_floats(const neosr_tblock_desc* d);
int neosr_tblock_forward(const neosr_tblock_desc* d, const float* x, float* out, float* save, void* stream);
int neosr_tblock_backward(const neosr_tblock_desc* d, const float* x, const float* dout, const float* save, float* dx,
                          const neosr_tblock_grads* grads, float* workspace, void* stream);
/* Library-owned side stream of the block plans (forked / joined with events INSIDE a call: when a call returns, everything
 * it enqueued is ordered in front of whatever the caller enqueues next on `stream`).
 *   3 (default; env NEOSR_AMD_BLOCK_STREAMS=3): a HAB's CAB branch — a chain of small launches that meets the attention
 *     branch only at the sum in front of norm2 (forward) / at norm1's input gradient (backward) — runs beside the attention
 *     branch: hat_l (B = 4) 95.2 -> 89.1 ms per step.  Blocks without a CAB, and calls on a stream under hipGraph capture,
 *     stay on the caller's stream.
 *   2: the block's weight gradients (4 split-K GEMMs, the 2 CAB convolutions' gradients) beside the data-gradient chain
 *     — measured neutral on swinir_medium and 4 % slower on hat_l (14 event calls per block).
 *   1: everything on the caller's stream.
 * A scheduling choice: bit-identical results.  Returns the previous setting. */
int neosr_set_tblock_streams(int n);
/* Forks onto the side stream since the library was loaded (diagnostics / tests: was the side-stream path taken?). */
int64_t neosr_tblock_side_forks(void);
/* The TAIL of neosr_tblock_backward — the grouped weight-gradient GEMMs and the batched column sums that finish the block's
 * parameter gradients; nothing in the backward pass reads them — runs on a second library stream and the call returns with
 * it in flight (1 = default, env NEOSR_AMD_BLOCK_TAIL; only with neosr_set_tblock_streams(3), never under hipGraph capture).
 * Contract for the caller (neosr_amd/hip/transformer.py implements it; reference call sites: autograd of
 * neosr/archs/swinir_arch.py:231-392, hat_arch.py:218-515):
 *   - `workspace`, `save` and `dout` of a call must stay allocated and unwritten until its tail has finished:
 *     neosr_tblock_tail_done(n) (n = neosr_tblock_tails() before the call) is a host-side event query, 1 = finished;
 *     individually answerable for the newest seven tails only — keep at most six calls' buffers, join beyond that — or
 *     until neosr_tblock_tail_join has been enqueued on the stream that will reuse the memory;
 *   - the parameter gradients are complete behind neosr_tblock_tail_join(stream): `stream` waits for every tail issued so
 *     far; returns how many were outstanding (0: nothing to wait for, -1: error).
 * neosr_tblock_tails(): tails issued so far (callers compare it around a call to learn whether one was issued). */
int neosr_set_tblock_tail(int on);
int64_t neosr_tblock_tail_join(void* stream);
int64_t neosr_tblock_tails(void);
int neosr_tblock_tail_done(int64_t index);

#ifdef __cplusplus
}
#endif
#endif /* NEOSR_AMD_H */
