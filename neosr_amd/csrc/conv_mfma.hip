// conv_mfma.hip — 3x3/s1/p1 convolution, forward and backward-data, for gfx950 (MI355X).
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate):
//   M = 32 output pixels of one image row, N = 32 output channels (x2 per workgroup),
//   K = (chunk of 16 reduction channels) x 9 taps.
// A 256-thread workgroup owns a 4x32-pixel output tile (one row per wavefront).  Per chunk the
// 6x34 input halo tile and the weight slab are staged through LDS; the global loads of chunk c+1
// are issued before chunk c goes to the matrix pipe and stay in flight in registers (they are
// only consumed by the LDS store after the compute), so HBM/L2 latency hides behind 72..144 MFMAs.
// LDS layouts are padded to odd strides so that every MFMA operand read is conflict-free:
//   input  [pixel][16+1]         A[i=pixel][k]  lane l reads pixel l&31, channel 2*ks + (l>>5)
//   weight fwd   [n][16*9+1]     B[k][j=n]      straight copy of the canonical (N,K,3,3) rows
//   weight dgrad [k][64*9+1]     B[k][j=n]      transposed use of the same canonical rows, tap 8-t
// Epilogue fuses bias, LeakyReLU/PReLU/ReLU, two scaled residuals and optional accumulate.
// Loader fuses: nearest-x2 upsample (gather), activation-derivative mask, PReLU-on-load.
//
// Reference call sites replaced: neosr/archs/esrgan_arch.py:109-116,137-142,196-214;
// neosr/archs/compact_arch.py:76-79 (see include/neosr_amd.h).
#include <cstring>
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "conv_pack.h"
#include "../../include/neosr_amd.h"

#ifndef NEOSR_INTERLEAVE
#define NEOSR_INTERLEAVE 1  // spread the next chunk's global loads over the taps of the current one
#endif
#ifndef NEOSR_CONV_WPS
#define NEOSR_CONV_WPS 2  // waves per SIMD the register allocator must leave room for
#endif

namespace {

constexpr int TH = 4;               // output rows per workgroup (one per wave)
constexpr int TW = 32;              // output cols per workgroup (one MFMA M-tile)
constexpr int CK = 16;              // reduction channels per chunk
constexpr int HALO_W = TW + 2;      // 34
constexpr int HALO_H = TH + 2;      // 6
constexpr int IN_PIX = HALO_H * HALO_W;   // 204
constexpr int INS = CK + 1;         // LDS pixel stride (odd -> conflict-free A reads)
constexpr int NT = 2;               // 32-wide N tiles per workgroup
constexpr int NB = NT * 32;         // 64 output channels per workgroup
constexpr int WROW_F = CK * 9 + 1;  // fwd   weight LDS row stride 145
constexpr int WROW_D = NB * 9 + 1;  // dgrad weight LDS row stride 577
constexpr int IN_LDS = IN_PIX * INS;                                            // 3468 floats
constexpr int W_LDS = (NB * WROW_F > CK * WROW_D) ? NB * WROW_F : CK * WROW_D;  // 9280 floats
constexpr int IN_F4 = (IN_PIX * 4 + 255) / 256;                                 // 4 float4 / thread
constexpr int W_F4 = (NB * CK * 9) / (256 * 4);                                 // 9 float4 / thread

struct ConvArgs {
  neosr_conv_desc d;
  int tiles_x, tiles_y;
  int scalar_in;                 // thin-K kernel: the input's channel stride / base is not 16-byte friendly
  unsigned long long* timeline;  // debug only (NEOSR_TIMELINE builds)
};

#ifdef NEOSR_TIMELINE
#define TL_MARK(slot)                                                         \
  do {                                                                        \
    if (args.timeline && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) \
      args.timeline[(threadIdx.x >> 6) * 64 + (slot)] = clock64();             \
  } while (0)
#else
#define TL_MARK(slot) do {} while (0)
#endif

// 4x4 / stride-2 kernels run as a 3x3 over the space-to-depth tensor (neosr_conv_desc.s2d_c): a channel
// of sub-pixel (dy, dx) only meets block taps by in {1, dy ? 0 : 2}, bx in {1, dx ? 0 : 2}.  Returns the 9-bit
// mask of live taps in LOOP order (backward-data walks the taps flipped).
__device__ __forceinline__ int s2d_tap_mask(int sub, bool dgrad) {
  const int dy = (sub >> 1) & 1, dx = sub & 1;
  int r1 = dy ? 0 : 2, c1 = dx ? 0 : 2;
  if (dgrad) { r1 = 2 - r1; c1 = 2 - c1; }
  return (1 << 4) | (1 << (3 + c1)) | (1 << (r1 * 3 + 1)) | (1 << (r1 * 3 + c1));
}

// MASKED = false keeps the tap loop free of branches (the compiler software-pipelines the LDS reads across
// taps); the s2d_c launches pay a wave-uniform branch per tap instead.
template <bool DGRAD, int NTV, bool MASKED, class Hook>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ lin,
                                              const float* __restrict__ lw, int wave, int l31,
                                              int lh, f32x16 (&acc)[NT], Hook&& hook, int tapmask) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    // per-tap hook: issues a slice of the next chunk's global loads so that the 13 loads of a
    // chunk are spread over the 72..144 MFMAs instead of hitting the memory pipe as one burst
    hook(tap);
#if NEOSR_INTERLEAVE
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (MASKED && !((tapmask >> tap) & 1)) continue;  // wave-uniform
    const int ty = tap / 3, tx = tap % 3;
    const float* ap = lin + ((wave + ty) * HALO_W + l31 + tx) * INS + lh;
#pragma unroll
    for (int ks = 0; ks < CK / 2; ++ks) {
      const float a = ap[ks * 2];
#pragma unroll
      for (int nt = 0; nt < NTV; ++nt) {
        float b;
        if (DGRAD)
          b = lw[(ks * 2 + lh) * WROW_D + (nt * 32 + l31) * 9 + (8 - tap)];
        else
          b = lw[(nt * 32 + l31) * WROW_F + (ks * 2 + lh) * 9 + tap];
        // D = W * X^T: rows i = output channel, cols j = pixel.  Each lane then holds, for ITS
        // pixel (lane&31), four runs of 4 consecutive channels -> dwordx4 epilogue stores
        // (dword-per-lane stores are issue-bound at ~7 B/clk/CU on gfx950).
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[nt], 0, 0, 0);
      }
    }
  }
}

// Out-of-range lanes are redirected on the ADDRESS side (to a zero page for loads, to a per-lane
// trash slot for stores) so that no VALU ever touches a loaded value before the LDS store and the
// epilogue is straight-line code: a select on the DATA side makes hipcc wait for the load right
// where it was issued, which serialises the prefetch (measured: 1.6-5.6k cycles per chunk).
__device__ __attribute__((aligned(256))) float g_zero_page[64];
__device__ __attribute__((aligned(256))) float g_trash[1024];  // 16 B per thread of a workgroup

// generic guarded 4-channel load (any alignment, ragged channel count)
__device__ __forceinline__ float4 ld4_generic(const float* p, int c, int C, float fill) {
  float4 v = make_float4(fill, fill, fill, fill);
  if (c < C) v.x = p[0];
  if (c + 1 < C) v.y = p[1];
  if (c + 2 < C) v.z = p[2];
  if (c + 3 < C) v.w = p[3];
  return v;
}

// FAST-path epilogue of one 32-channel tile, in two halves so that a kernel can issue the loads
// (bias, slopes, residuals, accumulate-in, derivative mask) ahead of its last chunk of MFMAs.
// D layout (D = W * X^T): lane holds pixel lane&31 and channels nbase + 8g + 4*(lane>>5) + {0..3} in
// acc[4g..4g+3].  Straight-line code: invalid lanes are redirected on the address side.
struct EpiRegs {
  float4 bias[4], sl[4], a0[4], a1[4], a2[4], mk[4];
};

__device__ __forceinline__ void epi_load(const neosr_conv_desc& d, int nbase, int64_t pix, bool pix_ok,
                                         int lh, float s_uni, bool extra, EpiRegs& R) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int chq = nbase + 8 * g + 4 * lh;
    const bool ok = pix_ok && chq < d.N;
    const int cs0 = chq < d.N ? chq : 0;
    R.bias[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    R.sl[g] = make_float4(s_uni, s_uni, s_uni, s_uni);
    R.a0[g] = R.a1[g] = R.a2[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    R.mk[g] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (d.bias) R.bias[g] = *reinterpret_cast<const float4*>(d.bias + cs0);
    if (d.act == ACT_PRELU) R.sl[g] = *reinterpret_cast<const float4*>(d.prelu + cs0);
    if (extra) {
      R.a1[g] = *reinterpret_cast<const float4*>(
          (ok && d.res1 && chq < d.res1_nch) ? d.res1 + pix * d.res1_cs + chq : g_zero_page);
      R.a2[g] = *reinterpret_cast<const float4*>(
          (ok && d.res2 && chq < d.res2_nch) ? d.res2 + pix * d.res2_cs + chq : g_zero_page);
      R.a0[g] = *reinterpret_cast<const float4*>(
          (ok && d.accumulate) ? d.out + pix * d.out_cs + chq : g_zero_page);
    }
    if (d.out_mask)  // invalid lanes read zeros -> scaled garbage goes to the trash slot
      R.mk[g] = *reinterpret_cast<const float4*>(ok ? d.out_mask + pix * d.out_mask_cs + chq : g_zero_page);
  }
}

__device__ __forceinline__ void epi_store(const neosr_conv_desc& d, const f32x16& acc, int nbase,
                                          int64_t pix, bool pix_ok, int lh, int tid, const EpiRegs& R) {
  float4 o[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float bb[4] = {R.bias[g].x, R.bias[g].y, R.bias[g].z, R.bias[g].w};
    const float ss[4] = {R.sl[g].x, R.sl[g].y, R.sl[g].z, R.sl[g].w};
    const float r1[4] = {R.a1[g].x, R.a1[g].y, R.a1[g].z, R.a1[g].w};
    const float r2[4] = {R.a2[g].x, R.a2[g].y, R.a2[g].z, R.a2[g].w};
    const float r0[4] = {R.a0[g].x, R.a0[g].y, R.a0[g].z, R.a0[g].w};
    const float mm[4] = {R.mk[g].x, R.mk[g].y, R.mk[g].z, R.mk[g].w};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[4 * g + e] + bb[e];
      t = t > 0.f ? t : t * ss[e];
      t = t * d.alpha + r1[e];
      t = t * d.alpha2 + r2[e];
      t += r0[e];
      v[e] = mm[e] > 0.f ? t : t * d.out_mask_slope;
    }
    o[g] = make_float4(v[0], v[1], v[2], v[3]);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int chq = nbase + 8 * g + 4 * lh;
    const bool ok = pix_ok && chq < d.N;
    *reinterpret_cast<float4*>(ok ? d.out + pix * d.out_cs + chq : g_trash + tid * 4) = o[g];
  }
}

// staging loads are issued early in the chunk so they have >= 3 taps of MFMAs to land:
// input quads i (0..3) in pieces 0,0,1,1; weight quads rr (0..8) in pieces 1,2,2,3,3,4,4,5,5
__device__ __forceinline__ constexpr int w_piece(int rr) { return 1 + ((rr + 1) >> 1); }

// FAST path preconditions (checked on the host): in/mask/w (and the per-channel slope vector of a
// PReLU-on-load or PReLU-derivative mask, at most one of the two) 16-byte aligned, in_cs, mask_cs, K,
// w_cin, N multiples of 4.
template <bool DGRAD, bool MASK, bool GENERIC, bool S2D = false>
__global__ __launch_bounds__(256, GENERIC ? 2 : NEOSR_CONV_WPS) void conv3x3_mfma_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ float lds[IN_LDS + W_LDS];
  float* lin = lds;
  float* lw = lds + IN_LDS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  TL_MARK(0);

  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * NB;
  const int nvalid = min(NB, d.N - n0);
  const int ntv = (nvalid + 31) >> 5;

  const int H = d.H, W = d.W, K = d.K;
  const int Hin = d.ups ? (H >> 1) : H, Win = d.ups ? (W >> 1) : W;
  const float* __restrict__ inb = d.in + (int64_t)b * Hin * Win * d.in_cs;
  const bool has_mask = GENERIC ? (d.in_mask != nullptr) : MASK;
  const float* __restrict__ maskb =
      has_mask ? d.in_mask + (int64_t)b * Hin * Win * d.mask_cs : nullptr;

  // per-thread input staging slots (chunk-invariant): pixel offset or -1, and channel quad
  const int q4 = (tid & 3) << 2;
  int in_off[IN_F4], mk_off[IN_F4];
#pragma unroll
  for (int i = 0; i < IN_F4; ++i) {
    const int pix = (tid >> 2) + i * 64;
    in_off[i] = -1;
    mk_off[i] = 0;
    if (pix < IN_PIX) {
      const int py = pix / HALO_W, px = pix - py * HALO_W;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int sy = d.ups ? (gy >> 1) : gy, sx = d.ups ? (gx >> 1) : gx;
        in_off[i] = (sy * Win + sx) * d.in_cs + q4;
        mk_off[i] = (sy * Win + sx) * d.mask_cs + q4;
      }
    }
  }

  // weight staging: thread -> (row, quad) of the slab; 9 float4 per thread
  //   fwd  : rows = 64 output channels n, 36 quads per row (16 k x 9 taps contiguous in memory)
  //   dgrad: rows = 16 reduction channels k, 144 quads per row (64 n x 9 taps contiguous)
  const int w_row = DGRAD ? (tid >> 4) : (tid >> 2);
  const int w_q = DGRAD ? (tid & 15) : (tid & 3);
  constexpr int W_QSTEP = DGRAD ? 16 : 4;

  float4 rin[IN_F4];
  float4 rmk[(MASK || GENERIC) ? IN_F4 : 1];
  float4 rw[W_F4];
  float4 rsl = make_float4(1.f, 1.f, 1.f, 1.f);  // FAST: per-channel slopes of this thread's quad (PReLU on
                                                 // load / PReLU derivative mask), fetched with the chunk

  // piece p (0..8) of the staging loads of the chunk starting at channel c0; p < 0 = all pieces
  auto gload = [&](int c0, int piece) {
    const int c = c0 + q4;
    if (!GENERIC && piece <= 0) {
      const float* sp = d.in_prelu ? d.in_prelu : d.mask_slopes;  // wave-uniform
      if (sp) rsl = *reinterpret_cast<const float4*>(c < K ? sp + c : g_zero_page);
    }
#pragma unroll
    for (int i = 0; i < IN_F4; ++i) {
      if (piece >= 0 && piece != (i >> 1)) continue;
      const bool ok = in_off[i] >= 0 && c < K;
      if (!GENERIC) {
        rin[i] = *reinterpret_cast<const float4*>(ok ? inb + in_off[i] + c0 : g_zero_page);
        if (MASK) rmk[i] = *reinterpret_cast<const float4*>(ok ? maskb + mk_off[i] + c0 : g_zero_page);
      } else {
        rin[i] = ok ? ld4_generic(inb + in_off[i] + c0, c, K, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_mask)
          rmk[i] = ok ? ld4_generic(maskb + mk_off[i] + c0, c, K, 1.f) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
    }
    const int ckv = min(CK, K - c0);
    if (!DGRAD) {
      const int n = w_row;  // < 64
      const float* wp = d.w + ((int64_t)(n0 + n) * d.w_cin + c0) * 9;
      const int rlim = ckv * 9;  // valid floats in this row
#pragma unroll
      for (int rr = 0; rr < W_F4; ++rr) {
        if (piece >= 0 && piece != w_piece(rr)) continue;
        const int r = (w_q + W_QSTEP * rr) * 4;
        const bool ok = n < nvalid && r < rlim;
        if (!GENERIC)
          rw[rr] = *reinterpret_cast<const float4*>(ok ? wp + r : g_zero_page);
        else
          rw[rr] = ok ? ld4_generic(wp + r, r, rlim, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int k = w_row;  // < 16
      const float* wp = d.w + ((int64_t)(c0 + k) * d.w_cin + n0) * 9;
      const int rlim = nvalid * 9;
#pragma unroll
      for (int rr = 0; rr < W_F4; ++rr) {
        if (piece >= 0 && piece != w_piece(rr)) continue;
        const int r = (w_q + W_QSTEP * rr) * 4;
        const bool ok = k < ckv && r < rlim;
        if (!GENERIC)
          rw[rr] = *reinterpret_cast<const float4*>(ok ? wp + r : g_zero_page);
        else
          rw[rr] = ok ? ld4_generic(wp + r, r, rlim, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  auto sstore = [&](int c0) {
#pragma unroll
    for (int i = 0; i < IN_F4; ++i) {
      const int pix = (tid >> 2) + i * 64;
      if (pix < IN_PIX) {
        float4 v = rin[i];
        if (!GENERIC && d.in_prelu) {
          v.x = v.x > 0.f ? v.x : v.x * rsl.x;
          v.y = v.y > 0.f ? v.y : v.y * rsl.y;
          v.z = v.z > 0.f ? v.z : v.z * rsl.z;
          v.w = v.w > 0.f ? v.w : v.w * rsl.w;
        }
        if (GENERIC && d.in_prelu) {
          const int c = c0 + q4;
          const float4 s = ld4_generic(d.in_prelu + c, c, K, 1.f);
          v.x = v.x > 0.f ? v.x : v.x * s.x;
          v.y = v.y > 0.f ? v.y : v.y * s.y;
          v.z = v.z > 0.f ? v.z : v.z * s.z;
          v.w = v.w > 0.f ? v.w : v.w * s.w;
        }
        if (has_mask) {
          const float4 m = rmk[i];
          float4 s = make_float4(d.mask_slope, d.mask_slope, d.mask_slope, d.mask_slope);
          if (!GENERIC && d.mask_slopes) s = rsl;
          if (GENERIC && d.mask_slopes) {
            const int c = c0 + q4;
            s = ld4_generic(d.mask_slopes + c, c, K, 1.f);
          }
          v.x = m.x > 0.f ? v.x : v.x * s.x;
          v.y = m.y > 0.f ? v.y : v.y * s.y;
          v.z = m.z > 0.f ? v.z : v.z * s.z;
          v.w = m.w > 0.f ? v.w : v.w * s.w;
        }
        float* p = lin + pix * INS + q4;
        p[0] = v.x;
        p[1] = v.y;
        p[2] = v.z;
        p[3] = v.w;
      }
    }
    float* lp = lw + w_row * (DGRAD ? WROW_D : WROW_F) + w_q * 4;
#pragma unroll
    for (int rr = 0; rr < W_F4; ++rr) {
      float* p = lp + rr * W_QSTEP * 4;
      p[0] = rw[rr].x;
      p[1] = rw[rr].y;
      p[2] = rw[rr].z;
      p[3] = rw[rr].w;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const int nchunks = (K + CK - 1) / CK;
  gload(0, -1);
  TL_MARK(1);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    TL_MARK(2 + c * 4);
    sstore(c * CK);
    __syncthreads();
    TL_MARK(3 + c * 4);
    const int cn = (c + 1) * CK;
#define CHUNK(NTV_) compute_chunk<DGRAD, NTV_, S2D>(lin, lw, wave, l31, lh, acc, hook, tapmask)
    int tapmask = 0x1ff;
    if (S2D) {  // FWD: this chunk's channels, DGRAD: this workgroup's output channels, in one sub-pixel
      if (!DGRAD && d.s2d_c % CK == 0) tapmask = s2d_tap_mask((c * CK) / d.s2d_c, false);
      if (DGRAD && d.s2d_c % NB == 0) tapmask = s2d_tap_mask(n0 / d.s2d_c, true);
    }
#if NEOSR_INTERLEAVE
    TL_MARK(4 + c * 4);
    if (c + 1 < nchunks) {
      auto hook = [&](int tap) { gload(cn, tap); };
      if (ntv == 2) CHUNK(2);
      else CHUNK(1);
    } else {
      auto hook = [](int) {};
      if (ntv == 2) CHUNK(2);
      else CHUNK(1);
    }
#else
    if (c + 1 < nchunks) gload(cn, -1);
    TL_MARK(4 + c * 4);
    auto hook = [](int) {};
    if (ntv == 2) CHUNK(2);
    else CHUNK(1);
#endif
    TL_MARK(5 + c * 4);
  }

  // D layout (32x32, D = W * X^T): col j = lane&31 -> pixel x0+j of this wave's row,
  // row i = (r&3) + 8*(r>>2) + 4*(lane>>5) -> channel: register quad g = r>>2 holds channels
  // 8g + 4*(lane>>5) + {0,1,2,3}.  Straight-line code: invalid lanes are redirected on the
  // address side (zero page / trash); FAST path moves 16 bytes per lane per instruction.
  TL_MARK(62);
  const int y = y0 + wave, x = x0 + l31;
  const bool pix_ok = y < H && x < W;
  const int64_t pix = pix_ok ? ((int64_t)b * H + y) * W + x : 0;
  // every supported activation is  v > 0 ? v : v * s
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const bool extra = d.res1 || d.res2 || d.accumulate;  // wave-uniform
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt >= ntv) break;
    if (!GENERIC) {
      EpiRegs R;
      epi_load(d, n0 + nt * 32, pix, pix_ok, lh, s_uni, extra, R);
      epi_store(d, acc[nt], n0 + nt * 32, pix, pix_ok, lh, tid, R);
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int chq = n0 + nt * 32 + 8 * g + 4 * lh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = chq + e;
          const bool ok = pix_ok && ch < d.N;
          const int cs0 = ch < d.N ? ch : 0;
          const float bias = d.bias ? d.bias[cs0] : 0.f;
          const float sl = d.act == ACT_PRELU ? d.prelu[cs0] : s_uni;
          const float a1 = *((ok && d.res1 && ch < d.res1_nch) ? d.res1 + pix * d.res1_cs + ch : g_zero_page);
          const float a2 = *((ok && d.res2 && ch < d.res2_nch) ? d.res2 + pix * d.res2_cs + ch : g_zero_page);
          float* const op = d.out + pix * d.out_cs + ch;
          const float a0 = *((ok && d.accumulate) ? op : g_zero_page);
          float t = acc[nt][4 * g + e] + bias;
          t = t > 0.f ? t : t * sl;
          t = t * d.alpha + a1;
          t = t * d.alpha2 + a2;
          t += a0;
          if (d.out_mask) {
            const float m = *(ok ? d.out_mask + pix * d.out_mask_cs + ch : g_zero_page);
            t = m > 0.f ? t : t * d.out_mask_slope;
          }
          *(ok ? op : g_trash + tid * 4 + e) = t;
        }
      }
    }
  }
  TL_MARK(63);
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant (needs d.w_pack).  One workgroup = 4 rows x 32 pixels x 32 output channels;
// chunks of 16 reduction channels; two LDS buffers of 31 KB (input halo 13 KB + weight slab 18 KB) so
// two workgroups share a CU.  Per chunk a wave issues 7-8 global_load_lds_dwordx4 (no staging VGPRs,
// no ds_write pass) and the workgroup meets at ONE barrier.
//   input image : granule (16 B) index = p*4 + (kq ^ ((p >> 2) & 3)), p = halo pixel (6 x 34), kq =
//                 channel quad; the permutation is applied on the (per-lane) global address, so 4
//                 lanes still fetch one pixel's 64 contiguous bytes, and the 16 lanes of a
//                 ds_read_b128 phase (consecutive pixels, same kq) land in 16 different bank groups
//   weight image: [tap][kq][n 32] granules — already the order of neosr_conv3x3_pack_weights()
// Fragments: lanes lh = 0 read quad 2s, lanes lh = 1 quad 2s+1; MFMA e of step s multiplies channel
// 8s + 4lh + e on both operands.
constexpr int GL_IN_GRAN = 13 * 64;                    // 816 used
constexpr int GL_W_GRAN = 9 * 4 * 32;                  // 1152 = 18 wave loads
constexpr int GL_BUF = (GL_IN_GRAN + GL_W_GRAN) * 4;   // floats per buffer (31 744 B)

__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <bool S2D>
__global__ __launch_bounds__(256, 2) void conv3x3_glds_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ __attribute__((aligned(1024))) float lds[2 * GL_BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  TL_MARK(0);

  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * 32;
  const int H = d.H, W = d.W, K = d.K;
  const int Hin = d.ups ? (H >> 1) : H, Win = d.ups ? (W >> 1) : W;
  const float* __restrict__ inb = d.in + (int64_t)b * Hin * Win * d.in_cs;
  const int nchunks = (K + CK - 1) / CK;
  const float* __restrict__ wp = d.w_pack + (int64_t)blockIdx.y * nchunks * (GL_W_GRAN * 4) + lane * 4;

  // input granule of this thread in wave-load i: g = i*256 + tid -> pixel i*64 + tid/4, slot tid&3;
  // the slot holds channel quad (tid & 3) ^ ((p >> 2) & 3), and (p >> 2) & 3 == (tid >> 4) & 3 for all i
  const int q4 = ((tid & 3) ^ ((tid >> 4) & 3)) << 2;
  int in_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pix = (tid >> 2) + i * 64;
    in_off[i] = -1;
    if (pix < IN_PIX) {
      const int py = pix / HALO_W, px = pix - py * HALO_W;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int sy = d.ups ? (gy >> 1) : gy, sx = d.ups ? (gx >> 1) : gx;
        in_off[i] = (sy * Win + sx) * d.in_cs + q4;
      }
    }
  }

  auto issue = [&](int c, int buf) {
    float* ibuf = lds + buf * GL_BUF;
    float* wbuf = ibuf + GL_IN_GRAN * 4;
    const int c0 = c * CK;
    const bool kq_ok = c0 + q4 < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 3 && wave != 0) break;
      const float* src = (in_off[i] >= 0 && kq_ok) ? inb + in_off[i] + c0 : g_zero_page;
      glds16(src, ibuf + (i * 4 + wave) * 256);
    }
    const float* ws = wp + (int64_t)c * (GL_W_GRAN * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = (3 - wave) + 4 * i;  // 18 slab loads dealt so that every wave issues 7-8 in total
      if (j < 18) glds16(ws + j * 256, wbuf + j * 256);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const bool dg = d.mode == NEOSR_CONV_DGRAD;
  const int s2d_dgrad_mask = (d.s2d_c > 0 && dg && d.s2d_c % 32 == 0) ? s2d_tap_mask(n0 / d.s2d_c, true) : 0x1ff;
  // S2D = false keeps the tap loop branch-free (the compiler software-pipelines the LDS reads across
  // taps); s2d_c launches pay a wave-uniform branch per tap
  auto compute = [&](int buf, int tapmask) {
    const float* ibuf = lds + buf * GL_BUF;
    const float* wbuf = ibuf + GL_IN_GRAN * 4 + (lh * 32 + l31) * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (S2D && !((tapmask >> tap) & 1)) continue;  // wave-uniform
      const int p = (wave + tap / 3) * HALO_W + l31 + tap % 3;
      const int sw = (p >> 2) & 3;
      const float* ap = ibuf + p * 16;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + ((lh ^ sw) << 2));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + (((2 + lh) ^ sw) << 2));
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(wbuf + (tap * 4) * 128);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(wbuf + (tap * 4 + 2) * 128);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[e], a0[e], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[e], a1[e], acc, 0, 0, 0);
    }
  };

  auto chunk_mask = [&](int c) {
    if (d.s2d_c > 0 && !dg && d.s2d_c % CK == 0) return s2d_tap_mask((c * CK) / d.s2d_c, false);
    return s2d_dgrad_mask;
  };
  issue(0, 0);
  const int y = y0 + wave, x = x0 + l31;
  const bool pix_ok = y < H && x < W;
  const int64_t pix = pix_ok ? ((int64_t)b * H + y) * W + x : 0;
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const bool extra = d.res1 || d.res2 || d.accumulate;
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  TL_MARK(1);
  for (int c = 0; c + 1 < nchunks; ++c) {
    issue(c + 1, (c + 1) & 1);
    TL_MARK(2 + c * 4);
    compute(c & 1, chunk_mask(c));
    TL_MARK(3 + c * 4);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // chunk c+1 has landed ...
    __syncthreads();                      // ... for every wave, and buffer c&1 is free again
    TL_MARK(4 + c * 4);
  }
  EpiRegs R;
  epi_load(d, n0, pix, pix_ok, lh, s_uni, extra, R);  // in flight under the last 72 MFMAs
  compute((nchunks - 1) & 1, chunk_mask(nchunks - 1));
  TL_MARK(62);
  epi_store(d, acc, n0, pix, pix_ok, lh, tid, R);
  TL_MARK(63);
}

// ---------------------------------------------------------------------------------------------
// Thin convolutions: the first / last layers of every network here have 3 (or 1) channels on one side
// (esrgan / swinir / hat conv_first and conv_last, U-Net conv0 / conv9, VGG conv1_1).  On the 32-wide
// tiles above they waste 5-10x of the matrix pipe; these two kernels make them memory-bound instead.
//
// (1) K <= 4 reduction channels (3 -> 64 forward, 64 -> 3 backward-data): the reduction index is packed
//     as k' = tap*4 + ch (36 values, 18 MFMA steps of 32x32x2) instead of 9 taps x a 16-channel chunk.
//     One pass: stage the 6x34x4 halo and the 64 x 36 weight slab, 18 (x2) MFMAs per wave, epilogue.
constexpr int TK_INS = 5;               // LDS pixel stride (odd -> conflict-free b32 reads)
constexpr int TK_WROW = 37;             // weight row stride (36 + 1)

__global__ __launch_bounds__(256, 3) void conv3x3_thin_k_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ float lin[IN_PIX * TK_INS];
  __shared__ float lw[NB * TK_WROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * NB;
  const int nvalid = min(NB, d.N - n0);
  const int ntv = (nvalid + 31) >> 5;
  const int H = d.H, W = d.W, K = d.K;
  const bool dgrad = d.mode == NEOSR_CONV_DGRAD;

  if (tid < IN_PIX) {
    const int py = tid / HALO_W, px = tid - py * HALO_W;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* src = ok ? d.in + (((int64_t)b * H + gy) * W + gx) * d.in_cs : g_zero_page;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (args.scalar_in) {  // channel stride not a multiple of 4 / unaligned base: K scalar loads
      v.x = src[0];
      if (K > 1) v.y = src[1];
      if (K > 2) v.z = src[2];
      if (K > 3) v.w = src[3];
    } else {
      v = *reinterpret_cast<const float4*>(src);
    }
    float* q = lin + tid * TK_INS;  // channels >= K of the quad are padding of the buffer: never used
    q[0] = v.x;
    q[1] = K > 1 ? v.y : 0.f;
    q[2] = K > 2 ? v.z : 0.f;
    q[3] = K > 3 ? v.w : 0.f;
  }
  {  // 64 x 36 slab = 9 elements per thread, all loads in flight before the first LDS store
    float wv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int e = j * 256 + tid;
      const int n = e / 36, kp = e - n * 36, tap = kp >> 2, ch = kp & 3;
      const bool ok = n < nvalid && ch < K;
      const float* src = dgrad ? d.w + ((int64_t)ch * d.w_cin + n0 + n) * 9 + (8 - tap)
                               : d.w + ((int64_t)(n0 + n) * d.w_cin + ch) * 9 + tap;
      wv[j] = *(ok ? src : g_zero_page);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int e = j * 256 + tid;
      const int n = e / 36;
      lw[n * TK_WROW + (e - n * 36)] = wv[j];
    }
  }
  const int y = y0 + wave, x = x0 + l31;
  const bool pix_ok = y < H && x < W;
  const int64_t pix = pix_ok ? ((int64_t)b * H + y) * W + x : 0;
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const bool extra = d.res1 || d.res2 || d.accumulate;
  __syncthreads();

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int tap = s >> 1;                    // k' = 2s + lh -> tap = k' / 4, ch = k' % 4
    const int chb = (s & 1) * 2;               // ch = chb + lh
    const float a = lin[((wave + tap / 3) * HALO_W + l31 + tap % 3) * TK_INS + chb + lh];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt >= ntv) break;
      const float bv = lw[(nt * 32 + l31) * TK_WROW + 2 * s + lh];
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a, acc[nt], 0, 0, 0);
    }
  }
  // memory-bound kernel: registers are spent on resident waves (latency hiding), not on hoisted loads
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    if (nt < ntv) {
      EpiRegs R;
      epi_load(d, n0 + nt * 32, pix, pix_ok, lh, s_uni, extra, R);
      epi_store(d, acc[nt], n0 + nt * 32, pix, pix_ok, lh, tid, R);
    }
}

// (2) N <= 4 output channels (64 -> 3 forward, 3 <- 64 backward-data): v_mfma_f32_4x4x1_16b_f32, whose 16
//     independent 4x4 blocks are used as 4 output channels (rows) x 64 pixels (4 per block): lane = pixel,
//     the 4 accumulator registers = the 4 output channels, so the matrix pipe runs at 3/4 (N = 3) instead of
//     3/32 utilisation and every lane ends up holding exactly its own pixel.  Workgroup = 4 rows x 64 pixels,
//     32-channel chunks: input halo [6 x 66 px][32 + 4] in LDS (16-byte fragment reads, stride 36 floats:
//     conflict-free), weights [tap][k quad][n][4 k] read as 4-address broadcasts.
constexpr int TN_W = 64, TN_HW = TN_W + 2, TN_PIX = HALO_H * TN_HW;  // 396
constexpr int TN_CK = 32, TN_INS = TN_CK + 4;
constexpr int TN_F4 = (TN_PIX * (TN_CK / 4) + 255) / 256;            // 13 float4 per thread

template <bool MASK>
__global__ __launch_bounds__(256, 2) void conv3x3_thin_n_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ __attribute__((aligned(16))) float lin[TN_PIX * TN_INS];
  __shared__ __attribute__((aligned(16))) float lw[9 * (TN_CK / 4) * 4 * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TN_W, y0 = ty * TH;
  const int H = d.H, W = d.W, K = d.K, N = d.N;
  const bool dgrad = d.mode == NEOSR_CONV_DGRAD;
  const float* __restrict__ inb = d.in + (int64_t)b * H * W * d.in_cs;
  const float* __restrict__ mkb = MASK ? d.in_mask + (int64_t)b * H * W * d.mask_cs : nullptr;

  // staging slots: granule g = i*256 + tid -> pixel g / 8, channel quad g % 8
  int in_off[TN_F4], mk_off[TN_F4];
#pragma unroll
  for (int i = 0; i < TN_F4; ++i) {
    const int g = i * 256 + tid, p = g >> 3;
    in_off[i] = -1;
    mk_off[i] = 0;
    if (p < TN_PIX) {
      const int py = p / TN_HW, px = p - py * TN_HW;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        in_off[i] = (gy * W + gx) * d.in_cs + (g & 7) * 4;
        mk_off[i] = (gy * W + gx) * d.mask_cs + (g & 7) * 4;
      }
    }
  }

  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < K; c0 += TN_CK) {
    float4 rin[TN_F4], rmk[MASK ? TN_F4 : 1];
#pragma unroll
    for (int i = 0; i < TN_F4; ++i) {
      const bool ok = in_off[i] >= 0 && c0 + ((i * 256 + tid) & 7) * 4 < K;
      rin[i] = *reinterpret_cast<const float4*>(ok ? inb + in_off[i] + c0 : g_zero_page);
      if (MASK) rmk[i] = *reinterpret_cast<const float4*>(ok ? mkb + mk_off[i] + c0 : g_zero_page);
    }
    // weight slab of this chunk: lw[tap][kq][n][e] = W(n, c0 + 4 kq + e, tap)
    float wv[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = j * 256 + tid;
      wv[j] = 0.f;
      if (e < 9 * (TN_CK / 4) * 16) {
        const int ke = e & 3, n = (e >> 2) & 3, kq = (e >> 4) & 7, tap = e >> 7;
        const int k = c0 + kq * 4 + ke;
        if (n < N && k < K)
          wv[j] = dgrad ? d.w[((int64_t)k * d.w_cin + n) * 9 + (8 - tap)] : d.w[((int64_t)n * d.w_cin + k) * 9 + tap];
      }
    }
    __syncthreads();  // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < TN_F4; ++i) {
      const int g = i * 256 + tid, p = g >> 3;
      if (p < TN_PIX) {
        float4 v = rin[i];
        if (MASK) {
          const float4 m = rmk[i];
          v.x = m.x > 0.f ? v.x : v.x * d.mask_slope;
          v.y = m.y > 0.f ? v.y : v.y * d.mask_slope;
          v.z = m.z > 0.f ? v.z : v.z * d.mask_slope;
          v.w = m.w > 0.f ? v.w : v.w * d.mask_slope;
        }
        *reinterpret_cast<float4*>(lin + p * TN_INS + (g & 7) * 4) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = j * 256 + tid;
      if (e < 9 * (TN_CK / 4) * 16) lw[e] = wv[j];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* xp = lin + ((wave + tap / 3) * TN_HW + lane + tap % 3) * TN_INS;
      const float* wq = lw + (tap * (TN_CK / 4) * 4 + (lane & 3)) * 4;
#pragma unroll
      for (int kq = 0; kq < TN_CK / 4; ++kq) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + kq * 4);
        const f32x4 wf = *reinterpret_cast<const f32x4*>(wq + kq * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[e], xv[e], acc[e], 0, 0, 0);
      }
    }
  }
  const f32x4 r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  const int y = y0 + wave, x = x0 + lane;
  if (y < H && x < W) {
    float s_uni = 1.f;
    if (d.act == ACT_LRELU) s_uni = d.slope;
    else if (d.act == ACT_RELU) s_uni = 0.f;
    float* op = d.out + (((int64_t)b * H + y) * W + x) * d.out_cs;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (n < N) {
        float t = r[n] + (d.bias ? d.bias[n] : 0.f);
        op[n] = t > 0.f ? t : t * s_uni;
      }
    }
  }
}

// weight repack (see conv_pack.h): one thread per 16-byte granule of the destination image
__global__ __launch_bounds__(256) void conv_pack_kernel(const neosr_pack::Batch batch) {
  const neosr_pack::Image& im = batch.im[blockIdx.y];
  const int nch = (im.K + 15) >> 4, nblk = (im.N + 31) >> 5;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nblk * nch * GL_W_GRAN) return;
  const int n32 = g & 31, kq = (g >> 5) & 3;
  int rest = g >> 7;
  const int tap = rest % 9;
  rest /= 9;
  const int chunk = rest % nch, nb = rest / nch;
  const int n = nb * 32 + n32, k0 = chunk * 16 + kq * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < im.N && k0 < im.K) {
    for (int s = 0; s < im.nseg; ++s) {
      const neosr_pack::Seg& sg = im.seg[s];
      if (k0 < sg.k_lo || k0 >= sg.k_lo + sg.k_cnt) continue;
      const int kk = k0 - sg.k_lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kk + e >= sg.k_cnt) break;
        v[e] = im.mode == NEOSR_CONV_FWD
                   ? sg.w[((int64_t)(sg.n_lo + n) * sg.w_cin + kk + e) * 9 + tap]
                   : sg.w[((int64_t)(kk + e) * sg.w_cin + sg.n_lo + n) * 9 + (8 - tap)];
      }
    }
  }
  *reinterpret_cast<float4*>(im.dst + (int64_t)g * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

unsigned long long* g_timeline = nullptr;

}  // namespace

// debug hook (NEOSR_TIMELINE builds only record anything): device buffer of 4*64 uint64
extern "C" int neosr_debug_set_timeline(void* dev_buf) {
  g_timeline = (unsigned long long*)dev_buf;
  return 0;
}

extern "C" int neosr_conv3x3(const neosr_conv_desc* dp, void* stream) {
  const neosr_conv_desc& d = *dp;
  NEOSR_CHECK(d.in && (d.w || d.w_pack) && d.out, "conv3x3: null tensor");
  NEOSR_CHECK(d.B > 0 && d.H > 0 && d.W > 0 && d.K > 0 && d.N > 0, "conv3x3: bad geometry");
  NEOSR_CHECK(d.mode == NEOSR_CONV_FWD || d.mode == NEOSR_CONV_DGRAD, "conv3x3: bad mode");
  if (!d.w) {
    // packed image only: geometry was fixed when it was built
  } else if (d.mode == NEOSR_CONV_FWD)
    NEOSR_CHECK(d.K == d.w_cin && d.N <= d.w_cout, "conv3x3 fwd: K=%d N=%d vs w (%d,%d)", d.K, d.N,
                d.w_cout, d.w_cin);
  else
    NEOSR_CHECK(d.K == d.w_cout && d.N <= d.w_cin, "conv3x3 dgrad: K=%d N=%d vs w (%d,%d)", d.K,
                d.N, d.w_cout, d.w_cin);
  NEOSR_CHECK(!d.ups || ((d.H % 2 == 0) && (d.W % 2 == 0) && !d.in_mask),
              "conv3x3: ups needs even H,W and no mask");
  NEOSR_CHECK(d.act != NEOSR_ACT_PRELU || d.prelu, "conv3x3: PReLU needs slopes");
  ConvArgs a;
  a.d = d;
  a.tiles_x = ceil_div(d.W, TW);
  a.tiles_y = ceil_div(d.H, TH);
  a.scalar_in = 0;
  a.timeline = g_timeline;
  const bool al_in = (d.in_cs % 4 == 0) && ((uintptr_t)d.in % 16 == 0);
  const bool al_mk = !d.in_mask || ((d.mask_cs % 4 == 0) && ((uintptr_t)d.in_mask % 16 == 0));
  const bool al_w = ((uintptr_t)d.w % 16 == 0) && (d.w_cin % 4 == 0);
  auto al16 = [](const void* p, int cs) { return !p || (((uintptr_t)p % 16 == 0) && (cs % 4 == 0)); };
  const bool al_ep = al16(d.out, d.out_cs) && al16(d.res1, d.res1_cs) && al16(d.res2, d.res2_cs) &&
                     al16(d.bias, 0) && al16(d.prelu, 0) && (d.res1_nch % 4 == 0) &&
                     (d.res2_nch % 4 == 0) && al16(d.out_mask, d.out_mask_cs);
  // per-channel slopes ride along as one 16-byte load per chunk; both at once is not a fused case
  const bool fast = al_in && al_mk && al_w && al_ep && (d.K % 4 == 0) && (d.N % 4 == 0) &&
                    al16(d.in_prelu, 0) && al16(d.mask_slopes, 0) && !(d.in_prelu && d.mask_slopes);
  dim3 grid(a.tiles_x * a.tiles_y * d.B, ceil_div(d.N, NB));
  hipStream_t st = (hipStream_t)stream;
  // the direct-to-LDS kernel needs 16-byte granules everywhere; otherwise the staged kernel serves the
  // launch from the canonical weights (a gather-form launch has none and is refused)
  const bool use_pack = d.w_pack && al_in && al_ep && (d.K % 4 == 0) && (d.N % 4 == 0) && !d.in_mask &&
                        !d.in_prelu && ((uintptr_t)d.w_pack % 16 == 0);
  NEOSR_CHECK(use_pack || d.w, "conv3x3: w_pack launch needs 16-byte aligned tensors, K,N %% 4 == 0");
  if (use_pack) grid.y = ceil_div(d.N, 32);
  const bool prof = neosr_prof_on();
  if (prof) {
    const double px = (double)d.B * d.H * d.W;
    // algorithmic traffic: read |x| + |W|, write |y| (fp32)
    neosr_prof_begin(d.mode == NEOSR_CONV_FWD ? NEOSR_PROF_CONV_FWD : NEOSR_PROF_CONV_DGRAD, stream,
                     2.0 * px * d.K * d.N * 9.0,
                     4.0 * (px / (d.ups ? 4.0 : 1.0) * d.K + px * d.N + 9.0 * d.K * d.N));
  }
  // thin layers (see conv3x3_thin_*_kernel)
  const bool plain_in = d.w && al_in && !d.ups && !d.in_prelu && !d.mask_slopes;
  const bool thin_k = d.w && !d.ups && !d.in_prelu && !d.mask_slopes && d.K <= 4 && !d.in_mask && al_ep &&
                      (d.N % 4 == 0);
  a.scalar_in = thin_k && !al_in;
  const bool thin_n = plain_in && d.N <= 4 && d.K >= 8 && (d.K % 4 == 0) && al_mk && !d.res1 && !d.res2 &&
                      !d.accumulate && !d.out_mask && d.act != NEOSR_ACT_PRELU;
  if (use_pack) {
    if (d.s2d_c > 0) hipLaunchKernelGGL(conv3x3_glds_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3x3_glds_kernel<false>, grid, dim3(256), 0, st, a);
  } else if (thin_k) {
    hipLaunchKernelGGL(conv3x3_thin_k_kernel, grid, dim3(256), 0, st, a);
  } else if (thin_n) {
    a.tiles_x = ceil_div(d.W, TN_W);
    dim3 g2(a.tiles_x * a.tiles_y * d.B, 1);
    if (d.in_mask) hipLaunchKernelGGL(conv3x3_thin_n_kernel<true>, g2, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv3x3_thin_n_kernel<false>, g2, dim3(256), 0, st, a);
  } else if (d.mode == NEOSR_CONV_FWD) {
    if (!fast)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, false, true>), grid, dim3(256), 0, st, a);
    else if (d.in_mask)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, true, false>), grid, dim3(256), 0, st, a);
    else if (d.s2d_c > 0)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, false, false, true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, false, false>), grid, dim3(256), 0, st, a);
  } else {
    if (!fast)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, false, true>), grid, dim3(256), 0, st, a);
    else if (d.in_mask && d.s2d_c > 0)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, true, false, true>), grid, dim3(256), 0, st, a);
    else if (d.in_mask)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, true, false>), grid, dim3(256), 0, st, a);
    else if (d.s2d_c > 0)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, false, false, true>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, false, false>), grid, dim3(256), 0, st, a);
  }
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

int neosr_pack::launch(const Image* images, int n, void* stream) {
  NEOSR_CHECK(images && n > 0, "conv pack: bad arguments");
  for (int i0 = 0; i0 < n; i0 += BATCH) {
    const int cnt = n - i0 < BATCH ? n - i0 : BATCH;
    Batch bt;
    memset(&bt, 0, sizeof(bt));
    int64_t gran = 0;
    for (int i = 0; i < cnt; ++i) {
      bt.im[i] = images[i0 + i];
      const int64_t g = image_floats(bt.im[i].N, bt.im[i].K) / 4;
      gran = g > gran ? g : gran;
    }
    dim3 grid((unsigned)((gran + 255) / 256), cnt);
    hipLaunchKernelGGL(conv_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, bt);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t neosr_conv3x3_pack_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return -1;
  return neosr_pack::image_floats(N, K) * 4;
}

extern "C" int neosr_conv3x3_pack_weights(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode,
                                          float* dst, void* stream) {
  NEOSR_CHECK(w && dst && w_cout > 0 && w_cin > 0, "conv3x3_pack_weights: bad arguments");
  NEOSR_CHECK(mode == NEOSR_CONV_FWD || mode == NEOSR_CONV_DGRAD, "conv3x3_pack_weights: bad mode");
  NEOSR_CHECK((uintptr_t)dst % 16 == 0, "conv3x3_pack_weights: dst must be 16-byte aligned");
  neosr_pack::Image im;
  memset(&im, 0, sizeof(im));
  im.dst = dst;
  im.mode = mode;
  im.N = mode == NEOSR_CONV_FWD ? w_cout : w_cin;
  im.K = mode == NEOSR_CONV_FWD ? w_cin : w_cout;
  im.nseg = 1;
  im.seg[0].w = w;
  im.seg[0].w_cin = w_cin;
  im.seg[0].k_lo = 0;
  im.seg[0].k_cnt = im.K;
  im.seg[0].n_lo = 0;
  return neosr_pack::launch(&im, 1, stream);
}
