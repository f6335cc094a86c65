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
#include "common.h"
#include "prof.h"
#include "../../include/neosr_amd.h"

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

template <bool DGRAD, int NTV>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ lin,
                                              const float* __restrict__ lw, int wave, int l31,
                                              int lh, f32x16 (&acc)[NT]) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
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
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
      }
    }
  }
}

// Out-of-range lanes are redirected on the ADDRESS side (to a zero page for loads, to a per-lane
// trash slot for stores) so that no VALU ever touches a loaded value before the LDS store and the
// epilogue is straight-line code: a select on the DATA side makes hipcc wait for the load right
// where it was issued, which serialises the prefetch (measured: 1.6-5.6k cycles per chunk).
__device__ __attribute__((aligned(256))) float g_zero_page[64];
__device__ __attribute__((aligned(256))) float g_trash[256];

// generic guarded 4-channel load (any alignment, ragged channel count)
__device__ __forceinline__ float4 ld4_generic(const float* p, int c, int C, float fill) {
  float4 v = make_float4(fill, fill, fill, fill);
  if (c < C) v.x = p[0];
  if (c + 1 < C) v.y = p[1];
  if (c + 2 < C) v.z = p[2];
  if (c + 3 < C) v.w = p[3];
  return v;
}

// FAST path preconditions (checked on the host): in/mask/w 16-byte aligned, in_cs, mask_cs, K,
// w_cin, N multiples of 4, no PReLU-on-load, no per-channel mask slopes.
template <bool DGRAD, bool MASK, bool GENERIC>
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

  auto gload = [&](int c0) {
    const int c = c0 + q4;
#pragma unroll
    for (int i = 0; i < IN_F4; ++i) {
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
  gload(0);
  TL_MARK(1);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    TL_MARK(2 + c * 4);
    sstore(c * CK);
    __syncthreads();
    TL_MARK(3 + c * 4);
    if (c + 1 < nchunks) gload((c + 1) * CK);
    TL_MARK(4 + c * 4);
    if (ntv == 2)
      compute_chunk<DGRAD, 2>(lin, lw, wave, l31, lh, acc);
    else
      compute_chunk<DGRAD, 1>(lin, lw, wave, l31, lh, acc);
    TL_MARK(5 + c * 4);
  }

  // epilogue.  D layout (32x32): col j = lane&31 -> channel, row i = (r&3)+8*(r>>2)+4*(lane>>5) -> pixel
  // Straight-line code: invalid lanes are redirected on the address side (zero page / trash), all
  // residual loads of a tile are issued before any arithmetic, then 16 stores back to back.
  TL_MARK(62);
  const int y = y0 + wave;
  const bool row_ok = y < H;
  const int64_t rowpix = ((int64_t)b * H + (row_ok ? y : 0)) * W;
  const bool extra = d.res1 || d.res2 || d.accumulate;  // wave-uniform
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt >= ntv) break;
    const int ch = n0 + nt * 32 + l31;
    const bool ch_ok = ch < d.N;
    const int chs = ch_ok ? ch : 0;
    const float bias = d.bias ? d.bias[chs] : 0.f;
    // every supported activation is  v > 0 ? v : v * s
    float s = 1.f;
    if (d.act == ACT_LRELU) s = d.slope;
    else if (d.act == ACT_RELU) s = 0.f;
    else if (d.act == ACT_PRELU) s = d.prelu[chs];
    float* const obase = d.out + rowpix * d.out_cs + chs;
    float vals[16];
    if (extra) {
      const bool r1 = d.res1 && ch < d.res1_nch, r2 = d.res2 && ch < d.res2_nch;
      const float* const r1base = d.res1 ? d.res1 + rowpix * d.res1_cs + chs : g_zero_page;
      const float* const r2base = d.res2 ? d.res2 + rowpix * d.res2_cs + chs : g_zero_page;
      float a0[16], a1[16], a2[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool ok = row_ok && ch_ok && x < W;
        a1[r] = *((ok && r1) ? r1base + x * d.res1_cs : g_zero_page);
        a2[r] = *((ok && r2) ? r2base + x * d.res2_cs : g_zero_page);
        a0[r] = *((ok && d.accumulate) ? obase + x * d.out_cs : g_zero_page);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nt][r] + bias;
        v = v > 0.f ? v : v * s;
        v = v * d.alpha + a1[r];
        v = v * d.alpha2 + a2[r];
        vals[r] = v + a0[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[nt][r] + bias;
        v = v > 0.f ? v : v * s;
        v *= d.alpha;
        vals[r] = v * d.alpha2;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const bool ok = row_ok && ch_ok && x < W;
      float* p = ok ? obase + x * d.out_cs : g_trash + tid;
      *p = vals[r];
    }
  }
  TL_MARK(63);
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
  NEOSR_CHECK(d.in && d.w && d.out, "conv3x3: null tensor");
  NEOSR_CHECK(d.B > 0 && d.H > 0 && d.W > 0 && d.K > 0 && d.N > 0, "conv3x3: bad geometry");
  NEOSR_CHECK(d.mode == NEOSR_CONV_FWD || d.mode == NEOSR_CONV_DGRAD, "conv3x3: bad mode");
  if (d.mode == NEOSR_CONV_FWD)
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
  a.timeline = g_timeline;
  const bool al_in = (d.in_cs % 4 == 0) && ((uintptr_t)d.in % 16 == 0);
  const bool al_mk = !d.in_mask || ((d.mask_cs % 4 == 0) && ((uintptr_t)d.in_mask % 16 == 0));
  const bool al_w = ((uintptr_t)d.w % 16 == 0) && (d.w_cin % 4 == 0);
  const bool fast = al_in && al_mk && al_w && (d.K % 4 == 0) && (d.N % 4 == 0) && !d.in_prelu &&
                    !d.mask_slopes;
  dim3 grid(a.tiles_x * a.tiles_y * d.B, ceil_div(d.N, NB));
  hipStream_t st = (hipStream_t)stream;
  const bool prof = neosr_prof_on();
  if (prof) {
    const double px = (double)d.B * d.H * d.W;
    // algorithmic traffic: read |x| + |W|, write |y| (fp32)
    neosr_prof_begin(d.mode == NEOSR_CONV_FWD ? NEOSR_PROF_CONV_FWD : NEOSR_PROF_CONV_DGRAD, stream,
                     2.0 * px * d.K * d.N * 9.0,
                     4.0 * (px / (d.ups ? 4.0 : 1.0) * d.K + px * d.N + 9.0 * d.K * d.N));
  }
  if (d.mode == NEOSR_CONV_FWD) {
    if (!fast)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, false, true>), grid, dim3(256), 0, st, a);
    else if (d.in_mask)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, true, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv3x3_mfma_kernel<false, false, false>), grid, dim3(256), 0, st, a);
  } else {
    if (!fast)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, false, true>), grid, dim3(256), 0, st, a);
    else if (d.in_mask)
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, true, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv3x3_mfma_kernel<true, false, false>), grid, dim3(256), 0, st, a);
  }
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
