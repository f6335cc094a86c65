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
#include "conv_common.h"
#include "conv_wino4_chain.h"
#include "prof.h"
#include <stdlib.h>

#ifndef NEOSR_INTERLEAVE
#define NEOSR_INTERLEAVE 1  // spread the next chunk's global loads over the taps of the current one
#endif
#ifndef NEOSR_CONV_WPS
#define NEOSR_CONV_WPS 2  // waves per SIMD the register allocator must leave room for
#endif

using namespace neosr_conv;

namespace {

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

  int bid = xcd_tile(blockIdx.x, gridDim.x, args.xcd);
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


unsigned long long* g_timeline = nullptr;
int g_xcd = -1;  // -1: read NEOSR_AMD_XCD on first use (default on)

}  // namespace

bool neosr_conv::xcd_enabled() {
  if (g_xcd < 0) {
    const char* e = getenv("NEOSR_AMD_XCD");
    g_xcd = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_xcd != 0;
}

// XCD-aware workgroup order of the conv / weight-gradient kernels on (default) or off; returns the previous setting.
// Results do not depend on it (a permutation of independent workgroups; the split-K reduce order is fixed).
extern "C" int neosr_set_xcd_aware(int on) {
  const int prev = neosr_conv::xcd_enabled() ? 1 : 0;
  g_xcd = on ? 1 : 0;
  return prev;
}

unsigned long long* neosr_conv::debug_timeline() { return g_timeline; }

// debug hook (NEOSR_TIMELINE builds only record anything): device buffer of 4*64 uint64 (chain kernel: 12*128)
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
  a.tx_shift = a.ty_shift = -1;
  a.timeline = g_timeline;
  a.xcd = neosr_conv::xcd_enabled() ? 1 : 0;
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
  // Winograd F(2x2,3x3) form of the same launch (conv_wino.hip)
  const bool use_wino = use_pack && d.w_wino && ((uintptr_t)d.w_wino % 16 == 0) && d.s2d_c == 0 &&
                        d.act != NEOSR_ACT_PRELU && neosr_conv::wino_enabled();
  // ... or its F(4x4,3x3) form (conv_wino4.hip): same conditions (nearest-upsampled inputs are gathered by its DMA addresses too)
  // (32-bit byte offsets in its epilogue: tensors of 2 GB and more stay with F(2x2,3x3); small launches too, if they can)
  const int64_t w4_wgs = (int64_t)d.B * ceil_div(d.H, 16) * ceil_div(d.W, 16) * ceil_div(d.N, 32);
  auto small = [&](const void* p, int cs) { return !p || (int64_t)d.B * d.H * d.W * cs * 4 < (int64_t(1) << 31); };
  // (the F(4x4,3x3) kernel needs no direct image: a launch may come with w_wino4 alone; it also is the one kernel with a
  // per-channel PReLU epilogue, per-channel out_mask slopes and the second output)
  const bool pack_ok = al_in && al_ep && (d.K % 4 == 0) && (d.N % 4 == 0) && !d.in_mask && !d.in_prelu;
  const bool use_wino4 = pack_ok && d.w_wino4 && ((uintptr_t)d.w_wino4 % 16 == 0) && d.s2d_c == 0 &&
                         (d.act != NEOSR_ACT_PRELU || ((uintptr_t)d.prelu % 16 == 0 && d.prelu)) &&
                         (uintptr_t)d.out_mask_slopes % 16 == 0 && (uintptr_t)d.out2 % 16 == 0 && d.out2_cs % 4 == 0 &&
                         small(d.out2, d.out2_cs) && neosr_conv::wino_mode() == 2 &&
                         (w4_wgs >= NEOSR_WINO4_MIN_WGS || !use_wino) && small(d.out, d.out_cs) && small(d.res1, d.res1_cs) &&
                         small(d.res2, d.res2_cs) && small(d.out_mask, d.out_mask_cs) && small(d.in, d.in_cs);
  NEOSR_CHECK(use_wino4 || (!d.out2 && !d.out_mask_slopes && d.act != NEOSR_ACT_GELU && !d.out_mask_gelu),
              "conv3x3: out2 / out_mask_slopes / GELU need the F(4x4,3x3) kernel (w_wino4, winograd mode 2, aligned tensors)");
  NEOSR_CHECK(!d.out_mask_gelu || d.out_mask, "conv3x3: out_mask_gelu needs out_mask");
  const bool prof = neosr_prof_on();
  if (prof) {
    const double px = (double)d.B * d.H * d.W;
    // algorithmic traffic: read |x| + |W|, write |y| (fp32)
    neosr_prof_begin((d.mode == NEOSR_CONV_FWD ? NEOSR_PROF_CONV_FWD : NEOSR_PROF_CONV_DGRAD) + (use_pack || use_wino4 ? 0 : 4), stream,
                     // (a 4x4 / stride-2 layer run as a 3x3 over the space-to-depth tensor multiplies 16 of its 36 (tap,
                     // sub-pixel) blocks: the other 20 are structurally zero and skipped)
                     2.0 * px * d.K * d.N * (d.s2d_c > 0 ? 4.0 : 9.0),
                     4.0 * (px / (d.ups ? 4.0 : 1.0) * d.K + px * d.N + 9.0 * d.K * d.N));
    neosr_prof_algo(use_wino4 ? 2 : use_wino ? 1 : 0);
  }
  // thin layers (see conv3x3_thin_*_kernel)
  const bool plain_in = d.w && al_in && !d.ups && !d.in_prelu && !d.mask_slopes;
  const bool thin_k = d.w && !d.ups && !d.in_prelu && !d.mask_slopes && d.K <= 4 && !d.in_mask && al_ep &&
                      (d.N % 4 == 0);
  a.scalar_in = thin_k && !al_in;
  const bool thin_n = plain_in && d.N <= 4 && d.K >= 8 && (d.K % 4 == 0) && al_mk && !d.res1 && !d.res2 &&
                      !d.accumulate && !d.out_mask && d.act != NEOSR_ACT_PRELU;
  if (use_wino4) {
    const int rc = neosr_conv::launch_wino4_strips(d, stream);
    if (rc > 0) return rc;
    if (rc < 0) launch_wino4(a, st);
  } else if (use_wino) {
    launch_wino(a, st);
  } else if (use_pack) {
    launch_glds(a, grid, st);
  } else if (thin_k) {
    launch_thin_k(a, grid, st);
  } else if (thin_n) {
    launch_thin_n(a, st);
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

