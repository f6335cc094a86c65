// wgrad.hip — backward-weight of the 3x3/s1/p1 convolution for gfx950, one launch for up to
// 8 convolutions that share the same pixel grid (all 5 convs of a Residual Dense Block).
//
// GEMM view per (conv, cout-tile of 32, cin-tile of 32):  D[co][ci](tap) += G[p][co] * X[p+tap][ci]
//   M = 32 cout, N = 32 cin, 9 accumulators (one per tap), K = pixels, on
//   v_mfma_f32_32x32x2_f32 (exact fp32).
// A workgroup (4 waves) walks a strip of 4x32-pixel tiles.  Per tile the gradient tile
// G[128 px][32 co] and the input halo tile X[6x34 px][32 ci] are staged through LDS (lanes run
// along channels -> conflict-free ds_read_b32, no address VALU in the MFMA loop) while the next
// tile's global loads are already in flight in registers.  Wave w owns row w of every tile; the
// four waves' accumulators are summed through LDS in a fixed order and each workgroup writes one
// partial; a second kernel sums the partials over the pixel splits, again in a fixed order
// (no float atomics => run-to-run deterministic), applies `scale` and scatters into the canonical
// (Cout, Cin, 3, 3) layout.
//
// Reference call sites replaced: autograd's convolution_backward (weight/bias part) for
// neosr/archs/esrgan_arch.py:109-116,196-214 and neosr/archs/compact_arch.py:76-79.
#include "common.h"
#include "prof.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "../../include/neosr_amd.h"

namespace neosr_conv { bool xcd_enabled(); bool wino_enabled(); int wino_mode(); }

namespace {

#ifndef NEOSR_WG_UNROLL
#define NEOSR_WG_UNROLL 16  // the whole 32-pixel row: LDS reads scheduled across k-steps (300.8 vs 304.3 us per RDB launch)
#endif
constexpr int TH = 4, TW = 32;
constexpr int HALO_W = TW + 2, HALO_H = TH + 2;
constexpr int G_PIX = TH * TW;            // 128
constexpr int X_PIX = HALO_H * HALO_W;    // 204
constexpr int G_F4 = G_PIX * 8 / 256;     // 4 float4 per thread
constexpr int X_F4 = (X_PIX * 8 + 255) / 256;  // 7 float4 per thread
constexpr int G_LDS = G_PIX * 32;         // floats
constexpr int X_LDS = X_PIX * 32;
constexpr int WG_TILE = 9 * 32 * 32;      // partial tile, floats
constexpr int MAXD = NEOSR_WGRAD_MAX;

struct WgradMultiArgs {
  neosr_wgrad_desc d[MAXD];
  int pair_start[MAXD + 1];  // prefix sums of (nnt*nkt) per desc
  int nkt[MAXD];
  int vec_in[MAXD], vec_g[MAXD], vec_m[MAXD];
  int ndesc;
  int B, H, W, ups;
  int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit;
  float* part;    // [pair][split][WG_TILE]
  float* bpart;   // [desc-cout-tile][split][32]
  int btile_start[MAXD + 1];  // prefix sums of nnt per desc
  // XCD-pinned order (1-D grid of 8 * xcd_q workgroups, see plan()): 0 = (pair, split) grid in dispatch order
  int xcd, xcd_full, xcd_q;
  // Winograd path (conv3x3_wgrad_wino_kernel): units of 2 rows x 32 columns, its own split of them
  int w_units_x, w_units_y, w_nunits, w_units_per_split, w_nsplit;
  unsigned long long* timeline;  // debug only (NEOSR_TIMELINE builds)
};

__device__ __forceinline__ float4 ld4(const float* p, bool vec, int c, int C) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c >= C) return v;
  if (vec && c + 3 < C) return *reinterpret_cast<const float4*>(p);
  v.x = p[0];
  if (c + 1 < C) v.y = p[1];
  if (c + 2 < C) v.z = p[2];
  if (c + 3 < C) v.w = p[3];
  return v;
}

__device__ __attribute__((aligned(256))) float wg_zero_page[64];

// FAST: every descriptor has 16-byte aligned tensors, channel strides / K / N multiples of 4.
// PLAIN: no descriptor has a derivative mask or a PReLU on load (the gather-form RDB backward): the mask
// staging registers and the per-channel slope code disappear, which leaves the allocator room to keep the
// next k-step's LDS fragments in flight.
template <bool FAST, bool S2D = false, bool PLAIN = false>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_multi_kernel(const WgradMultiArgs args) {
  __shared__ __attribute__((aligned(16))) float lds[G_LDS + X_LDS];  // 42.5 KB (>= WG_TILE)
  __shared__ float bred[4 * 32];
  float* lg = lds;
  float* lx = lds + G_LDS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  // which conv / tile pair and which pixel split does this workgroup own?
  int pair = blockIdx.x, s = blockIdx.y;
  if (args.xcd) {
    // Workgroup b is observed to run on XCD b % 8 (a speed assumption only).  All pairs of one pixel split read the
    // same G / X pixels (each tile is staged by up to 6 pairs): whole splits are pinned to one XCD so that its L2
    // fetches them once; the splits left over after 8 * xcd_full are dealt in x-major order over the free slots.
    const int P = args.pair_start[MAXD];
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pinned = args.xcd_full * P;
    if (q < pinned) {
      s = x * args.xcd_full + q / P;
      pair = q % P;
    } else {
      const int r = x * (args.xcd_q - pinned) + (q - pinned);
      s = 8 * args.xcd_full + r / P;
      pair = r % P;
      if (s >= args.nsplit) return;
    }
  }
  int di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < args.ndesc && pair >= args.pair_start[i]) di = i;
  const neosr_wgrad_desc& d = args.d[di];
  const int local = pair - args.pair_start[di];
  const int nkt = args.nkt[di];
  const int ntile = local / nkt, kt = local - ntile * nkt;
  const int co0 = ntile * 32, ci0 = kt * 32;
  const bool vec_in = args.vec_in[di], vec_g = args.vec_g[di], vec_m = args.vec_m[di];

  const int t_lo = s * args.tiles_per_split;
  const int t_hi = min(args.ntiles, t_lo + args.tiles_per_split);

  const int H = args.H, W = args.W;
  const int Hin = args.ups ? (H >> 1) : H, Win = args.ups ? (W >> 1) : W;

  // staging slots: (pixel, 4-channel quad) per thread
  const int q4 = (tid & 7) << 2;  // channel quad within the 32-wide tile
  float4 rg[G_F4], rm[PLAIN ? 1 : G_F4], rx[X_F4];

  auto gload = [&](int t) {
    const int txi = t % args.tiles_x;
    const int r = t / args.tiles_x;
    const int tyi = r % args.tiles_y;
    const int b = r / args.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH;
#pragma unroll
    for (int i = 0; i < G_F4; ++i) {
      const int pix = (tid >> 3) + i * 32;  // 0..127
      const int py = pix >> 5, px = pix & 31;
      const int gy = y0 + py, gx = x0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(1.f, 1.f, 1.f, 1.f);
      if (FAST) {
        // branch-free: invalid lanes read the zero page (address-side redirect, see conv_mfma.hip)
        const bool ok = gy < H && gx < W && co0 + q4 < d.N;
        const int64_t p = ok ? ((int64_t)b * H + gy) * W + gx : 0;
        v = *reinterpret_cast<const float4*>(ok ? d.g + p * d.g_cs + co0 + q4 : wg_zero_page);
        if (!PLAIN && d.g_mask)
          m = *reinterpret_cast<const float4*>(ok ? d.g_mask + p * d.mask_cs + co0 + q4 : wg_zero_page);
      } else if (gy < H && gx < W) {
        const int64_t p = ((int64_t)b * H + gy) * W + gx;
        v = ld4(d.g + p * d.g_cs + co0 + q4, vec_g, co0 + q4, d.N);
        if (d.g_mask) m = ld4(d.g_mask + p * d.mask_cs + co0 + q4, vec_m, co0 + q4, d.N);
      }
      rg[i] = v;
      if (!PLAIN) rm[i] = m;
    }
#pragma unroll
    for (int i = 0; i < X_F4; ++i) {
      const int idx = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (FAST) {
        const int pix = idx >> 3;
        const int py = pix / HALO_W, px = pix - py * HALO_W;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = idx < X_PIX * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W && ci0 + q4 < d.K;
        const int sy = args.ups ? (gy >> 1) : gy, sx = args.ups ? (gx >> 1) : gx;
        const int64_t p = ok ? ((int64_t)b * Hin + sy) * Win + sx : 0;
        v = *reinterpret_cast<const float4*>(ok ? d.in + p * d.in_cs + ci0 + q4 : wg_zero_page);
      } else if (idx < X_PIX * 8) {
        const int pix = idx >> 3;
        const int py = pix / HALO_W, px = pix - py * HALO_W;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
          const int sy = args.ups ? (gy >> 1) : gy, sx = args.ups ? (gx >> 1) : gx;
          const int64_t p = ((int64_t)b * Hin + sy) * Win + sx;
          v = ld4(d.in + p * d.in_cs + ci0 + q4, vec_in, ci0 + q4, d.K);
        }
      }
      rx[i] = v;
    }
  };

  // per-channel slopes of this thread's quad (mask derivative / PReLU on load)
  float ms[4] = {d.mask_slope, d.mask_slope, d.mask_slope, d.mask_slope};
  float ps[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (d.mask_slopes && co0 + q4 + j < d.N) ms[j] = d.mask_slopes[co0 + q4 + j];
    if (d.in_prelu && ci0 + q4 + j < d.K) ps[j] = d.in_prelu[ci0 + q4 + j];
  }

  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < G_F4; ++i) {
      float4 v = rg[i];
      if (!PLAIN && d.g_mask) {
        const float4 m = rm[PLAIN ? 0 : i];
        v.x = m.x > 0.f ? v.x : v.x * ms[0];
        v.y = m.y > 0.f ? v.y : v.y * ms[1];
        v.z = m.z > 0.f ? v.z : v.z * ms[2];
        v.w = m.w > 0.f ? v.w : v.w * ms[3];
      }
      const int pix = (tid >> 3) + i * 32;
      *reinterpret_cast<float4*>(lg + pix * 32 + q4) = v;
    }
#pragma unroll
    for (int i = 0; i < X_F4; ++i) {
      const int idx = tid + i * 256;
      if (idx < X_PIX * 8) {
        float4 v = rx[i];
        if (!PLAIN && d.in_prelu) {
          v.x = v.x > 0.f ? v.x : v.x * ps[0];
          v.y = v.y > 0.f ? v.y : v.y * ps[1];
          v.z = v.z > 0.f ? v.z : v.z * ps[2];
          v.w = v.w > 0.f ? v.w : v.w * ps[3];
        }
        *reinterpret_cast<float4*>(lx + (idx >> 3) * 32 + q4) = v;
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  // 4x4 / stride-2 kernel as a 3x3 over the space-to-depth input: this cin tile lies in one sub-pixel
  // (dy, dx) and only meets block taps by in {1, dy ? 0 : 2}, bx in {1, dx ? 0 : 2}
  int tapmask = 0x1ff;
  if (S2D && d.s2d_c > 0 && d.s2d_c % 32 == 0) {
    const int sub = ci0 / d.s2d_c, r1 = (sub & 2) ? 0 : 2, c1 = (sub & 1) ? 0 : 2;
    tapmask = (1 << 4) | (1 << (3 + c1)) | (1 << (r1 * 3 + 1)) | (1 << (r1 * 3 + c1));
  }

  if (t_lo < t_hi) gload(t_lo);
  for (int t = t_lo; t < t_hi; ++t) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (t + 1 < t_hi) gload(t + 1);
    // wave `wave` owns row `wave` of the tile: 16 k-steps of 2 pixels
    const float* ga = lg + (wave * TW + lh) * 32 + l31;
    const float* xb = lx + (wave * HALO_W + lh) * 32 + l31;
#pragma unroll NEOSR_WG_UNROLL
    for (int ks = 0; ks < TW / 2; ++ks) {
      const float a = ga[ks * 64];
      bsum += a;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          if (S2D && !((tapmask >> (ty * 3 + tx)) & 1)) continue;  // workgroup-uniform; dense loop is branch-free
          const float bv = xb[(ty * HALO_W + ks * 2 + tx) * 32];
          acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[ty * 3 + tx], 0, 0, 0);
        }
    }
  }

  // fixed-order reduction of the 4 waves through LDS: tile[tap][co_i][ci_j]
  __syncthreads();
  float* red = lds;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
          float* p = red + t * 1024 + i * 32 + l31;
          *p = (w == 0) ? acc[t][r] : (*p + acc[t][r]);
        }
    }
    __syncthreads();
  }
  float* part = args.part + ((int64_t)pair * args.nsplit + s) * WG_TILE;
  for (int e = tid * 4; e < WG_TILE; e += 1024)
    *reinterpret_cast<float4*>(part + e) = *reinterpret_cast<const float4*>(red + e);

  if (d.db && kt == 0) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0) bred[wave * 32 + l31] = bsum;
    __syncthreads();
    if (tid < 32) {
      const float v = ((bred[tid] + bred[32 + tid]) + bred[64 + tid]) + bred[96 + tid];
      args.bpart[((int64_t)(args.btile_start[di] + ntile) * args.nsplit + s) * 32 + tid] = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Winograd form of the same weight gradient, F(3x3 taps <- 2x2 gradient tile (*) 4x4 input patch):
//   dW(3x3) = A'^T [ sum over tiles and batch of  (G' dy G'^T) (.) (B^T d B) ] A'
//   B^T as in conv_wino.hip; G' = [[1,0],[1,1],[1,-1],[0,-1]] (adds only); A'^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
// (the F(2x2,3x3) algorithm with the roles of filter and output exchanged: the trilinear form is symmetric in them).
// 16 multiplications per (tile, cout, cin) instead of 36.  A workgroup owns a (32 cout, 32 cin) pair and a split of
// the pixel tiles exactly like conv3x3_wgrad_multi_kernel (same XCD-pinned order, same bias partials); each of the 16
// transform positions is a 32 x 32 GEMM whose reduction index is the TILE.  WAVE i OWNS ROW i of both transforms:
//   * MFMA K = 2 tiles per instruction: lanes lh = 0 / 1 walk the left / right 8 tiles of a 2-row x 32-column unit, so
//     consecutive steps of a lane are ADJACENT tiles and the row-combined columns t2, t3 of one patch are t0, t1 of the
//     next: 4 new input floats + 2 (or 4) gradient floats per step, lane = channel -> conflict-free ds_read_b32;
//   * A operand = row i of G' dy G'^T for the lane's cout, B operand = row i of B^T d B for the lane's cin, both built in
//     registers from the raw tiles; nothing transformed is ever stored;
//   * raw tiles (4 x 34 input pixels x 32 cin, 2 x 32 gradient pixels x 32 cout) go global -> LDS with
//     buffer_load_dwordx4 ... lds into two 25 KB buffers, ONE barrier per unit.
// The 16 x 32 x 32 partial of a workgroup is summed over the splits in a fixed order and inverse-transformed by
// conv3x3_wgrad_wino_reduce_kernel (run-to-run deterministic, like the direct path).
constexpr int WW_XR = 4, WW_XC = 34;                 // raw input rows / columns of a unit
constexpr int WW_XGRAN = WW_XR * WW_XC * 8;          // 1088 16-byte granules (32 channels = 8 per pixel)
constexpr int WW_XROUNDS = (WW_XGRAN + 255) / 256;   // 5: four workgroup-wide DMA rounds + one of wave 0 alone (64 granules)
static_assert(WW_XGRAN == 4 * 256 + 64, "the last DMA round is exactly wave 0");
constexpr int WW_XF = WW_XGRAN * 4;                  // 4352 floats (17 KB)
constexpr int WW_GF = 2 * 32 * 32;                   // gradient tile floats (8 KB) = 2 DMA rounds
constexpr int WW_BUF = WW_XF + WW_GF;                // 6400 floats = 25 KB; two buffers -> three workgroups per CU
constexpr int WW_SLOTS = 768;                        // resident workgroups the split count aims at (3 per CU)
constexpr int WW_PART = 16 * 1024;                   // floats per (pair, split) partial

__global__ __launch_bounds__(256, 3) void conv3x3_wgrad_wino_kernel(const WgradMultiArgs args) {
  // two DISTINCT LDS objects: hipcc waits vmcnt(0) before any ds_read that may alias a pending LDS-DMA write, which
  // would serialise the next unit's DMA with this unit's compute if both buffers lived in one array
  __shared__ __attribute__((aligned(1024))) float ldsA[WW_BUF];
  __shared__ __attribute__((aligned(1024))) float ldsB[WW_BUF];
  __shared__ float bred[2 * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  int pair = blockIdx.x, s = blockIdx.y;
  if (args.xcd) {  // XCD-pinned order, see conv3x3_wgrad_multi_kernel
    const int P = args.pair_start[MAXD];
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pinned = args.xcd_full * P;
    if (q < pinned) {
      s = x * args.xcd_full + q / P;
      pair = q % P;
    } else {
      const int r = x * (args.xcd_q - pinned) + (q - pinned);
      s = 8 * args.xcd_full + r / P;
      pair = r % P;
      if (s >= args.w_nsplit) return;
    }
  }
  int di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < args.ndesc && pair >= args.pair_start[i]) di = i;
  const neosr_wgrad_desc& d = args.d[di];
  const int local = pair - args.pair_start[di];
  const int nkt = args.nkt[di];
  const int ntile = local / nkt, kt = local - ntile * nkt;
  const int co0 = ntile * 32, ci0 = kt * 32;
  // descriptor fields this loop needs, once, in registers (d is indexed dynamically: every use would be a scalar load)
  const float* __restrict__ d_in = d.in;
  const float* __restrict__ d_g = d.g;
  const int d_in_cs = d.in_cs, d_g_cs = d.g_cs, d_K = d.K, d_N = d.N;
  const int ups = args.ups, units_x = args.w_units_x, units_y = args.w_units_y;
  const int H = args.H, W = args.W;
  const int Hin = args.ups ? (H >> 1) : H, Win = args.ups ? (W >> 1) : W;  // input = nearest x2 of a (H/2, W/2) map
  const int u_lo = s * args.w_units_per_split;
  const int u_hi = min(args.w_nunits, u_lo + args.w_units_per_split);

  // DMA granules of this thread: input rounds i = 0..4: G = i*256 + tid -> pixel G >> 3 (row, col of the 4 x 34 tile),
  // channel quad G & 7; gradient rounds i = 0, 1: pixel (row 0..1, col 0..31).  Per unit only a scalar base offset
  // changes: the per-lane offsets relative to the unit's origin are fixed (with `ups`, (y0 - 1 + r) >> 1 = y0/2 +
  // ((r - 1) >> 1) because y0 is even), so a round costs two range checks, one select and one 64-bit add.
  const int q4 = (tid & 7) << 2;
  const bool ci_ok = ci0 + q4 < d_K, co_ok = co0 + q4 < d_N;
  int xr[WW_XROUNDS], xc[WW_XROUNDS], xrel[WW_XROUNDS], grel[2];
#pragma unroll
  for (int i = 0; i < WW_XROUNDS; ++i) {
    const int G = i * 256 + tid;
    const int pix = G >> 3;
    const int r = pix / WW_XC, c = pix - r * WW_XC;
    xr[i] = (G < WW_XGRAN && ci_ok) ? r - 1 : -100000;  // image row / column relative to the unit's first output pixel
    xc[i] = c - 1;
    const int ry = ups ? ((r - 1) >> 1) : r - 1, rx = ups ? ((c - 1) >> 1) : c - 1;
    xrel[i] = (ry * Win + rx) * d_in_cs + ci0 + q4;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pix = (i * 256 + tid) >> 3;
    grel[i] = ((pix >> 5) * W + (pix & 31)) * d_g_cs + co0 + q4;
  }

  // LDS-DMA through BUFFER loads: base = the whole tensor, lane offset = unit origin (scalar) + the lane's fixed relative
  // offset, and a granule outside the image / past K or N simply gets an out-of-range offset (the DMA writes zeros):
  // two unsigned range checks, one add and one select per instruction, no pointers, no zero page, no branches.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d_in), 0, args.B * Hin * Win * d_in_cs * 4, 0x00020000);
  const auto rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d_g), 0, args.B * H * W * d_g_cs * 4, 0x00020000);
  auto issue = [&](int u, float* xb) {
    const int xx = u % units_x;
    const int r = u / units_x;
    const int yy = r % units_y;
    const int b = r / units_y;
    const int x0 = xx * 32, y0 = yy * 2;
    float* gb = xb + WW_XF;
    const int xbase = (((b * Hin + (ups ? (y0 >> 1) : y0)) * Win + (ups ? (x0 >> 1) : x0)) * d_in_cs) * 4;
    const int gbase = (((b * H + y0) * W + x0) * d_g_cs) * 4;
#pragma unroll
    for (int i = 0; i < WW_XROUNDS; ++i) {
      if (i == WW_XROUNDS - 1 && wv != 0) break;  // granules 1024..1087
      // (xr = -100000 for unused slots / channels >= K: fails the unsigned row test)
      const bool ok = ((unsigned)(y0 + xr[i]) < (unsigned)H) & ((unsigned)(x0 + xc[i]) < (unsigned)W);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(xb + (i * 4 + wv) * 256), 16,
                                               ok ? xbase + xrel[i] * 4 : 0x7ffffff0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pix = (i * 256 + tid) >> 3;
      const bool ok = co_ok & (y0 + (pix >> 5) < H) & (x0 + (pix & 31) < W);  // (no short-circuit: one select)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (__attribute__((address_space(3))) void*)(gb + (i * 4 + wv) * 256), 16,
                                               ok ? gbase + grel[i] * 4 : 0x7ffffff0, 0, 0, 0);
    }
  };

  // this wave's row i = wave of the transforms, each as ONE fused multiply-add per element and an overall sign that is
  // applied to the accumulators once, at the store:
  //   input rows  x0 - x2 | x1 + x2 | -(x1 - x2) | x1 - x3      = sx * (x[ra] + sg * x[rb])
  //   gradient    g0      | g0 + g1 | g0 - g1    | -g1          = sr * (g[gr] + gq * g[1])   (gq = 0 for waves 0, 3: they
  //                                                                read row 1 with a zero coefficient rather than diverge)
  const int ra = wave == 0 ? 0 : 1, rb = wave == 3 ? 3 : 2;
  const float sg = (wave == 1) ? 1.f : -1.f;
  const int gr = wave == 3 ? 1 : 0;
  const float gq = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
  const bool flip = (wave == 2) != (wave == 3);  // sx * sr = -1 for waves 2 and 3

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const bool want_b = d.db && kt == 0;
  // (fp32 MFMAs and ordinary vector instructions share the SIMD's FMA lanes, so every vector instruction saved is
  // matrix time won: the row passes work on the PAIRS of adjacent columns that one ds_read2_b32 returns, with packed
  // 2-wide instructions — (t2, t3) and (r0, r1) one v_pk_fma_f32 each, (t0 - t2, t1 - t3) one v_pk_add_f32; the last
  // product is taken as r1 (x) (t1 - t3) and its accumulator's sign flipped at the store)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 sg2 = {sg, sg}, gq2 = {gq, gq};
  f32x2 bs2 = {0.f, 0.f};
  auto compute = [&](const float* buf) {
    const float* xb = buf + l31;
    const float* gb = buf + WW_XF + l31;
    const float* xa = xb + (ra * WW_XC + lh * 16) * 32;   // raw column of tile t, patch column s: 2 t + s, t = lh*8 + kt
    const float* xq = xb + (rb * WW_XC + lh * 16) * 32;
    const float* g0 = gb + (gr * 32 + lh * 16) * 32;
    const float* g1 = gb + (32 + lh * 16) * 32;
    auto pair = [](const float* p) { return f32x2{p[0], p[32]}; };
    f32x2 P = __builtin_elementwise_fma(sg2, pair(xq), pair(xa));  // (t0, t1)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x2 T = __builtin_elementwise_fma(sg2, pair(xq + (2 * k + 2) * 32), pair(xa + (2 * k + 2) * 32));  // (t2, t3)
      const f32x2 R = __builtin_elementwise_fma(gq2, pair(g1 + (2 * k) * 32), pair(g0 + (2 * k) * 32));          // (r0, r1)
      const f32x2 D = P - T;  // (t0 - t2, t1 - t3)
      bs2 += R;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.x, D.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.x + R.y, P.y + T.x, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.x - R.y, T.x - P.y, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(R.y, D.y, acc[3], 0, 0, 0);
      P = T;
    }
  };

  if (u_lo < u_hi) {
    issue(u_lo, ldsA);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    for (int u = u_lo; u < u_hi; u += 2) {
      if (u + 1 < u_hi) issue(u + 1, ldsB);
      compute(ldsA);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
      if (u + 1 >= u_hi) break;
      if (u + 2 < u_hi) issue(u + 2, ldsA);
      compute(ldsB);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
    }
  }

  // partial[pos = 4 wave + j][co row][ci column]: D rows = (r & 3) + 8 (r >> 2) + 4 lh, column = lane & 31
  float* part = args.part + ((int64_t)pair * args.w_nsplit + s) * WW_PART + (wave * 4) * 1024 + l31;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)  // (position 3 of the row was accumulated with the opposite sign)
      part[j * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 32] = (flip != (j == 3)) ? -acc[j][r] : acc[j][r];

  if (want_b) {  // sum of the gradient tile = row 0 sums (wave 0: r = g0) + row 1 sums (wave 3: r = +g1 here)
    float bsum = bs2.x + bs2.y;
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0 && (wave == 0 || wave == 3)) bred[(wave ? 32 : 0) + l31] = bsum;
    __syncthreads();
    if (tid < 32)
      args.bpart[((int64_t)(args.btile_start[di] + ntile) * args.w_nsplit + s) * 32 + tid] = bred[tid] + bred[32 + tid];
  }
}

// stage 2 of the Winograd path: sum the 16-position partials over the splits (fixed order), inverse transform
// A'^T M A', scale, scatter into the canonical (N, K, 3, 3) layout; bias gradient as in conv3x3_wgrad_reduce_kernel
__global__ __launch_bounds__(256) void conv3x3_wgrad_wino_reduce_kernel(const WgradMultiArgs args) {
  __shared__ float red[4][16][64];
  const int pair = blockIdx.y;
  int di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < args.ndesc && pair >= args.pair_start[i]) di = i;
  const neosr_wgrad_desc& d = args.d[di];
  const int local = pair - args.pair_start[di];
  const int nkt = args.nkt[di];
  const int ntile = local / nkt, kt = local - ntile * nkt;
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + o;  // < 1024: (co i, ci j) of the tile
  const int i = r >> 5, j = r & 31;
  const int co = ntile * 32 + i, ci = kt * 32 + j;
  const bool live = co < d.N && ci < d.K;
  float m[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) m[p] = 0.f;
  if (live) {
    const float* p0 = args.part + (int64_t)pair * args.w_nsplit * WW_PART + r;
    for (int s = sl; s < args.w_nsplit; s += 4) {
      const float* ps = p0 + (int64_t)s * WW_PART;
#pragma unroll
      for (int p = 0; p < 16; ++p) m[p] += ps[p * 1024];
    }
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) red[sl][p][o] = m[p];
  __syncthreads();
  if (sl == 0 && live) {
    float M[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) M[p >> 2][p & 3] = ((red[0][p][o] + red[1][p][o]) + red[2][p][o]) + red[3][p][o];
    float u[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[0][q] = M[0][q] + 0.5f * (M[1][q] + M[2][q]);
      u[1][q] = 0.5f * (M[1][q] - M[2][q]);
      u[2][q] = 0.5f * (M[1][q] + M[2][q]) + M[3][q];
    }
    float* qd = d.dw + ((int64_t)co * d.K + ci) * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float w0 = u[a][0] + 0.5f * (u[a][1] + u[a][2]);
      const float w1 = 0.5f * (u[a][1] - u[a][2]);
      const float w2 = 0.5f * (u[a][1] + u[a][2]) + u[a][3];
      const float v[3] = {w0 * d.scale, w1 * d.scale, w2 * d.scale};
#pragma unroll
      for (int b = 0; b < 3; ++b) qd[a * 3 + b] = d.accumulate ? qd[a * 3 + b] + v[b] : v[b];
    }
  }
  if (d.db && kt == 0 && blockIdx.x == 0) {
    __shared__ float bred[8][33];
    const int cbl = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const float* bp = args.bpart + (int64_t)(args.btile_start[di] + ntile) * args.w_nsplit * 32 + cbl;
    float sum = 0.f;
    for (int s = bl; s < args.w_nsplit; s += 8) sum += bp[(int64_t)s * 32];
    bred[bl][cbl] = sum;
    __syncthreads();
    const int cb = ntile * 32 + cbl;
    if (bl == 0 && cb < d.N) {
      float tot = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) tot += bred[l][cbl];
      tot *= d.scale;
      d.db[cb] = d.accumulate ? (d.db[cb] + tot) : tot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Winograd F(3x3 taps <- 4x4 gradient tile (*) 6x6 input patch) form of the weight gradient (round 3):
//   dW(3x3) = C [ sum over tiles and batch of (G'' dy G''^T) (.) (B^T d B) ] C^T
//   B^T: the 6x6 matrix of conv_wino4.hip (points 0, +-1, +-2, inf); G'' = rows (1,0,0,0), (1,1,1,1), (1,-1,1,-1),
//   (1,2,4,8), (1,-2,4,-8), (0,0,0,1) — the F(3,4) filter transform with its row scales s = (1/4, -1/6, -1/6, 1/24, 1/24, 1)
//   taken out; C = A'^T diag(s), A'^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1]  (checked against the direct sum in
//   float64: experiments/wgrad43_check.py).  36 multiplications per (4x4 tile, cout, cin) instead of 64 with the
//   F(2x2) form above and 144 direct.
// Same (pair, split) decomposition, XCD pinning and bias partials as the kernels above; MFMA reduction index = tile.
//   * the 36 positions are dealt 9 per wave: wave 0 = row 0 + (1, 0..2), wave 1 = row 2 + (1, 3..5), wave 2 = row 3 +
//     (4, 0..2), wave 3 = row 5 + (4, 3..5) — every wave runs the column pass for TWO rows of both transforms (waves 1, 2:
//     the row pairs {1,2} / {3,4} share their even / odd parts, 4 FMAs per element and row pair) and the row pass for one
//     full row and one half row; 9 accumulators of 16 registers;
//   * unit = 4 rows x 16 columns of gradient pixels = 4 tiles; lanes lh = 0 / 1 walk the left / right two tiles, so the
//     row-combined columns 4, 5 of a lane's first patch are columns 0, 1 of its second; lane = channel, and both raw
//     tiles are stored [row][column-in-half][half][32 channels] so a ds_read_b32 of the wave touches 64 consecutive
//     floats (6 x 10 x 2 input pixels x 32 cin — columns 8, 9 twice —, 4 x 8 x 2 gradient pixels x 32 cout; buffer-load
//     DMA, THREE 24 KB buffers, one barrier per unit);
//   * schedule of a step (one tile per lane half): the step's ds_reads, then the 9 products of the PREVIOUS step (the
//     reads land under them), then the packed-fp32 transforms.  On this SIMD the fp32 MFMA passes and the vector
//     instructions of both resident waves share one issue stream (experiments/ub: time = sum), so what counts is the
//     vector instruction total: ~45 per 9 products.  The six DMA rounds of the unit after next are issued between the
//     first products of a unit's first step (the texture path takes 16 cycles per round; issued in a block after the
//     barrier, the eight waves of the CU queued there with the matrix pipe idle).
//   * the inverse transform runs IN the kernel: each wave contracts its positions with C along j, the 8 row parts cross
//     LDS once per output column b, and the workgroup writes a plain 9-tap partial — 36 KB instead of the 64 KB of
//     16-position partials, summed over the splits by conv3x3_wgrad_reduce_kernel (fixed order: deterministic).
constexpr int W4_XF = 4 * 256 * 4;                    // input region: 4 workgroup-wide DMA rounds (16 KB), 960 granules used
constexpr int W4_XSLOTS = 6 * 10 * 2;                 // pixel slots of the input tile
constexpr int W4_GF = 4 * 16 * 32;                    // gradient tile floats (8 KB) = 2 DMA rounds
constexpr int W4_BUF = W4_XF + W4_GF;                 // 6144 floats = 24 KB; three buffers, two workgroups per CU (144 KB)
constexpr int W4_SLOTS = 512;
static_assert(W4_BUF >= 4096, "a buffer holds four 32 x 32 row parts of the exchange");

#ifdef NEOSR_TIMELINE  // debug builds: clock64 marks of workgroup (0, 0), see tools/timeline.py
#define W4_TL(slot)                                                                                  \
  do {                                                                                               \
    if (args.timeline && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)              \
      args.timeline[(threadIdx.x >> 6) * 64 + (slot)] = clock64();                                   \
  } while (0)
#define W4_TLU(j) do { const int it_ = (u - u_lo) / 3; if (it_ < 10) W4_TL(2 + 6 * it_ + (j)); } while (0)
#else
#define W4_TL(slot) do {} while (0)
#define W4_TLU(j) do {} while (0)
#endif

typedef int w4_i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA round (64 lanes x 16 bytes -> 1 KB at lds_addr) as inline assembly: with three unit buffers in flight the
// rounds of two units are outstanding at a barrier, and hipcc's wait-count pass (which cannot tell the generations of one
// buffer apart across the loop back-edge) turns every vmcnt(6) into vmcnt(0).  Hidden from it, the waits below stand as written.
__device__ __forceinline__ void w4_dma16(w4_i32x4 rsrc, unsigned lds_addr, int voff, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

typedef float w4_f32x2 __attribute__((ext_vector_type(2)));

template <bool EDGE, bool HI>
__device__ __forceinline__ void wgrad_w4_body(const WgradMultiArgs& args, const neosr_wgrad_desc& d, int di, int pair, int s,
                                              int ntile, int kt, float* ldsA, float* ldsB, float* ldsC, float* bred) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int co0 = ntile * 32, ci0 = kt * 32;
  const float* __restrict__ d_in = d.in;
  const float* __restrict__ d_g = d.g;
  const int d_in_cs = d.in_cs, d_g_cs = d.g_cs, d_K = d.K, d_N = d.N;
  const int ups = args.ups, units_x = args.w_units_x, units_y = args.w_units_y;
  const int H = args.H, W = args.W;
  const int Hin = ups ? (H >> 1) : H, Win = ups ? (W >> 1) : W;
  const int u_lo = s * args.w_units_per_split;
  const int u_hi = min(args.w_nunits, u_lo + args.w_units_per_split);

  // DMA granules.  Input rounds i = 0..3: granule G = i*256 + tid -> slot G >> 3 = 2 * (row * 10 + cc) + h (< 120, the
  // rest unused: zeros), image column cc + 8 h, channel quad G & 7; gradient rounds i = 0, 1: slot = 2 * (row * 8 + cc) + h.
  // A granule's byte offset inside its unit never changes (xvo / gvo; the unit's origin goes into the scalar offset), and
  // whether it falls outside the image depends only on which border the unit touches: bit 0 = above the first unit row,
  // 1 = below the image in the last unit row, 2 = left of the first unit column, 3 = right of the image in the last one,
  // 4 = every granule (units past the split's end load zeros).  Interior units issue the six loads with no vector work.
  const int q4 = (tid & 7) << 2;
  const bool ci_ok = ci0 + q4 < d_K, co_ok = co0 + q4 < d_N;
  const int ylast0 = (units_y - 1) * 4, xlast0 = (units_x - 1) * 16;
  const int xbias = (Win + 1) * d_in_cs * 4;  // the resource starts this many bytes early: offsets of row / column -1 stay >= 0
  constexpr int OOB = 0x7ffffff0;
  int xvo[4], xbits[4], gvo[2], gbits[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int slot = (i * 256 + tid) >> 3;
    const int h = slot & 1, rc = slot >> 1;
    const int r = rc / 10, c = rc - r * 10 + 8 * h;
    const int ry = ups ? ((r - 1) >> 1) : r - 1, rxx = ups ? ((c - 1) >> 1) : c - 1;
    xvo[i] = (slot < W4_XSLOTS && ci_ok) ? ((ry * Win + rxx) * d_in_cs + ci0 + q4) * 4 + xbias : OOB;
    xbits[i] = 16 | (r == 0 ? 1 : 0) | (ylast0 + r - 1 >= H ? 2 : 0) | (c == 0 ? 4 : 0) | (xlast0 + c - 1 >= W ? 8 : 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = (i * 256 + tid) >> 3;
    const int h = slot & 1, rc = slot >> 1;
    const int gy = rc >> 3, gx = (rc & 7) + 8 * h;
    gvo[i] = co_ok ? ((gy * W + gx) * d_g_cs + co0 + q4) * 4 : OOB;
    gbits[i] = 16 | (ylast0 + gy >= H ? 2 : 0) | (xlast0 + gx >= W ? 8 : 0);
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // raw buffer resources (base, stride 0, num_records bytes, DATA_FORMAT = 32 bit); the input's starts xbias bytes early
  auto make_rsrc = [](const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    return w4_i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), bytes, 0x00020000};
  };
  const w4_i32x4 rx = make_rsrc(reinterpret_cast<const char*>(d_in) - xbias, args.B * Hin * Win * d_in_cs * 4 + xbias);
  const w4_i32x4 rg = make_rsrc(d_g, args.B * H * W * d_g_cs * 4);
  auto lds_addr = [](const float* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)p; };
  // the unit the next issue() loads: (ixx, iyy, ib) walk x-fastest from u_lo
  int iu = u_lo, ixx, iyy, ib;
  {
    ixx = u_lo % units_x;
    const int r = u_lo / units_x;
    iyy = r % units_y;
    ib = r / units_y;
  }
  int dvo[6], dxbase = 0, dgbase = 0;  // the unit being loaded: per-granule voffsets, scalar origins
  auto issue_prep = [&]() {
    const int x0 = ixx * 16, y0 = iyy * 4;
    const int inval = (iyy == 0 ? 1 : 0) | (iyy == units_y - 1 ? 2 : 0) | (ixx == 0 ? 4 : 0) | (ixx == units_x - 1 ? 8 : 0) |
                      (iu < u_hi ? 0 : 16);
    dxbase = iu < u_hi ? (((ib * Hin + (ups ? (y0 >> 1) : y0)) * Win + (ups ? (x0 >> 1) : x0)) * d_in_cs) * 4 : 0;
    dgbase = iu < u_hi ? (((ib * H + y0) * W + x0) * d_g_cs) * 4 : 0;
    ++iu;
    if (++ixx == units_x) {
      ixx = 0;
      if (++iyy == units_y) { iyy = 0; ++ib; }
    }
    if (inval == 0) {  // no granule leaves the image
#pragma unroll
      for (int i = 0; i < 4; ++i) dvo[i] = xvo[i];
      dvo[4] = gvo[0];
      dvo[5] = gvo[1];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dvo[i] = (xbits[i] & inval) ? OOB : xvo[i];
      dvo[4] = (gbits[0] & inval) ? OOB : gvo[0];
      dvo[5] = (gbits[1] & inval) ? OOB : gvo[1];
    }
  };
  auto issue_one = [&](float* xb, int i) {  // granule round i of the prepared unit: 0..3 input, 4, 5 gradient
    if (i < 4) w4_dma16(rx, lds_addr(xb + (i * 4 + wv) * 256), dvo[i], dxbase);
    else w4_dma16(rg, lds_addr(xb + W4_XF + ((i - 4) * 4 + wv) * 256), dvo[i], dgbase);
  };

  // per-wave constants:
  //   inner waves (1: rows F = 2, H = 1; 2: F = 3, H = 4):  x: p = d4 + al d2, q = d3 + al d1, tF = p + gF q, tH = p + gH q
  //                                                          g: e = g0 + be g2, o = g1 + be g3, rF = e + dF o, rH = e + dH o
  //   edge waves (0: F = 0, H = 1; 3: F = 5, H = 4):          x: tF = 4 d[xa] - 5 d[xa + 2] + d[xa + 4]; tH as above
  //                                                          g: rF = g0 | g3; rH as above
  constexpr bool LOW = EDGE != HI;               // waves 0, 1
  constexpr int WV = EDGE ? (HI ? 3 : 0) : (HI ? 1 : 2);  // = wave
  constexpr float al = LOW ? -4.f : -1.f;
  constexpr float gH = LOW ? 1.f : -2.f;
  constexpr float gF = LOW ? -1.f : 2.f;         // (inner only)
  constexpr float be = LOW ? 1.f : 4.f;
  constexpr float dH = LOW ? 1.f : -2.f;
  constexpr float dF = LOW ? -1.f : 2.f;         // (inner only)
  constexpr int xa = HI ? 1 : 0;                 // (edge only) first row of the three-term row
  constexpr bool glast = HI;                     // (edge only) gradient row copied: 0 | 3

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const bool want_b = d.db && kt == 0;
  w4_f32x2 bsum2 = {0.f, 0.f};

  // The transforms run on column PAIRS in packed fp32 (v_pk_fma_f32): a ds_read2st64_b32 returns the columns (c, c + 1)
  // of one row in a register pair.
  auto ld2 = [](const float* p) { return w4_f32x2{p[0], p[64]}; };
  // MFMA operands of the step transformed last: the products are issued one step late, right after the next step's LDS
  // reads, so the reads' latency and the transforms of the co-resident wave run under the matrix pipe's 9 x 16 passes
  float uF[6], vF[6], uH[3], vH[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) uF[j] = vF[j] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) uH[j] = vH[j] = 0.f;
  // one step = one tile per lane half.  ST: 0 / 1 (compile time).  In step 0 the six DMA rounds of the unit after next go
  // out between the first products: the texture path takes 16 cycles per 1 KB round and all eight waves of the CU would
  // otherwise queue there right after the barrier with the matrix pipe idle.
  w4_f32x2 fP, hP;  // row-combined columns (4,5) of rows F / H: the next step's columns (0,1)
  auto step = [&](const float* buf, float* dma_buf, auto ST) {
    constexpr int st = decltype(ST)::value;
    const float* xs = buf + lane + st * 256;            // + (row * 10 + cc) * 64
    const float* gs = buf + W4_XF + lane + st * 256;    // + (row * 8 + cc) * 64
    // ---- LDS reads of this step: input column pairs (0,1) [first step], (2,3), (4,5) and the gradient tile's column pairs
    w4_f32x2 c1, c2, c3, c4, cA = {0.f, 0.f}, cB = {0.f, 0.f}, cC = {0.f, 0.f};   // (cA..cC: image-edge rows only)
    if (st == 0) {
      c1 = ld2(xs + 640); c2 = ld2(xs + 2 * 640); c3 = ld2(xs + 3 * 640); c4 = ld2(xs + 4 * 640);
      if (EDGE) { cA = ld2(xs + xa * 640); cB = ld2(xs + (xa + 2) * 640); cC = ld2(xs + (xa + 4) * 640); }
    }
    const w4_f32x2 q1 = ld2(xs + 128 + 640), q2 = ld2(xs + 128 + 2 * 640), q3 = ld2(xs + 128 + 3 * 640), q4 = ld2(xs + 128 + 4 * 640);
    const w4_f32x2 p1 = ld2(xs + 256 + 640), p2 = ld2(xs + 256 + 2 * 640), p3 = ld2(xs + 256 + 3 * 640), p4 = ld2(xs + 256 + 4 * 640);
    w4_f32x2 qA = q1, qB = q1, qC = q1, pA = p1, pB = p1, pC = p1;
    if (EDGE) {
      qA = ld2(xs + 128 + xa * 640); qB = ld2(xs + 128 + (xa + 2) * 640); qC = ld2(xs + 128 + (xa + 4) * 640);
      pA = ld2(xs + 256 + xa * 640); pB = ld2(xs + 256 + (xa + 2) * 640); pC = ld2(xs + 256 + (xa + 4) * 640);
    }
    const w4_f32x2 ga0 = ld2(gs), ga1 = ld2(gs + 512), ga2 = ld2(gs + 2 * 512), ga3 = ld2(gs + 3 * 512);
    const w4_f32x2 gb0 = ld2(gs + 128), gb1 = ld2(gs + 128 + 512), gb2 = ld2(gs + 128 + 2 * 512), gb3 = ld2(gs + 128 + 3 * 512);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the previous step's products (+ the DMA rounds)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uF[j], vF[j], acc[j], 0, 0, 0);
      if (st == 0) {
        issue_one(dma_buf, j);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[6 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uH[j], vH[j], acc[6 + j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- input: column pass (rows F and H of B^T d)
    auto xcol = [&](w4_f32x2 d1, w4_f32x2 d2, w4_f32x2 d3, w4_f32x2 d4, w4_f32x2 dA, w4_f32x2 dB, w4_f32x2 dC, w4_f32x2& f,
                    w4_f32x2& h) {
      const w4_f32x2 p = __builtin_elementwise_fma(w4_f32x2{al, al}, d2, d4), q = __builtin_elementwise_fma(w4_f32x2{al, al}, d1, d3);
      h = __builtin_elementwise_fma(w4_f32x2{gH, gH}, q, p);
      if (EDGE) f = __builtin_elementwise_fma(w4_f32x2{4.f, 4.f}, dA, __builtin_elementwise_fma(w4_f32x2{-5.f, -5.f}, dB, dC));
      else f = __builtin_elementwise_fma(w4_f32x2{gF, gF}, q, p);
    };
    w4_f32x2 fR, hR, fQ, hQ;
    if (st == 0) xcol(c1, c2, c3, c4, cA, cB, cC, fR, hR);
    else { fR = fP; hR = hP; }
    xcol(q1, q2, q3, q4, qA, qB, qC, fQ, hQ);
    xcol(p1, p2, p3, p4, pA, pB, pC, fP, hP);
    // ---- input: row passes.  v0 = 4 t0 - 5 t2 + t4, v5 = 4 t1 - 5 t3 + t5;  a = t4 - 4 t2, b = t3 - 4 t1: v1, v2 = a +- b;
    //      c = t4 - t2, d = t3 - t1: v3, v4 = c +- 2 d
    {
      const w4_f32x2 v05 = __builtin_elementwise_fma(w4_f32x2{4.f, 4.f}, fR, __builtin_elementwise_fma(w4_f32x2{-5.f, -5.f}, fQ, fP));
      const w4_f32x2 ac = __builtin_elementwise_fma(w4_f32x2{-4.f, -1.f}, fQ.xx, fP.xx);
      const w4_f32x2 bd = __builtin_elementwise_fma(w4_f32x2{-4.f, -1.f}, fR.yy, fQ.yy);
      const w4_f32x2 v13 = __builtin_elementwise_fma(w4_f32x2{1.f, 2.f}, bd, ac);
      const w4_f32x2 v24 = __builtin_elementwise_fma(w4_f32x2{-1.f, -2.f}, bd, ac);
      vF[0] = v05.x; vF[5] = v05.y; vF[1] = v13.x; vF[3] = v13.y; vF[2] = v24.x; vF[4] = v24.y;
    }
    if (!HI) {
      const float a = fmaf(-4.f, hQ.x, hP.x), bq = fmaf(-4.f, hR.y, hQ.y);
      const w4_f32x2 v12 = __builtin_elementwise_fma(w4_f32x2{1.f, -1.f}, w4_f32x2{bq, bq}, w4_f32x2{a, a});
      vH[0] = fmaf(4.f, hR.x, fmaf(-5.f, hQ.x, hP.x));
      vH[1] = v12.x; vH[2] = v12.y;
    } else {
      const float cc = hP.x - hQ.x, dd = hQ.y - hR.y;
      const w4_f32x2 v34 = __builtin_elementwise_fma(w4_f32x2{2.f, -2.f}, w4_f32x2{dd, dd}, w4_f32x2{cc, cc});
      vH[0] = v34.x; vH[1] = v34.y;
      vH[2] = fmaf(4.f, hR.y, fmaf(-5.f, hQ.y, hP.y));
    }
    // ---- gradient: column pass (rows F, H of G'' dy) for the tile's two column pairs, then the row passes
    auto gcol = [&](w4_f32x2 g0, w4_f32x2 g1, w4_f32x2 g2, w4_f32x2 g3, w4_f32x2& rf, w4_f32x2& rh) {
      const w4_f32x2 e = __builtin_elementwise_fma(w4_f32x2{be, be}, g2, g0), o = __builtin_elementwise_fma(w4_f32x2{be, be}, g3, g1);
      rh = __builtin_elementwise_fma(w4_f32x2{dH, dH}, o, e);
      rf = EDGE ? (glast ? g3 : g0) : __builtin_elementwise_fma(w4_f32x2{dF, dF}, o, e);
    };
    w4_f32x2 rFa, rFb, rHa, rHb;
    gcol(ga0, ga1, ga2, ga3, rFa, rHa);
    gcol(gb0, gb1, gb2, gb3, rFb, rHb);
    if (want_b) {  // bias gradient: wave w sums gradient row w (every wave holds the four raw rows)
      const w4_f32x2 mine = WV == 0 ? ga0 + gb0 : WV == 1 ? ga1 + gb1 : WV == 2 ? ga2 + gb2 : ga3 + gb3;
      bsum2 += mine;
    }
    {
      const w4_f32x2 eo1 = rFa + rFb, eo2 = __builtin_elementwise_fma(w4_f32x2{4.f, 4.f}, rFb, rFa);
      const w4_f32x2 u12 = __builtin_elementwise_fma(w4_f32x2{1.f, -1.f}, eo1.yy, eo1.xx);
      const w4_f32x2 u34 = __builtin_elementwise_fma(w4_f32x2{2.f, -2.f}, eo2.yy, eo2.xx);
      uF[0] = rFa.x; uF[1] = u12.x; uF[2] = u12.y; uF[3] = u34.x; uF[4] = u34.y; uF[5] = rFb.y;
    }
    if (!HI) {
      const w4_f32x2 eo1 = rHa + rHb;
      const w4_f32x2 u12 = __builtin_elementwise_fma(w4_f32x2{1.f, -1.f}, eo1.yy, eo1.xx);
      uH[0] = rHa.x; uH[1] = u12.x; uH[2] = u12.y;
    } else {
      const w4_f32x2 eo2 = __builtin_elementwise_fma(w4_f32x2{4.f, 4.f}, rHb, rHa);
      const w4_f32x2 u34 = __builtin_elementwise_fma(w4_f32x2{2.f, -2.f}, eo2.yy, eo2.xx);
      uH[0] = u34.x; uH[1] = u34.y; uH[2] = rHb.y;
    }
  };
  auto compute = [&](const float* buf, float* dma_buf) {
    issue_prep();
    step(buf, dma_buf, std::integral_constant<int, 0>{});
    step(buf, dma_buf, std::integral_constant<int, 1>{});
  };
  auto mac = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uF[j], vF[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[6 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uH[j], vH[j], acc[6 + j], 0, 0, 0);
  };

  // units run in pairs (buffer A, buffer B); a unit index past the split's end loads zeros (no contribution), which keeps
  // the loop a single straight-line body — the accumulators stay in place
  // three unit buffers: while unit u is read, unit u + 1 has landed or is landing and the rounds of unit u + 2 go out
  // (a DMA round needs ~2000 cycles from issue to LDS under load; with two buffers every barrier waited for it)
  W4_TL(0);
  if (u_lo < u_hi) {
    issue_prep();
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_one(ldsA, i);
    issue_prep();
#pragma unroll
    for (int i = 0; i < 6; ++i) issue_one(ldsB, i);
    __builtin_amdgcn_s_waitcnt(0x0f76);  // vmcnt(6): the first unit is in
    __syncthreads();
    W4_TL(1);
    for (int u = u_lo; u < u_hi; u += 3) {
      W4_TLU(0);
      compute(ldsA, ldsC);
      W4_TLU(1);
      __builtin_amdgcn_s_waitcnt(0x0f76);
      __syncthreads();
      W4_TLU(2);
      if (u + 1 < u_hi) compute(ldsB, ldsA);
      W4_TLU(3);
      __builtin_amdgcn_s_waitcnt(0x0f76);
      __syncthreads();
      W4_TLU(4);
      if (u + 2 < u_hi) compute(ldsC, ldsB);
      __builtin_amdgcn_s_waitcnt(0x0f76);
      __syncthreads();
      W4_TLU(5);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // the zero rounds of the units past the end: the exchange reuses the buffers
    __syncthreads();
  }
  mac();  // the last step's products
  W4_TL(62);

  // ---- inverse transform.  C[b][j] = A'^T[b][j] s_j:  b = 0: (1/4, -1/6, -1/6, 1/24, 1/24, 0)
  //                                                      b = 1: (0, -1/6, 1/6, 1/12, -1/12, 0)   b = 2: (0, -1/6, -1/6, 1/6, 1/6, 1)
  // row parts in LDS: part 2 w = full row of wave w, 2 w + 1 = its half row (rows 1 and 4 arrive in two halves)
  constexpr float S6 = 1.f / 6.f, S12 = 1.f / 12.f, S24 = 1.f / 24.f;
  float* outp = args.part + ((int64_t)pair * args.nsplit + s) * WG_TILE;
  float* pF = (wave < 2 ? ldsA : ldsB) + ((2 * wave) & 3) * 1024 + l31;
  float* pH = (wave < 2 ? ldsA : ldsB) + ((2 * wave + 1) & 3) * 1024 + l31;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const float c0 = b == 0 ? 0.25f : 0.f;
    const float c1 = -S6, c2 = b == 1 ? S6 : -S6;
    const float c3 = b == 0 ? S24 : (b == 1 ? S12 : S6), c4 = b == 0 ? S24 : (b == 1 ? -S12 : S6);
    const float c5 = b == 2 ? 1.f : 0.f;
    if (b) __syncthreads();  // the combine of column b - 1 is done reading
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      float xf = c1 * acc[1][r] + c2 * acc[2][r] + c3 * acc[3][r] + c4 * acc[4][r];
      if (b == 0) xf += c0 * acc[0][r];
      if (b == 2) xf += c5 * acc[5][r];
      float xh;
      if (!HI) {
        xh = c1 * acc[7][r] + c2 * acc[8][r];
        if (b == 0) xh += c0 * acc[6][r];
      } else {
        xh = c3 * acc[6][r] + c4 * acc[7][r];
        if (b == 2) xh += c5 * acc[8][r];
      }
      pF[i * 32] = xf;
      pH[i * 32] = xh;
    }
    __syncthreads();
    {
      // rows: X0 = part 0, X1 = parts 1 + 3, X2 = part 2 (ldsA);  X3 = part 4, X4 = parts 5 + 7, X5 = part 6 (ldsB)
      const int e = tid * 4;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(ldsA + e);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(ldsA + 1024 + e) + *reinterpret_cast<const f32x4*>(ldsA + 3072 + e);
      const f32x4 x2 = *reinterpret_cast<const f32x4*>(ldsA + 2048 + e);
      const f32x4 x3 = *reinterpret_cast<const f32x4*>(ldsB + e);
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(ldsB + 1024 + e) + *reinterpret_cast<const f32x4*>(ldsB + 3072 + e);
      const f32x4 x5 = *reinterpret_cast<const f32x4*>(ldsB + 2048 + e);
      const f32x4 s12 = x1 + x2, d12 = x2 - x1, s34 = x3 + x4, d34 = x3 - x4;
      const f32x4 w0 = 0.25f * x0 - S6 * s12 + S24 * s34;
      const f32x4 w1v = S6 * d12 + S12 * d34;
      const f32x4 w2 = S6 * (s34 - s12) + x5;
      *reinterpret_cast<f32x4*>(outp + (0 * 3 + b) * 1024 + e) = w0;
      *reinterpret_cast<f32x4*>(outp + (1 * 3 + b) * 1024 + e) = w1v;
      *reinterpret_cast<f32x4*>(outp + (2 * 3 + b) * 1024 + e) = w2;
    }
  }

  if (want_b) {
    float bsum = bsum2.x + bsum2.y;
    bsum += __shfl_xor(bsum, 32, 64);
    if (lh == 0) bred[WV * 32 + l31] = bsum;
    __syncthreads();
    if (tid < 32)
      args.bpart[((int64_t)(args.btile_start[di] + ntile) * args.nsplit + s) * 32 + tid] =
          (bred[tid] + bred[32 + tid]) + (bred[64 + tid] + bred[96 + tid]);
  }
  W4_TL(63);
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_wino4_kernel(const WgradMultiArgs args) {
  __shared__ __attribute__((aligned(1024))) float ldsA[W4_BUF];
  __shared__ __attribute__((aligned(1024))) float ldsB[W4_BUF];
  __shared__ __attribute__((aligned(1024))) float ldsC[W4_BUF];
  __shared__ float bred[128];
  int pair = blockIdx.x, s = blockIdx.y;
  if (args.xcd) {  // XCD-pinned order, see conv3x3_wgrad_multi_kernel
    const int P = args.pair_start[MAXD];
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int pinned = args.xcd_full * P;
    if (q < pinned) {
      s = x * args.xcd_full + q / P;
      pair = q % P;
    } else {
      const int r = x * (args.xcd_q - pinned) + (q - pinned);
      s = 8 * args.xcd_full + r / P;
      pair = r % P;
      if (s >= args.nsplit) return;
    }
  }
  int di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < args.ndesc && pair >= args.pair_start[i]) di = i;
  const neosr_wgrad_desc& d = args.d[di];
  const int local = pair - args.pair_start[di];
  const int nkt = args.nkt[di];
  const int ntile = local / nkt, kt = local - ntile * nkt;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // wave 0: edge rows (0, 1 lo), 1: inner (2, 1 hi), 2: inner (3, 4 lo), 3: edge (5, 4 hi)
  if (wave == 0) wgrad_w4_body<true, false>(args, d, di, pair, s, ntile, kt, ldsA, ldsB, ldsC, bred);
  else if (wave == 1) wgrad_w4_body<false, true>(args, d, di, pair, s, ntile, kt, ldsA, ldsB, ldsC, bred);
  else if (wave == 2) wgrad_w4_body<false, false>(args, d, di, pair, s, ntile, kt, ldsA, ldsB, ldsC, bred);
  else wgrad_w4_body<true, true>(args, d, di, pair, s, ntile, kt, ldsA, ldsB, ldsC, bred);
}

// ---------------------------------------------------------------------------------------------
// Thin layers (3 or 1 channels on one side: conv_first / conv_last, U-Net conv0 / conv9): the 32 x 32
// tile above would compute 10x zeros.  Here v_mfma_f32_4x4x1_16b_f32 takes ONE pixel per instruction:
// the 4 rows of every 4x4 block are the thin side's channels t, the 64 columns (16 blocks x 4) are 64
// channels c of the wide side -> lane = wide channel (coalesced 256-byte row loads straight from
// global, no LDS), 9 accumulators of 4 registers = D[tap][t] for channel c.  XWIDE: wide = input x
// (shifted by the tap), thin = g (conv_last, N <= 4); else wide = g, thin = x shifted (conv_first, K <= 4).
// Waves stride over 64-pixel row segments; a workgroup sums its 4 waves through LDS in a fixed order and
// writes one partial; conv3x3_wgrad_thin_reduce_kernel sums the partials in index order.
constexpr int THIN_WGS = 512;
constexpr int THIN_PART = 9 * 4 * 64;  // floats per (workgroup, 64-channel group)
constexpr int THIN_UNROLL = 4;

struct ThinArgs {
  neosr_wgrad_desc d;
  int xwide;        // 1: wide = x (K channels), thin = g (N <= 4);  0: wide = g (N), thin = x (K <= 4)
  int wide_c, thin_c;
  int segs_per_row, nseg;
  float* part;      // [group][THIN_WGS][THIN_PART]
  float* bpart;     // [THIN_WGS][64 * groups] (wide = g) or [THIN_WGS][4] (thin = g)
  int groups;
};

template <bool XWIDE>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_thin_kernel(const ThinArgs a) {
  __shared__ float red[THIN_PART];
  __shared__ float bred[4][64];
  const neosr_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.y;
  const int c = grp * 64 + lane;            // wide channel of this lane
  const int t = lane & 3;                   // thin channel this lane feeds as the A operand
  const bool c_ok = c < a.wide_c, t_ok = t < a.thin_c;
  const int H = d.H, W = d.W;
  const float* __restrict__ wide = XWIDE ? d.in : d.g;
  const float* __restrict__ thin = XWIDE ? d.g : d.in;
  const int wide_cs = XWIDE ? d.in_cs : d.g_cs, thin_cs = XWIDE ? d.g_cs : d.in_cs;
  const bool masked = !XWIDE && d.g_mask != nullptr;

  f32x4 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int nwaves = gridDim.x * 4;
  for (int seg = blockIdx.x * 4 + wave; seg < a.nseg; seg += nwaves) {
    const int xs = seg % a.segs_per_row;
    const int r = seg / a.segs_per_row;
    const int y = r % H, b = r / H;
    const int x_lo = xs * 64, x_hi = min(W, x_lo + 64);
    // 32-bit element offsets from clamped coordinates (tensors < 2^31 floats, checked on the host);
    // out-of-image taps are zeroed on the value: one multiply-add per address instead of 64-bit chains
    const unsigned cw = c_ok ? c : 0, tw = t_ok ? t : 0;
    unsigned shrow[3];   // row offsets of the SHIFTED operand (x), per ky
    bool rowok[3];
    const unsigned sh_cs = XWIDE ? wide_cs : thin_cs, sh_ch = XWIDE ? cw : tw;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      rowok[ky] = yy >= 0 && yy < H;
      shrow[ky] = (unsigned)((b * H + min(max(yy, 0), H - 1)) * W) * sh_cs + sh_ch;
    }
    const unsigned fxrow = (unsigned)((b * H + y) * W) * (XWIDE ? thin_cs : wide_cs) + (XWIDE ? tw : cw);
    const float* __restrict__ shp = XWIDE ? wide : thin;   // shifted operand = the layer input x
    const float* __restrict__ fxp = XWIDE ? thin : wide;   // unshifted operand = g
    const bool fx_lane_ok = XWIDE ? t_ok : c_ok;   // (lanes outside the shifted operand's range produce rows / columns nobody stores)
    for (int x = x_lo; x < x_hi; x += THIN_UNROLL) {
      float wv[THIN_UNROLL][9], tv[THIN_UNROLL][9];
      // every load of the batch of pixels first (global latency >> 72 cycles of MFMA per pixel)
#pragma unroll
      for (int u = 0; u < THIN_UNROLL; ++u) {
        const int xx = x + u;
        const bool in_seg = xx < x_hi;
        const unsigned xc = (unsigned)min(xx, W - 1);
        float fx = fxp[fxrow + xc * (XWIDE ? thin_cs : wide_cs)];
        if (masked) {
          const float m = d.g_mask[(unsigned)((b * H + y) * W + xc) * d.mask_cs + cw];
          fx = m > 0.f ? fx : fx * d.mask_slope;
        }
        fx = (in_seg && fx_lane_ok) ? fx : 0.f;
        // the nine shifted loads are unconditional (clamped coordinates -> finite image data, no select
        // behind a load); an out-of-image tap is cancelled on the OTHER operand, which was loaded once
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int xq = xx + k % 3 - 1;
          const float v = shp[shrow[k / 3] + (unsigned)min(max(xq, 0), W - 1) * sh_cs];
          const float f = (rowok[k / 3] && xq >= 0 && xq < W) ? fx : 0.f;
          wv[u][k] = XWIDE ? v : f;
          tv[u][k] = XWIDE ? f : v;
        }
      }
#pragma unroll
      for (int u = 0; u < THIN_UNROLL; ++u) {
        bsum += XWIDE ? tv[u][4] : wv[u][4];  // centre tap: always inside the image
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(tv[u][k], wv[u][k], acc[k], 0, 0, 0);
      }
    }
  }
  // fixed-order sum of the 4 waves: red[(tap*4 + t)*64 + lane]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float* p = red + (k * 4 + v) * 64 + lane;
          *p = (w == 0) ? acc[k][v] : (*p + acc[k][v]);
        }
    }
    __syncthreads();
  }
  float* part = a.part + ((int64_t)grp * gridDim.x + blockIdx.x) * THIN_PART;
  for (int e = tid; e < THIN_PART; e += 256) part[e] = red[e];
  if (d.db) {
    bred[wave][lane] = bsum;
    __syncthreads();
    if (wave == 0) {
      const float v = ((bred[0][lane] + bred[1][lane]) + bred[2][lane]) + bred[3][lane];
      if (XWIDE) {
        if (grp == 0 && lane < 4) a.bpart[(int64_t)blockIdx.x * 4 + lane] = v;   // lanes 0..3 hold t = 0..3
      } else {
        a.bpart[(int64_t)blockIdx.x * (64 * a.groups) + c] = v;
      }
    }
  }
}

// dw[n][k][tap] (+)= scale * sum over workgroups, db[n] likewise (outputs N*K*9 .. N*K*9 + N - 1): 16 outputs
// per workgroup, 16 split lanes each walking every 16th partial, combined through LDS in a fixed order
__global__ __launch_bounds__(256) void conv3x3_wgrad_thin_reduce_kernel(const ThinArgs a, int nwg) {
  __shared__ float red[16][17];
  const neosr_wgrad_desc& d = a.d;
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const int total = d.N * d.K * 9;
  const bool is_w = idx < total, is_b = !is_w && d.db && idx < total + d.N;
  const float* p = nullptr;
  int64_t stride = 0;
  float* q = nullptr;
  if (is_w) {
    const int tap = idx % 9, k = (idx / 9) % d.K, n = idx / (9 * d.K);
    const int cw = a.xwide ? k : n, tt = a.xwide ? n : k;   // wide channel, thin channel
    p = a.part + (int64_t)(cw >> 6) * nwg * THIN_PART + (tap * 4 + tt) * 64 + (cw & 63);
    stride = THIN_PART;
    q = d.dw + idx;  // canonical (N, K, 3, 3) is exactly n*K*9 + k*9 + tap
  } else if (is_b) {
    const int nb = idx - total;
    stride = a.xwide ? 4 : 64 * a.groups;
    p = a.bpart + nb;
    q = d.db + nb;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (p) {
    int w = sl;
    for (; w + 48 < nwg; w += 64) {
      s0 += p[(int64_t)w * stride];
      s1 += p[(int64_t)(w + 16) * stride];
      s2 += p[(int64_t)(w + 32) * stride];
      s3 += p[(int64_t)(w + 48) * stride];
    }
    for (; w < nwg; w += 16) s0 += p[(int64_t)w * stride];
  }
  red[sl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && p) {
    float sum = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) sum += red[l][o];
    sum *= d.scale;
    *q = d.accumulate ? (*q + sum) : sum;
  }
}

// stage 2: sum partials over splits, scatter into canonical (N,K,3,3).  A workgroup owns 64 outputs;
// four split lanes walk the partials s = lane, lane + 4, ... and are combined through LDS in a fixed order
// (run-to-run deterministic); launches with few tile pairs carry > 100 splits, which one thread per
// output would walk as a serial chain of dependent-latency loads.
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const WgradMultiArgs args) {
  __shared__ float red[4][64];
  const int pair = blockIdx.y;
  int di = 0;
#pragma unroll
  for (int i = 1; i < MAXD; ++i)
    if (i < args.ndesc && pair >= args.pair_start[i]) di = i;
  const neosr_wgrad_desc& d = args.d[di];
  const int local = pair - args.pair_start[di];
  const int nkt = args.nkt[di];
  const int ntile = local / nkt, kt = local - ntile * nkt;
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + o;  // < WG_TILE
  const int tap = r >> 10, i = (r >> 5) & 31, j = r & 31;
  const int co = ntile * 32 + i, ci = kt * 32 + j;
  const bool live = co < d.N && ci < d.K;
  {
    const float* p = args.part + (int64_t)pair * args.nsplit * WG_TILE + r;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = sl;
    if (live) {
      for (; s + 12 < args.nsplit; s += 16) {
        s0 += p[(int64_t)s * WG_TILE];
        s1 += p[(int64_t)(s + 4) * WG_TILE];
        s2 += p[(int64_t)(s + 8) * WG_TILE];
        s3 += p[(int64_t)(s + 12) * WG_TILE];
      }
      for (; s < args.nsplit; s += 4) s0 += p[(int64_t)s * WG_TILE];
    }
    red[sl][o] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (sl == 0 && live) {
    float sum = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    sum *= d.scale;
    float* q = d.dw + ((int64_t)co * d.K + ci) * 9 + tap;
    *q = d.accumulate ? (*q + sum) : sum;
  }
  if (d.db && kt == 0 && blockIdx.x == 0) {  // bias gradient of this cout tile: 32 outputs x 8 split lanes
    __shared__ float bred[8][33];
    const int cbl = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const float* bp = args.bpart + (int64_t)(args.btile_start[di] + ntile) * args.nsplit * 32 + cbl;
    float sum = 0.f;
    for (int s = bl; s < args.nsplit; s += 8) sum += bp[(int64_t)s * 32];
    bred[bl][cbl] = sum;
    __syncthreads();
    const int cb = ntile * 32 + cbl;
    if (bl == 0 && cb < d.N) {
      float tot = 0.f;
#pragma unroll
      for (int l = 0; l < 8; ++l) tot += bred[l][cbl];
      tot *= d.scale;
      d.db[cb] = d.accumulate ? (d.db[cb] + tot) : tot;
    }
  }
}

int plan(const neosr_wgrad_desc* ds, int n, WgradMultiArgs& a) {
  NEOSR_CHECK(ds && n >= 1 && n <= MAXD, "wgrad: need 1..%d descriptors", MAXD);
  a.ndesc = n;
  a.B = ds[0].B; a.H = ds[0].H; a.W = ds[0].W; a.ups = ds[0].ups;
  NEOSR_CHECK(a.B > 0 && a.H > 0 && a.W > 0, "wgrad: bad geometry");
  NEOSR_CHECK(!a.ups || ((a.H % 2 == 0) && (a.W % 2 == 0)), "wgrad: ups needs even H,W");
  int pairs = 0, btiles = 0;
  for (int i = 0; i < n; ++i) {
    const neosr_wgrad_desc& d = ds[i];
    NEOSR_CHECK(d.B == a.B && d.H == a.H && d.W == a.W && d.ups == a.ups,
                "wgrad: descriptors of one launch must share B,H,W,ups");
    NEOSR_CHECK(d.in && d.g && d.dw && d.K > 0 && d.N > 0, "wgrad: null tensor / bad K,N");
    a.d[i] = d;
    a.pair_start[i] = pairs;
    a.btile_start[i] = btiles;
    a.nkt[i] = ceil_div(d.K, 32);
    pairs += ceil_div(d.N, 32) * a.nkt[i];
    btiles += ceil_div(d.N, 32);
    a.vec_in[i] = (d.in_cs % 4 == 0) && ((uintptr_t)d.in % 16 == 0);
    a.vec_g[i] = (d.g_cs % 4 == 0) && ((uintptr_t)d.g % 16 == 0);
    a.vec_m[i] = d.g_mask && (d.mask_cs % 4 == 0) && ((uintptr_t)d.g_mask % 16 == 0);
  }
  for (int i = n; i <= MAXD; ++i) { a.pair_start[i] = pairs; a.btile_start[i] = btiles; }
  a.tiles_x = ceil_div(a.W, TW);
  a.tiles_y = ceil_div(a.H, TH);
  a.ntiles = a.tiles_x * a.tiles_y * a.B;
  // one resident round of workgroups (2 per CU on 256 CUs)
  int nsplit = 512 / pairs;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  a.tiles_per_split = ceil_div(a.ntiles, nsplit);
  a.nsplit = ceil_div(a.ntiles, a.tiles_per_split);
  // XCD-pinned order: an XCD has 32 CUs x 2 resident workgroups = 64 slots
  a.xcd = neosr_conv::xcd_enabled() ? 1 : 0;
  a.xcd_full = pairs <= 64 ? 64 / pairs : 0;
  if (a.xcd_full > a.nsplit / 8) a.xcd_full = a.nsplit / 8;
  a.xcd_q = a.xcd_full * pairs + ceil_div((a.nsplit - 8 * a.xcd_full) * pairs, 8);
  a.w_units_x = ceil_div(a.W, 32);
  a.w_units_y = ceil_div(a.H, 2);
  a.w_nunits = a.w_units_x * a.w_units_y * a.B;
  int wsplit = WW_SLOTS / pairs;
  if (wsplit < 1) wsplit = 1;
  if (wsplit > a.w_nunits) wsplit = a.w_nunits;
  a.w_units_per_split = ceil_div(a.w_nunits, wsplit);
  a.w_nsplit = ceil_div(a.w_nunits, a.w_units_per_split);
  return 0;
}

// single thin layer without PReLU-on-load / per-channel slopes / upsampling: the 4x4x1 path
bool thin_ok(const neosr_wgrad_desc* ds, int n) {
  if (n != 1) return false;
  const neosr_wgrad_desc& d = ds[0];
  if (d.ups || d.in_prelu || d.mask_slopes) return false;
  const int64_t px = (int64_t)d.B * d.H * d.W;  // the kernel addresses with 32-bit element offsets
  if (px * d.in_cs >= (1ll << 31) || px * d.g_cs >= (1ll << 31) || px * (d.g_mask ? d.mask_cs : 0) >= (1ll << 31))
    return false;
  if (d.N <= 4 && d.K > 4) return !d.g_mask;   // thin = g
  if (d.K <= 4 && d.N > 4) return true;        // thin = x (a scalar-slope mask on g is handled)
  return false;
}

int64_t thin_ws_floats(const neosr_wgrad_desc& d) {
  const int wide = d.N <= 4 ? d.K : d.N;
  const int groups = ceil_div(wide, 64);
  return (int64_t)groups * THIN_WGS * THIN_PART + (int64_t)THIN_WGS * 64 * groups + 64;
}

// unit / split geometry of conv3x3_wgrad_wino4_kernel (units of 4 rows x 16 columns); overwrites nsplit
void w4_geometry(WgradMultiArgs& w) {
  const int P = w.pair_start[MAXD];
  w.w_units_x = ceil_div(w.W, 16);
  w.w_units_y = ceil_div(w.H, 4);
  w.w_nunits = w.w_units_x * w.w_units_y * w.B;
  int wsplit = W4_SLOTS / P;
  if (wsplit < 1) wsplit = 1;
  if (wsplit > w.w_nunits) wsplit = w.w_nunits;
  w.w_units_per_split = ceil_div(w.w_nunits, wsplit);
  w.nsplit = ceil_div(w.w_nunits, w.w_units_per_split);
}

int g_wgrad4 = -1;  // F(4x4-tile) weight gradient: -1 = read NEOSR_AMD_WGRAD4 (default on)
bool wgrad4_enabled() {
  if (g_wgrad4 < 0) {
    const char* e = getenv("NEOSR_AMD_WGRAD4");
    g_wgrad4 = (e && e[0] == '0') ? 0 : 1;
  }
  return g_wgrad4 == 1;
}

int64_t ws_floats(const WgradMultiArgs& a) {
  int64_t w = (int64_t)a.pair_start[MAXD] * a.nsplit * WG_TILE + (int64_t)a.btile_start[MAXD] * a.nsplit * 32 + 64;
  {
    WgradMultiArgs w4 = a;
    w4_geometry(w4);
    const int64_t t = (int64_t)a.pair_start[MAXD] * w4.nsplit * WG_TILE + (int64_t)a.btile_start[MAXD] * w4.nsplit * 32 + 64;
    if (t > w) w = t;
  }
  const int64_t ww = (int64_t)a.pair_start[MAXD] * a.w_nsplit * WW_PART + (int64_t)a.btile_start[MAXD] * a.w_nsplit * 32 + 64;
  if (ww > w) w = ww;  // the Winograd path keeps 16 positions per partial
  if (thin_ok(a.d, a.ndesc)) {
    const int64_t t = thin_ws_floats(a.d[0]);
    if (t > w) w = t;
  }
  return w;
}

}  // namespace

unsigned long long* g_wgrad_timeline = nullptr;
extern "C" int neosr_debug_set_wgrad_timeline(void* dev_buf) {
  g_wgrad_timeline = (unsigned long long*)dev_buf;
  return 0;
}

extern "C" int neosr_set_wgrad4(int on) {
  const int prev = wgrad4_enabled() ? 1 : 0;
  g_wgrad4 = on ? 1 : 0;
  return prev;
}

extern "C" int64_t neosr_conv3x3_wgrad_multi_workspace_bytes(const neosr_wgrad_desc* ds, int32_t n) {
  WgradMultiArgs a;
  if (plan(ds, n, a)) return -1;
  return ws_floats(a) * 4;
}

extern "C" int neosr_conv3x3_wgrad_multi(const neosr_wgrad_desc* ds, int32_t n, float* workspace,
                                         void* stream) {
  WgradMultiArgs a;
  if (int rc = plan(ds, n, a)) return rc;
  NEOSR_CHECK(workspace && (uintptr_t)workspace % 16 == 0, "wgrad: workspace missing/unaligned");
  a.part = workspace;
  a.bpart = workspace + (int64_t)a.pair_start[MAXD] * a.nsplit * WG_TILE;
  hipStream_t st = (hipStream_t)stream;
  const bool prof = neosr_prof_on();
  if (prof) {
    double fl = 0, by = 0;
    const double px = (double)a.B * a.H * a.W;
    for (int i = 0; i < n; ++i) {
      fl += 2.0 * px * ds[i].K * ds[i].N * (ds[i].s2d_c > 0 ? 4.0 : 9.0);   // (space-to-depth layers: 4 live taps per sub-pixel)
      by += 4.0 * (px * ds[i].N + px / (a.ups ? 4.0 : 1.0) * ds[i].K + 9.0 * ds[i].K * ds[i].N);
    }
    neosr_prof_begin(NEOSR_PROF_CONV_WGRAD, stream, fl, by);
  }
  if (thin_ok(ds, n)) {
    ThinArgs t;
    memset(&t, 0, sizeof(t));
    t.d = ds[0];
    t.xwide = ds[0].N <= 4;
    t.wide_c = t.xwide ? ds[0].K : ds[0].N;
    t.thin_c = t.xwide ? ds[0].N : ds[0].K;
    t.groups = ceil_div(t.wide_c, 64);
    t.segs_per_row = ceil_div(a.W, 64);
    t.nseg = t.segs_per_row * a.H * a.B;
    int nwg = ceil_div(t.nseg, 4);
    if (nwg > THIN_WGS) nwg = THIN_WGS;
    t.part = workspace;
    t.bpart = workspace + (int64_t)t.groups * THIN_WGS * THIN_PART;
    if (t.xwide)
      hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel<true>, dim3(nwg, t.groups), dim3(256), 0, st, t);
    else
      hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel<false>, dim3(nwg, t.groups), dim3(256), 0, st, t);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv3x3_wgrad_thin_reduce_kernel, dim3(ceil_div(ds[0].N * ds[0].K * 9 + ds[0].N, 16)), dim3(256), 0, st,
                       t, nwg);
    NEOSR_LAUNCH_CHECK();
    return 0;
  }
  bool fast = true;
  for (int i = 0; i < n; ++i)
    fast = fast && a.vec_in[i] && a.vec_g[i] && (!ds[i].g_mask || a.vec_m[i]) &&
           (ds[i].K % 4 == 0) && (ds[i].N % 4 == 0);
  bool s2d = false, plain = true;
  for (int i = 0; i < n; ++i) {
    s2d = s2d || ds[i].s2d_c > 0;
    plain = plain && !ds[i].g_mask && !ds[i].in_prelu && !ds[i].mask_slopes;
  }
  bool small = true;  // the Winograd kernel addresses a tensor through a buffer resource: 32-bit byte offsets
  for (int i = 0; i < n; ++i) {
    const int64_t pin = (int64_t)ds[i].B * (ds[i].ups ? (ds[i].H >> 1) * (ds[i].W >> 1) : ds[i].H * ds[i].W);
    small = small && pin * ds[i].in_cs * 4 < (int64_t(1) << 31) && (int64_t)ds[i].B * ds[i].H * ds[i].W * ds[i].g_cs * 4 < (int64_t(1) << 31);
  }
  if (fast && plain && !s2d && small && neosr_conv::wino_mode() == 2 && wgrad4_enabled()) {  // conv3x3_wgrad_wino4_kernel
    WgradMultiArgs w = a;
    const int P = a.pair_start[MAXD];
    if (prof) neosr_prof_algo(2);
    w4_geometry(w);
    w.timeline = g_wgrad_timeline;
    w.bpart = workspace + (int64_t)P * w.nsplit * WG_TILE;
    w.xcd_full = P <= W4_SLOTS / 8 ? (W4_SLOTS / 8) / P : 0;
    if (w.xcd_full > w.nsplit / 8) w.xcd_full = w.nsplit / 8;
    w.xcd_q = w.xcd_full * P + ceil_div((w.nsplit - 8 * w.xcd_full) * P, 8);
    const dim3 wgrid = w.xcd ? dim3(8 * w.xcd_q) : dim3(P, w.nsplit);
    hipLaunchKernelGGL(conv3x3_wgrad_wino4_kernel, wgrid, dim3(256), 0, st, w);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    if (prof) neosr_prof_begin(NEOSR_PROF_WGRAD_REDUCE, stream, 0.0, 0.0);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(WG_TILE / 64, P), dim3(256), 0, st, w);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    return 0;
  }
  if (fast && plain && !s2d && small && neosr_conv::wino_enabled()) {  // Winograd form (see conv3x3_wgrad_wino_kernel)
    WgradMultiArgs w = a;
    const int P = a.pair_start[MAXD];
    if (prof) neosr_prof_algo(1);
    w.bpart = workspace + (int64_t)P * a.w_nsplit * WW_PART;
    w.xcd_full = P <= WW_SLOTS / 8 ? (WW_SLOTS / 8) / P : 0;  // 96 slots per XCD
    if (w.xcd_full > a.w_nsplit / 8) w.xcd_full = a.w_nsplit / 8;
    w.xcd_q = w.xcd_full * P + ceil_div((a.w_nsplit - 8 * w.xcd_full) * P, 8);
    const dim3 wgrid = w.xcd ? dim3(8 * w.xcd_q) : dim3(P, a.w_nsplit);
    hipLaunchKernelGGL(conv3x3_wgrad_wino_kernel, wgrid, dim3(256), 0, st, w);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    if (prof) neosr_prof_begin(NEOSR_PROF_WGRAD_REDUCE, stream, 0.0, 0.0);
    hipLaunchKernelGGL(conv3x3_wgrad_wino_reduce_kernel, dim3(1024 / 64, P), dim3(256), 0, st, w);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid = a.xcd ? dim3(8 * a.xcd_q) : dim3(a.pair_start[MAXD], a.nsplit);
  if (fast && s2d)
    hipLaunchKernelGGL((conv3x3_wgrad_multi_kernel<true, true>), grid, dim3(256), 0, st, a);
  else if (fast && plain)
    hipLaunchKernelGGL((conv3x3_wgrad_multi_kernel<true, false, true>), grid, dim3(256), 0, st, a);
  else if (fast)
    hipLaunchKernelGGL(conv3x3_wgrad_multi_kernel<true>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv3x3_wgrad_multi_kernel<false>, grid, dim3(256), 0, st, a);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  if (prof) neosr_prof_begin(NEOSR_PROF_WGRAD_REDUCE, stream, 0.0, 0.0);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(WG_TILE / 64, a.pair_start[MAXD]),
                     dim3(256), 0, st, a);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t neosr_conv3x3_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t K,
                                                       int32_t N) {
  neosr_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.H = H; d.W = W; d.K = K; d.N = N;
  d.in = d.g = (const float*)16; d.dw = (float*)16;  // geometry query only
  return neosr_conv3x3_wgrad_multi_workspace_bytes(&d, 1);
}

extern "C" int neosr_conv3x3_wgrad(const neosr_wgrad_desc* dp, void* stream) {
  NEOSR_CHECK(dp && dp->workspace, "wgrad: null descriptor/workspace");
  return neosr_conv3x3_wgrad_multi(dp, 1, dp->workspace, stream);
}
