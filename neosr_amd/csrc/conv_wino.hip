// conv_wino.hip — Winograd F(2x2, 3x3) form of the 3x3 / stride 1 / pad 1 convolution for gfx950 (fp32, MFMA).
//
// Used for the RDB trunk's forward and gather-form backward-data launches (neosr/archs/esrgan_arch.py:82-142 and
// their autograd backward): the same epilogue contract as conv3x3_glds_kernel (bias, LeakyReLU/ReLU, two residual
// scale-adds, accumulate, derivative mask of the output slice), 16/36 of its multiplications.
//
//   Y(2x2) = A^T [ sum_k U_k (.) V_k ] A,   U = G g G^T (4x4 per (cin, cout), precomputed by neosr_conv3x3_pack_wino),
//   V = B^T d B (4x4 per (tile, cin)),  d = the 4x4 input patch of the tile (Lavin & Gray, correlation form)
//
// MI355X mapping.  A workgroup (4 waves) owns 8 x 16 output pixels = 32 Winograd tiles and 32 output channels.  Each of
// the 16 transform positions (i, j) is an independent 32 tiles x 32 cout x K GEMM; WAVE i OWNS ROW i of the 4x4
// transform (positions (i, 0..3)): 4 accumulators of v_mfma_f32_32x32x2_f32.
//   * lane = (tile m = lane & 31, k half lh = lane >> 5) is exactly the A-fragment owner, so the lane loads the two
//     input rows of ITS tile that row i of B^T needs (i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), applies the
//     row pass in registers and feeds the four results straight into MFMAs as A operands: the transformed input V
//     never exists in memory or LDS;
//   * the B operand of position (i, j), channel pair, cout = lane & 31 is one float of the packed U image: each wave
//     reads ITS four positions of the next 16-channel chunk straight from global memory (coalesced 1 KB rows, L2
//     resident: every workgroup of the launch reads the same image) one chunk ahead — U never touches LDS either;
//   * LDS holds only the raw 10 x 18 input tile of a 32-channel chunk (2 x 11.25 KB, global_load_lds_dwordx4, two
//     buffers, ONE barrier per chunk = per 64 MFMAs of a wave), shared by the four waves; at the end each wave applies
//     the column pass of A^T M A to its own four accumulators, the 8 x 32 x 32 result crosses LDS once, and thread
//     (tile, cout quad) finishes the row pass, runs the epilogue and stores 2 x 2 pixels x 4 channels with 16-byte
//     stores.
// Exact fp32 products and sums, but not the direct form's summation order: results differ from it by ~1e-6 relative
// (tests: <= 1e-4 against the oracle at kernel level; north_star allows 1e-3).  neosr_set_winograd(0) /
// NEOSR_AMD_WINOGRAD=0 selects the direct kernel for the same launches.
#include <cstring>
#include <vector>
#include <stdlib.h>
#include "conv_common.h"
#include "conv_pack.h"

using namespace neosr_conv;

namespace {

constexpr int WT_H = 8, WT_W = 16;            // output pixels per workgroup
constexpr int WR_H = WT_H + 2, WR_W = WT_W + 2;  // raw input tile 10 x 18
constexpr int WR_PIX = WR_H * WR_W;           // 180
constexpr int WR_GRAN = WR_PIX * 4;           // 720 16-byte granules per 16-channel half chunk
constexpr int WR_HALF = 3 * 256 * 4;          // floats per half chunk (3 workgroup-wide DMA rounds = 12 KB)
constexpr int WR_BUF = 2 * WR_HALF;           // one raw buffer = a 32-channel chunk (24 KB); two buffers
constexpr int WCK = 32;                       // channels per DMA chunk / barrier (two 16-channel halves)
constexpr int WU_HALF = neosr_pack::WINO_IMG_FLOATS;  // 16 pos x 4 quads x 32 n x 4 = 8192 floats (32 KB) per 16 channels
constexpr int WM_S = 36;                      // tile stride of the accumulator exchange (conflict-free b128 writes)
static_assert(4 * 32 * WM_S <= WR_BUF, "each half of the exchange image must fit one raw buffer");

__device__ __forceinline__ void glds16w(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Raw tile image in LDS (per 16-channel half): granule index = pix' * 4 + slot,
//   pix' = py * 18 + (px & 1) * 9 + (px >> 1)   (even columns first: the tiles of a row read pixels 2 apart, which
//                                                 become consecutive 64-byte pixels),
//   slot = channel quad ^ ((py >> 1) & 3)
// -> the 16 lanes of every ds_read_b128 phase (8 consecutive tiles of a row x 2 tile rows) hit 16 distinct 16-byte bank
// groups for all patch positions (checked exhaustively, tools/wino_lds_layout.py).
__device__ __forceinline__ int raw_pix(int py, int px) { return py * WR_W + (px & 1) * (WR_W / 2) + (px >> 1); }

__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  // the two raw buffers are DISTINCT LDS objects (hipcc waits vmcnt(0) before a ds_read that may alias a pending LDS-DMA
  // write); the exchange image at the end reuses buffer 0
  __shared__ __attribute__((aligned(1024))) float lds[WR_BUF];
  __shared__ __attribute__((aligned(1024))) float ldsB[WR_BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, lh = lane >> 5;
  const int tyi = m >> 3, txi = m & 7;
  TL_MARK(0);

  int bid = xcd_tile(blockIdx.x, gridDim.x, args.xcd);
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * WT_W, y0 = ty * WT_H;
  const int n0 = blockIdx.y * 32;
  const int H = d.H, W = d.W, K = d.K;
  // `ups`: the input is the nearest x2 upsampling of a (H/2, W/2) map, gathered by the DMA addresses (never materialised)
  const int Hin = d.ups ? (H >> 1) : H, Win = d.ups ? (W >> 1) : W;
  const float* __restrict__ inb = d.in + (int64_t)b * Hin * Win * d.in_cs;
  const int nhalf = (K + 15) >> 4;          // 16-channel halves (= chunks of the U image)
  const int nchunks = (nhalf + 1) >> 1;     // 32-channel DMA chunks

  // DMA granule of this thread in round i: G = i*256 + tid -> pix' = G >> 2, slot = G & 3 (see raw_pix)
  int in_off[3], in_q4[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int G = i * 256 + tid;
    const int pp = G >> 2, slot = G & 3;
    in_off[i] = -1;
    in_q4[i] = 0;
    if (G < WR_GRAN) {
      const int py = pp / WR_W, rem = pp - py * WR_W;
      const int par = rem >= WR_W / 2 ? 1 : 0;
      const int px = 2 * (rem - par * (WR_W / 2)) + par;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      in_q4[i] = (slot ^ ((py >> 1) & 3)) << 2;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int sy = d.ups ? (gy >> 1) : gy, sx = d.ups ? (gx >> 1) : gx;
        in_off[i] = (sy * Win + sx) * d.in_cs + in_q4[i];
      }
    }
  }
  // this wave's row i = wave of B^T: t = sa * d[ra] + sb * d[rb]
  const int ra = wave == 0 ? 0 : 1, rb = wave == 3 ? 3 : 2;
  const float sa = wave == 2 ? -1.f : 1.f, sb = (wave == 0 || wave == 3) ? -1.f : 1.f;
  // LDS float offsets (within a half) of the 8 patch pixels this lane reads (2 rows x 4 columns) for k group 0;
  // k group 1 is the quad 2 higher: slot ^ 2 -> offset ^ 8
  int po[8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int py = 2 * tyi + (r ? rb : ra), px = 2 * txi + s;
      po[r * 4 + s] = raw_pix(py, px) * 16 + ((lh ^ ((py >> 1) & 3)) << 2);
    }

  // U image of this n-block: [half][pos 16][quad 4][n 32][4]; this lane's float4 of position (wave, j), quad q
  const float* __restrict__ wp = d.w_wino + (int64_t)blockIdx.y * nhalf * WU_HALF + (wave * 4) * 512 + m * 4;

  auto issue = [&](int c, int buf) {  // 32 channels: two halves of 3 DMA rounds each
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* rbuf = (buf ? ldsB : lds) + h * WR_HALF;
      const int c0 = c * WCK + h * 16;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float* src = (in_off[i] >= 0 && c0 + in_q4[i] < K) ? inb + in_off[i] + c0 : g_zero_page;
        glds16w(src, rbuf + (i * 4 + wave) * 256);
      }
    }
  };
  auto load_u = [&](int h, f32x4 (&u)[4][2]) {
    const float* p = wp + (int64_t)h * WU_HALF;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) u[j][g] = ld4f(p + j * 512 + (2 * g + lh) * 128);
  };

  // D = U^T V^T: rows = output channels (A operand = U), columns = tiles (B operand = V) -> a lane's accumulator
  // registers 4r..4r+3 are 4 consecutive channels of ITS tile (16-byte exchange writes)
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // row i of B^T as ONE fused multiply-add per component: t = d[ra] + sg * d[rb], sg = sa * sb; the overall sign sa
  // (-1 for wave 2 only) is applied to the wave's accumulators once, after the loop
  const float sg = sa * sb;
  // patch loads (8 x ds_read_b128 per k group) are issued one k group AHEAD of the MFMAs that consume them, so their
  // LDS latency hides under the previous group's 16 MFMAs
  auto loadp = [&](const float* rbuf, int g, f32x4 (&dd)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) dd[s] = ld4f(rbuf + (po[s] ^ (g << 3)));
  };
  // The fp32 MFMA shares the SIMD's FMA lanes with ordinary vector instructions (they do not overlap: measured, the
  // kernel's time is the SUM of both), so the transform is written on 2-wide vectors: v_pk_fma_f32 / v_pk_add_f32 do
  // two elements per lane in the 4 cycles of one scalar instruction — 16 packed instead of 32 scalar per 16 MFMAs.
  const f32x2 sg2 = {sg, sg};
  auto mac = [&](const f32x4 (&dd)[8], const f32x4 (&u)[4][2], int g) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x2 t[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x2 a = h ? __builtin_shufflevector(dd[s], dd[s], 2, 3) : __builtin_shufflevector(dd[s], dd[s], 0, 1);
        const f32x2 q = h ? __builtin_shufflevector(dd[4 + s], dd[4 + s], 2, 3) : __builtin_shufflevector(dd[4 + s], dd[4 + s], 0, 1);
        t[s] = __builtin_elementwise_fma(sg2, q, a);
      }
      const f32x2 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int e = 2 * h + e2;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[0][g][e], v0[e2], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[1][g][e], v1[e2], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[2][g][e], v2[e2], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[3][g][e], v3[e2], acc[3], 0, 0, 0);
      }
    }
  };

  f32x4 ua[4][2], ub[4][2], pa[8], pb[8];
  issue(0, 0);
  load_u(0, ua);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  TL_MARK(1);
  for (int c = 0; c < nchunks; ++c) {
    const float* rb0 = (c & 1) ? ldsB : lds;
    const float* rb1 = rb0 + WR_HALF;
    const bool h1 = 2 * c + 1 < nhalf, h2 = 2 * c + 2 < nhalf;  // workgroup-uniform
    // vmcnt retires in order: U of the second half is requested BEFORE the DMA of the next chunk, so that waiting for
    // it later leaves the DMA (and the U of the half after) in flight
    if (h1) load_u(2 * c + 1, ub);
    if (c + 1 < nchunks) issue(c + 1, (c + 1) & 1);
    // (sched_barrier: keep the issue order load(next) -> mac(current); the waitcnt pass then emits partial lgkmcnt counts)
    loadp(rb0, 0, pa);
    loadp(rb0, 1, pb);
    __builtin_amdgcn_sched_barrier(0);
    mac(pa, ua, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (h1) loadp(rb1, 0, pa);
    __builtin_amdgcn_sched_barrier(0);
    mac(pb, ua, 1);
    __builtin_amdgcn_sched_barrier(0);
    TL_MARK(2 + c * 3);
    // (hipcc waits vmcnt(0) wherever a register filled by a global load is first used once LDS-DMA loads are in
    // flight: the next half's U is therefore requested only AFTER the wait for this half's U, 16 MFMAs before its use)
    if (h1) {
      loadp(rb1, 1, pb);
      __builtin_amdgcn_sched_barrier(0);
      mac(pa, ub, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (h2) load_u(2 * c + 2, ua);
      __builtin_amdgcn_sched_barrier(0);
      mac(pb, ub, 1);
    } else if (h2) {
      load_u(2 * c + 2, ua);
    }
    TL_MARK(3 + c * 3);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // chunk c + 1 has landed ...
    __syncthreads();                      // ... for every wave, and buffer c & 1 is free again
    TL_MARK(4 + c * 3);
  }
  if (sa < 0.f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = -acc[j][r];
  }

  // ---- epilogue operands of thread (tile, cout quad), requested NOW: their global-load latency hides under the column
  // pass, the accumulator exchange and its barrier (the accumulators' inputs are dead, registers are plentiful here)
  const int et = tid >> 3, cq = (tid & 7) << 2;
  const int chq = n0 + cq;
  const bool ch_ok = chq < d.N;
  const int cs0 = ch_ok ? chq : 0;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (d.bias) bias = ld4f(d.bias + cs0);
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const int ey = y0 + 2 * (et >> 3), ex = x0 + 2 * (et & 7);
  bool okp[4];
  int64_t pixp[4];
  f32x4 r1[4], r2[4], r0[4], mk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int py = ey + (q >> 1), px = ex + (q & 1);
    okp[q] = ch_ok && py < H && px < W;
    pixp[q] = okp[q] ? ((int64_t)b * H + py) * W + px : 0;
    r1[q] = r2[q] = r0[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mk[q] = (f32x4){1.f, 1.f, 1.f, 1.f};
    if (d.res1) r1[q] = ld4f((okp[q] && chq < d.res1_nch) ? d.res1 + pixp[q] * d.res1_cs + chq : g_zero_page);
    if (d.res2) r2[q] = ld4f((okp[q] && chq < d.res2_nch) ? d.res2 + pixp[q] * d.res2_cs + chq : g_zero_page);
    if (d.accumulate) r0[q] = ld4f(okp[q] ? d.out + pixp[q] * d.out_cs + chq : g_zero_page);
    if (d.out_mask) mk[q] = ld4f(okp[q] ? d.out_mask + pixp[q] * d.out_mask_cs + chq : g_zero_page);
  }

  // ---- output transform, column pass IN THE WAVE: (M A)[i][b] = M[i][0] + M[i][1] + M[i][2] (b = 0), M[i][1] - M[i][2] - M[i][3]
  // (b = 1) on the wave's own four accumulators -> only 2 x 16 registers per lane cross the LDS
  // exchange image: X[i][b][tile][cout], tile stride WM_S; register 4rr + e <-> channel 8rr + 4lh + e
#pragma unroll
  for (int bq = 0; bq < 2; ++bq)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * rr + e;
        v[e] = bq == 0 ? (acc[0][r] + acc[1][r]) + acc[2][r] : (acc[1][r] - acc[2][r]) - acc[3][r];
      }
      // rows 0, 1 of the exchange image live in raw buffer 0, rows 2, 3 in raw buffer 1
      *reinterpret_cast<f32x4*>((wave < 2 ? lds : ldsB) + (((wave & 1) * 2 + bq) * 32 + m) * WM_S + 8 * rr + 4 * lh) = v;
    }
  __syncthreads();
  TL_MARK(61);

  // ---- row pass + epilogue: Y[a][b] = X[0][b] + X[1][b] + X[2][b] (a = 0), X[1][b] - X[2][b] - X[3][b]
  f32x4 y[2][2];
#pragma unroll
  for (int bq = 0; bq < 2; ++bq) {
    const f32x4 x0 = ld4f(lds + ((0 * 2 + bq) * 32 + et) * WM_S + cq);
    const f32x4 x1 = ld4f(lds + ((1 * 2 + bq) * 32 + et) * WM_S + cq);
    const f32x4 x2 = ld4f(ldsB + ((0 * 2 + bq) * 32 + et) * WM_S + cq);
    const f32x4 x3 = ld4f(ldsB + ((1 * 2 + bq) * 32 + et) * WM_S + cq);
    y[0][bq] = (x0 + x1) + x2;
    y[1][bq] = (x1 - x2) - x3;
  }
  TL_MARK(62);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = y[q >> 1][q & 1][e] + bias[e];
      t = t > 0.f ? t : t * s_uni;
      t = t * d.alpha + r1[q][e];
      t = t * d.alpha2 + r2[q][e];
      t += r0[q][e];
      o[e] = mk[q][e] > 0.f ? t : t * d.out_mask_slope;
    }
    *reinterpret_cast<f32x4*>(okp[q] ? d.out + pixp[q] * d.out_cs + chq : g_trash + tid * 4) = o;
  }
  TL_MARK(63);
}

// U = G g G^T of every (cin, cout) pair of an image: one thread per (n-block, chunk, k quad, n) loads the 4 x 9 taps of
// its four channels once and writes the 16 positions' 16-byte granules (each a coalesced 512-byte row across n):
//   dst[nblk][chunk][pos = i*4 + j][quad][n 32][4]
__global__ __launch_bounds__(256) void conv_pack_wino_kernel(const neosr_pack::Batch batch) {
  const neosr_pack::Image& im = batch.im[blockIdx.y];
  const int nch = (im.K + 15) >> 4, nblk = (im.N + 31) >> 5;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nblk * nch * 128) return;
  const int n32 = t & 31, q = (t >> 5) & 3;
  const int rest = t >> 7;
  const int chunk = rest % nch, nb = rest / nch;
  const int n = nb * 32 + n32, k0 = chunk * 16 + q * 4;
  float g[4][9];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) g[e][tap] = 0.f;
  if (n < im.N && k0 < im.K) {
    for (int s = 0; s < im.nseg; ++s) {
      const neosr_pack::Seg& sg = im.seg[s];
      if (k0 < sg.k_lo || k0 >= sg.k_lo + sg.k_cnt) continue;
      const int kk = k0 - sg.k_lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kk + e >= sg.k_cnt) break;
        const float* src = im.mode == NEOSR_CONV_FWD ? sg.w + ((int64_t)(sg.n_lo + n) * sg.w_cin + kk + e) * 9
                                                     : sg.w + ((int64_t)(kk + e) * sg.w_cin + sg.n_lo + n) * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) g[e][tap] = src[im.mode == NEOSR_CONV_FWD ? tap : 8 - tap];
      }
    }
  }
  // rows of G: (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1):  U = G g G^T, column pass then row pass
  float u[16][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float c[3][4];  // (g G^T)[a][j]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float g0 = g[e][a * 3], g1 = g[e][a * 3 + 1], g2 = g[e][a * 3 + 2];
      c[a][0] = g0;
      c[a][1] = 0.5f * g0 + 0.5f * g1 + 0.5f * g2;
      c[a][2] = 0.5f * g0 - 0.5f * g1 + 0.5f * g2;
      c[a][3] = g2;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[0 * 4 + j][e] = c[0][j];
      u[1 * 4 + j][e] = 0.5f * c[0][j] + 0.5f * c[1][j] + 0.5f * c[2][j];
      u[2 * 4 + j][e] = 0.5f * c[0][j] - 0.5f * c[1][j] + 0.5f * c[2][j];
      u[3 * 4 + j][e] = c[2][j];
    }
  }
  float* dst = im.dst + ((int64_t)(nb * nch + chunk) * WU_HALF) + (q * 32 + n32) * 4;
#pragma unroll
  for (int pos = 0; pos < 16; ++pos)
    *reinterpret_cast<float4*>(dst + pos * 512) = make_float4(u[pos][0], u[pos][1], u[pos][2], u[pos][3]);
}

int g_wino = -1;  // -1: read NEOSR_AMD_WINOGRAD on first use (default 2: F(4x4,3x3) where an image is given)

}  // namespace

int neosr_conv::wino_mode() {
  if (g_wino < 0) {
    const char* e = getenv("NEOSR_AMD_WINOGRAD");
    const int v = e ? atoi(e) : 2;
    g_wino = v <= 0 ? 0 : (v == 1 ? 1 : 2);
  }
  return g_wino;
}

bool neosr_conv::wino_enabled() { return wino_mode() != 0; }

extern "C" int neosr_get_winograd(void) { return neosr_conv::wino_mode(); }

extern "C" int neosr_set_winograd(int on) {
  const int prev = neosr_conv::wino_mode();
  g_wino = on <= 0 ? 0 : (on == 1 ? 1 : 2);
  return prev;
}

void neosr_conv::launch_wino(const ConvArgs& a, hipStream_t st) {
  ConvArgs w = a;
  w.tiles_x = ceil_div(a.d.W, WT_W);
  w.tiles_y = ceil_div(a.d.H, WT_H);
  dim3 grid(w.tiles_x * w.tiles_y * a.d.B, ceil_div(a.d.N, 32));
  hipLaunchKernelGGL(conv3x3_wino_kernel, grid, dim3(256), 0, st, w);
}

int neosr_pack::launch_wino(const Image* images, int n, void* stream) {
  NEOSR_CHECK(images && n > 0, "conv pack (winograd): bad arguments");
  for (int i0 = 0; i0 < n; i0 += BATCH) {
    const int cnt = n - i0 < BATCH ? n - i0 : BATCH;
    Batch bt;
    memset(&bt, 0, sizeof(bt));
    int64_t gran = 0;
    for (int i = 0; i < cnt; ++i) {
      bt.im[i] = images[i0 + i];
      const int64_t g = wino_image_floats(bt.im[i].N, bt.im[i].K) / 64;  // one thread per 16 granules
      gran = g > gran ? g : gran;
    }
    dim3 grid((unsigned)((gran + 255) / 256), cnt);
    hipLaunchKernelGGL(conv_pack_wino_kernel, grid, dim3(256), 0, (hipStream_t)stream, bt);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t neosr_conv3x3_pack_wino_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return -1;
  return neosr_pack::wino_image_floats(N, K) * 4;
}

extern "C" int neosr_conv3x3_pack_wino(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode, float* dst,
                                       void* stream) {
  NEOSR_CHECK(w && dst && w_cout > 0 && w_cin > 0, "conv3x3_pack_wino: bad arguments");
  NEOSR_CHECK(mode == NEOSR_CONV_FWD || mode == NEOSR_CONV_DGRAD, "conv3x3_pack_wino: bad mode");
  NEOSR_CHECK((uintptr_t)dst % 16 == 0, "conv3x3_pack_wino: dst must be 16-byte aligned");
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
  return neosr_pack::launch_wino(&im, 1, stream);
}

extern "C" int neosr_conv3x3_pack_many(const neosr_pack_item* items, int32_t n, void* stream) {
  NEOSR_CHECK(items && n > 0, "conv3x3_pack_many: bad arguments");
  std::vector<neosr_pack::Image> direct, wino, wino4;
  for (int i = 0; i < n; ++i) {
    const neosr_pack_item& it = items[i];
    NEOSR_CHECK(it.w && it.dst && it.w_cout > 0 && it.w_cin > 0, "conv3x3_pack_many: bad item");
    NEOSR_CHECK(it.mode == NEOSR_CONV_FWD || it.mode == NEOSR_CONV_DGRAD, "conv3x3_pack_many: bad mode");
    NEOSR_CHECK((uintptr_t)it.dst % 16 == 0, "conv3x3_pack_many: dst must be 16-byte aligned");
    neosr_pack::Image im;
    memset(&im, 0, sizeof(im));
    im.dst = it.dst;
    im.mode = it.mode;
    im.N = it.mode == NEOSR_CONV_FWD ? it.w_cout : it.w_cin;
    im.K = it.mode == NEOSR_CONV_FWD ? it.w_cin : it.w_cout;
    im.nseg = 1;
    im.seg[0].w = it.w;
    im.seg[0].w_cin = it.w_cin;
    im.seg[0].k_lo = 0;
    im.seg[0].k_cnt = im.K;
    im.seg[0].n_lo = 0;
    (it.kind == 2 ? wino4 : it.kind ? wino : direct).push_back(im);
  }
  if (!direct.empty())
    if (int rc = neosr_pack::launch(direct.data(), (int)direct.size(), stream)) return rc;
  if (!wino.empty())
    if (int rc = neosr_pack::launch_wino(wino.data(), (int)wino.size(), stream)) return rc;
  if (!wino4.empty())
    if (int rc = neosr_pack::launch_wino4(wino4.data(), (int)wino4.size(), stream)) return rc;
  return 0;
}
