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
//   * LDS holds only the raw 10 x 18 x 16-channel input tile of a chunk (11.25 KB, global_load_lds_dwordx4, two
//     buffers, ONE barrier per chunk), shared by the four waves, and at the end the 16 x 32 x 32 accumulator exchange
//     for the output transform A^T M A, after which thread (tile, cout quad) runs the epilogue and stores 2 x 2 pixels
//     x 4 channels with 16-byte stores.
// Exact fp32 products and sums, but not the direct form's summation order: results differ from it by ~1e-6 relative
// (tests: <= 1e-4 against the oracle at kernel level; north_star allows 1e-3).  neosr_set_winograd(0) /
// NEOSR_AMD_WINOGRAD=0 selects the direct kernel for the same launches.
#include <cstring>
#include <stdlib.h>
#include "conv_common.h"
#include "conv_pack.h"

using namespace neosr_conv;

namespace {

constexpr int WT_H = 8, WT_W = 16;            // output pixels per workgroup
constexpr int WR_H = WT_H + 2, WR_W = WT_W + 2;  // raw input tile 10 x 18
constexpr int WR_PIX = WR_H * WR_W;           // 180
constexpr int WR_GRAN = WR_PIX * 4;           // 720 16-byte granules per 16-channel chunk
constexpr int WR_BUF = 3 * 256 * 4;           // floats per raw buffer (3 workgroup-wide DMA rounds = 12 KB)
constexpr int WCK = 16;                       // channels per chunk
constexpr int WU_CHUNK = neosr_pack::WINO_IMG_FLOATS;  // 16 pos x 4 quads x 32 n x 4 = 8192 floats (32 KB)

__device__ __forceinline__ void glds16w(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ __attribute__((aligned(1024))) float lds[16 * 32 * 32];  // 64 KB: raw buffers in the loop, M at the end
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 31, lh = lane >> 5;
  const int tyi = m >> 3, txi = m & 7;

  int bid = xcd_tile(blockIdx.x, gridDim.x, args.xcd);
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * WT_W, y0 = ty * WT_H;
  const int n0 = blockIdx.y * 32;
  const int H = d.H, W = d.W, K = d.K;
  const float* __restrict__ inb = d.in + (int64_t)b * H * W * d.in_cs;
  const int nchunks = (K + WCK - 1) / WCK;

  // DMA granule of this thread in round i: G = i*256 + tid -> pixel G >> 2, slot G & 3; the slot holds channel quad
  // slot ^ ((pixel >> 1) & 3) (permutation applied on the global address: 4 lanes still fetch 64 contiguous bytes)
  int in_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int G = i * 256 + tid;
    const int pix = G >> 2, slot = G & 3;
    in_off[i] = -1;
    if (G < WR_GRAN) {
      const int py = pix / WR_W, px = pix - py * WR_W;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[i] = (gy * W + gx) * d.in_cs + ((slot ^ ((pix >> 1) & 3)) << 2);
    }
  }
  // this wave's row i = wave of B^T: t = sa * d[ra] + sb * d[rb]
  const int ra = wave == 0 ? 0 : 1, rb = wave == 3 ? 3 : 2;
  const float sa = wave == 2 ? -1.f : 1.f, sb = (wave == 0 || wave == 3) ? -1.f : 1.f;
  // LDS float offsets of the 8 patch pixels this lane reads (2 rows x 4 columns), before the per-group slot
  int pixo[8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) pixo[r * 4 + s] = (2 * tyi + (r ? rb : ra)) * WR_W + 2 * txi + s;

  // U image of this n-block: [chunk][pos 16][quad 4][n 32][4]; this lane's float4 of position (wave, j), quad q
  const float* __restrict__ wp = d.w_wino + (int64_t)blockIdx.y * nchunks * WU_CHUNK + (wave * 4) * 512 + m * 4;

  auto issue = [&](int c, int buf) {
    float* rbuf = lds + buf * WR_BUF;
    const int c0 = c * WCK;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int G = i * 256 + tid;
      const int q4 = ((G & 3) ^ (((G >> 2) >> 1) & 3)) << 2;
      const float* src = (in_off[i] >= 0 && c0 + q4 < K) ? inb + in_off[i] + c0 : g_zero_page;
      glds16w(src, rbuf + (i * 4 + wave) * 256);
    }
  };
  auto load_u = [&](int c, f32x4 (&u)[4][2]) {
    const float* p = wp + (int64_t)c * WU_CHUNK;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) u[j][g] = ld4f(p + j * 512 + (2 * g + lh) * 128);
  };

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  auto compute = [&](int buf, const f32x4 (&u)[4][2]) {
    const float* rbuf = lds + buf * WR_BUF;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 t[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int pa = pixo[s], pb = pixo[4 + s];
        const f32x4 da = ld4f(rbuf + pa * 16 + (((2 * g + lh) ^ ((pa >> 1) & 3)) << 2));
        const f32x4 db = ld4f(rbuf + pb * 16 + (((2 * g + lh) ^ ((pb >> 1) & 3)) << 2));
        t[s] = sa * da + sb * db;
      }
      const f32x4 v0 = t[0] - t[2], v1 = t[1] + t[2], v2 = t[2] - t[1], v3 = t[1] - t[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[e], u[0][g][e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[e], u[1][g][e], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v2[e], u[2][g][e], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v3[e], u[3][g][e], acc[3], 0, 0, 0);
      }
    }
  };

  f32x4 ua[4][2], ub[4][2];
  issue(0, 0);
  load_u(0, ua);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  int c = 0;
  for (; c + 2 < nchunks; c += 2) {
    issue(c + 1, 1);
    load_u(c + 1, ub);
    compute(0, ua);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    issue(c + 2, 0);
    load_u(c + 2, ua);
    compute(1, ub);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }
  if (c + 1 < nchunks) {  // two chunks left: c (buffer 0, ua) and c + 1
    issue(c + 1, 1);
    load_u(c + 1, ub);
    compute(0, ua);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    compute(1, ub);
  } else {
    compute(0, ua);
  }

  // ---- accumulator exchange: M[pos][tile][cout] (D rows = tiles: (r & 3) + 8 (r >> 2) + 4 lh, column = lane & 31)
  __syncthreads();  // every wave is done with the raw buffers
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = (r & 3) + 8 * (r >> 2) + 4 * lh;
      lds[((wave * 4 + j) * 32 + tile) * 32 + m] = acc[j][r];
    }
  __syncthreads();

  // ---- output transform + epilogue: thread = (tile, cout quad)
  const int et = tid >> 3, cq = (tid & 7) << 2;
  f32x4 s0[4], s1[4];  // rows of A^T M: s0[j] = M0j + M1j + M2j, s1[j] = M1j - M2j - M3j
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 m0 = ld4f(lds + ((0 * 4 + j) * 32 + et) * 32 + cq);
    const f32x4 m1 = ld4f(lds + ((1 * 4 + j) * 32 + et) * 32 + cq);
    const f32x4 m2 = ld4f(lds + ((2 * 4 + j) * 32 + et) * 32 + cq);
    const f32x4 m3 = ld4f(lds + ((3 * 4 + j) * 32 + et) * 32 + cq);
    s0[j] = m0 + m1 + m2;
    s1[j] = m1 - m2 - m3;
  }
  f32x4 y[2][2];
  y[0][0] = s0[0] + s0[1] + s0[2];
  y[0][1] = s0[1] - s0[2] - s0[3];
  y[1][0] = s1[0] + s1[1] + s1[2];
  y[1][1] = s1[1] - s1[2] - s1[3];

  const int chq = n0 + cq;
  const bool ch_ok = chq < d.N;
  const int cs0 = ch_ok ? chq : 0;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (d.bias) bias = ld4f(d.bias + cs0);
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const int ey = y0 + 2 * (et >> 3), ex = x0 + 2 * (et & 7);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int py = ey + a, px = ex + bb;
      const bool ok = ch_ok && py < H && px < W;
      const int64_t pix = ok ? ((int64_t)b * H + py) * W + px : 0;
      f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = r1, r0 = r1, mk = {1.f, 1.f, 1.f, 1.f};
      if (d.res1) r1 = ld4f((ok && chq < d.res1_nch) ? d.res1 + pix * d.res1_cs + chq : g_zero_page);
      if (d.res2) r2 = ld4f((ok && chq < d.res2_nch) ? d.res2 + pix * d.res2_cs + chq : g_zero_page);
      if (d.accumulate) r0 = ld4f(ok ? d.out + pix * d.out_cs + chq : g_zero_page);
      if (d.out_mask) mk = ld4f(ok ? d.out_mask + pix * d.out_mask_cs + chq : g_zero_page);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = y[a][bb][e] + bias[e];
        t = t > 0.f ? t : t * s_uni;
        t = t * d.alpha + r1[e];
        t = t * d.alpha2 + r2[e];
        t += r0[e];
        o[e] = mk[e] > 0.f ? t : t * d.out_mask_slope;
      }
      *reinterpret_cast<f32x4*>(ok ? d.out + pix * d.out_cs + chq : g_trash + tid * 4) = o;
    }
}

// U = G g G^T of every (cin, cout) pair of an image, one thread per 16-byte granule (pos, quad, n, 4 channels):
//   dst[nblk][chunk][pos = i*4 + j][quad][n 32][4]
__global__ __launch_bounds__(256) void conv_pack_wino_kernel(const neosr_pack::Batch batch) {
  const neosr_pack::Image& im = batch.im[blockIdx.y];
  const int nch = (im.K + 15) >> 4, nblk = (im.N + 31) >> 5;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nblk * nch * (WU_CHUNK / 4)) return;
  const int n32 = g & 31, q = (g >> 5) & 3, pos = (g >> 7) & 15;
  const int rest = g >> 11;
  const int chunk = rest % nch, nb = rest / nch;
  const int n = nb * 32 + n32, k0 = chunk * 16 + q * 4;
  const int i = pos >> 2, j = pos & 3;
  // rows of G: (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1)
  const float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < im.N && k0 < im.K) {
    for (int s = 0; s < im.nseg; ++s) {
      const neosr_pack::Seg& sg = im.seg[s];
      if (k0 < sg.k_lo || k0 >= sg.k_lo + sg.k_cnt) continue;
      const int kk = k0 - sg.k_lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kk + e >= sg.k_cnt) break;
        float u = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float row = 0.f;  // (g G^T)[a][j]
#pragma unroll
          for (int bq = 0; bq < 3; ++bq) {
            const int tap = a * 3 + bq;
            const float w = im.mode == NEOSR_CONV_FWD
                                ? sg.w[((int64_t)(sg.n_lo + n) * sg.w_cin + kk + e) * 9 + tap]
                                : sg.w[((int64_t)(kk + e) * sg.w_cin + sg.n_lo + n) * 9 + (8 - tap)];
            row += w * Gm[j][bq];
          }
          u += Gm[i][a] * row;
        }
        v[e] = u;
      }
    }
  }
  *reinterpret_cast<float4*>(im.dst + (int64_t)g * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

int g_wino = -1;  // -1: read NEOSR_AMD_WINOGRAD on first use (default on)

}  // namespace

bool neosr_conv::wino_enabled() {
  if (g_wino < 0) {
    const char* e = getenv("NEOSR_AMD_WINOGRAD");
    g_wino = (e && atoi(e) == 0) ? 0 : 1;
  }
  return g_wino != 0;
}

extern "C" int neosr_set_winograd(int on) {
  const int prev = neosr_conv::wino_enabled() ? 1 : 0;
  g_wino = on ? 1 : 0;
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
      const int64_t g = wino_image_floats(bt.im[i].N, bt.im[i].K) / 4;
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
