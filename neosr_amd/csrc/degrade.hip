// degrade.hip — the on-the-fly (Real-ESRGAN style, 2nd-order) degradation bank of neosr's `otf`
// model as HIP kernels for gfx950.  Images are planar NCHW fp32 in [0,1] like the reference's.
// Every kernel is HBM/LDS-bound elementwise or small-stencil work: coalesced loads along W, LDS
// tiles for the stencils, no MFMA.  Reference behaviour restated (paths under /root/reference):
//   filter2D            neosr/utils/diffjpeg.py:558-584    (reflect pad + per-sample correlation)
//   F.interpolate       neosr/models/otf.py:126,179-186,222-226,243-247 (area|bilinear|bicubic)
//   gaussian / poisson  neosr/data/degradations.py:569-605,738-786 (+ random_* wrappers)
//   DiffJPEG            neosr/utils/diffjpeg.py:65-555     (differentiable=False -> torch.round)
//   quantise / crop / pool  neosr/models/otf.py:251-260,37-90
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t work_items, int cap = 4096) {
  int64_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// --------------------------------------------------------------------------------- filter2D
// 32x32 output tile per workgroup, (32+k-1)^2 reflect-padded halo in LDS, per-sample kernel in LDS.
constexpr int F_T = 32;
constexpr int F_KMAX = 21;
constexpr int F_HALO = F_T + F_KMAX - 1;  // 52

__device__ __forceinline__ int reflect(int i, int n) {
  // F.pad(mode="reflect"): mirror without repeating the edge sample
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__global__ __launch_bounds__(256) void filter2d_kernel(const float* __restrict__ img,
                                                       const float* __restrict__ kern,
                                                       float* __restrict__ out, int B, int C, int H,
                                                       int W, int k, int kern_batched) {
  __shared__ float tile[F_HALO][F_HALO + 1];
  __shared__ float kw[F_KMAX * F_KMAX];
  const int plane = blockIdx.z;  // b*C + c
  const int b = plane / C;
  const int x0 = blockIdx.x * F_T, y0 = blockIdx.y * F_T;
  const int p = k >> 1, T = F_T + k - 1;
  const float* src = img + (int64_t)plane * H * W;
  const float* kp = kern + (kern_batched ? (int64_t)b * k * k : 0);
  for (int i = threadIdx.x; i < k * k; i += 256) kw[i] = kp[i];
  for (int i = threadIdx.x; i < T * T; i += 256) {
    const int ty = i / T, tx = i - ty * T;
    const int gy = reflect(y0 + ty - p, H), gx = reflect(x0 + tx - p, W);
    // tiles hanging over the image edge read clamped (unused) positions
    tile[ty][tx] = src[(int64_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;  // 8 row groups, 4 rows each
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < k; ++ky)
    for (int kx = 0; kx < k; ++kx) {
      const float w = kw[ky * k + kx];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += w * tile[ry + 8 * j + ky][cx + kx];
    }
  float* dst = out + (int64_t)plane * H * W;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int y = y0 + ry + 8 * j, x = x0 + cx;
    if (y < H && x < W) dst[(int64_t)y * W + x] = acc[j];
  }
}

// --------------------------------------------------------------------------------- resize
// ATen semantics, align_corners=False, antialias=False.  rs_h/rs_w = source-coordinate scale
// (1/scale_factor when the caller passed scale_factor, in/out when it passed size).
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ __launch_bounds__(256) void resize_kernel(const float* __restrict__ in,
                                                     float* __restrict__ out, int planes, int Hin,
                                                     int Win, int Hout, int Wout, int mode,
                                                     float rs_h, float rs_w) {
  const int64_t total = (int64_t)planes * Hout * Wout;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int ox = (int)(e % Wout);
    const int64_t t = e / Wout;
    const int oy = (int)(t % Hout);
    const int pl = (int)(t / Hout);
    const float* src = in + (int64_t)pl * Hin * Win;
    float v;
    if (mode == NEOSR_RESIZE_AREA) {
      // adaptive_avg_pool2d: [floor(i*in/out), ceil((i+1)*in/out))
      const int ys = (int)floorf((float)(oy * Hin) / Hout), ye = (int)ceilf((float)((oy + 1) * Hin) / Hout);
      const int xs = (int)floorf((float)(ox * Win) / Wout), xe = (int)ceilf((float)((ox + 1) * Win) / Wout);
      float s = 0.f;
      for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) s += src[(int64_t)y * Win + x];
      v = s / (float)((ye - ys) * (xe - xs));
    } else if (mode == NEOSR_RESIZE_BILINEAR) {
      float sy = rs_h * (oy + 0.5f) - 0.5f, sx = rs_w * (ox + 0.5f) - 0.5f;
      sy = sy < 0.f ? 0.f : sy;
      sx = sx < 0.f ? 0.f : sx;
      const int y0 = (int)sy, x0 = (int)sx;
      const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
      const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
      v = hy * (hx * src[(int64_t)y0 * Win + x0] + lx * src[(int64_t)y0 * Win + x1]) +
          ly * (hx * src[(int64_t)y1 * Win + x0] + lx * src[(int64_t)y1 * Win + x1]);
    } else {  // bicubic, A = -0.75, border indices clamped, no output clamp
      const float A = -0.75f;
      const float sy = rs_h * (oy + 0.5f) - 0.5f, sx = rs_w * (ox + 0.5f) - 0.5f;
      const int iy = (int)floorf(sy), ix = (int)floorf(sx);
      const float ty = sy - iy, tx = sx - ix;
      const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
      const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
      v = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = min(max(iy - 1 + j, 0), Hin - 1);
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) r += wx[i] * src[(int64_t)yy * Win + min(max(ix - 1 + i, 0), Win - 1)];
        v += wy[j] * r;
      }
    }
    out[e] = v;
  }
}

// --------------------------------------------------------------------------------- noise
// out = clamp(img + noise_c*(sigma/255)*(1-g) + noise_gray*(sigma/255)*g, 0, 1)
// noise_c (B,3,H,W) per-pixel N(0,1); noise_gray ONE (H,W) field shared by the batch
// (degradations.py:593-598) — may be NULL when no sample draws gray noise.
__global__ __launch_bounds__(256) void gaussian_noise_kernel(const float* __restrict__ img,
                                                             const float* __restrict__ noise,
                                                             const float* __restrict__ noise_gray,
                                                             const float* __restrict__ sigma,
                                                             const float* __restrict__ gray,
                                                             float* __restrict__ out, int B, int C,
                                                             int HW) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int b = (int)(e / ((int64_t)C * HW));
    const float sg = sigma[b];
    float n = noise[e] * sg / 255.f;
    if (noise_gray) {
      const float g = gray[b];
      n = n * (1.f - g) + (noise_gray[p] * sg / 255.f) * g;
    }
    out[e] = fminf(fmaxf(img[e] + n, 0.f), 1.f);
  }
}

__device__ __forceinline__ float quant255(float v) {  // clamp(round(v*255),0,255)/255, half-to-even
  return fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f) / 255.f;
}

// presence bitmap of the 8-bit levels of each sample: levels[b][8] (uint32), zeroed by the host.
// gray != 0: levels of rgb_to_grayscale (0.2989, 0.587, 0.114) instead of the 3 channels.
__global__ __launch_bounds__(256) void level_bitmap_kernel(const float* __restrict__ img,
                                                           unsigned* __restrict__ levels, int HW,
                                                           int gray) {
  __shared__ unsigned bits[8];
  const int b = blockIdx.y;
  if (threadIdx.x < 8) bits[threadIdx.x] = 0u;
  __syncthreads();
  const float* s = img + (int64_t)b * 3 * HW;
  const int n = gray ? HW : 3 * HW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float v = gray ? (0.2989f * s[i] + 0.587f * s[HW + i] + 0.114f * s[2 * HW + i]) : s[i];
    const int lv = (int)fminf(fmaxf(rintf(v * 255.f), 0.f), 255.f);
    atomicOr(&bits[lv >> 5], 1u << (lv & 31));
  }
  __syncthreads();
  if (threadIdx.x < 8 && bits[threadIdx.x]) atomicOr(&levels[b * 8 + threadIdx.x], bits[threadIdx.x]);
}

// vals[b] = 2^ceil(log2(#levels)); rate = quant255(img or gray(img)) * vals[b]
__global__ __launch_bounds__(256) void poisson_rate_kernel(const float* __restrict__ img,
                                                           const unsigned* __restrict__ levels,
                                                           float* __restrict__ vals,
                                                           float* __restrict__ rate, int B, int HW,
                                                           int gray) {
  const int C = gray ? 1 : 3;
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / ((int64_t)C * HW));
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) cnt += __popc(levels[b * 8 + w]);
    const float val = cnt <= 1 ? 1.f : (float)(1u << (32 - __clz(cnt - 1)));
    float v;
    if (gray) {
      const int p = (int)(e % HW);
      const float* s = img + (int64_t)b * 3 * HW;
      v = 0.2989f * s[p] + 0.587f * s[HW + p] + 0.114f * s[2 * HW + p];
    } else {
      v = img[e];
    }
    rate[e] = quant255(v) * val;
    if ((e % ((int64_t)C * HW)) == 0) vals[b] = val;
  }
}

// out = clamp(img + ((P/vals - q(img))*(1-g) + (Pg/vals_g - q(gray(img)))*g) * scale, 0, 1)
__global__ __launch_bounds__(256) void poisson_noise_kernel(
    const float* __restrict__ img, const float* __restrict__ P, const float* __restrict__ vals,
    const float* __restrict__ Pg, const float* __restrict__ vals_g, const float* __restrict__ scale,
    const float* __restrict__ gray, float* __restrict__ out, int B, int HW) {
  const int64_t total = (int64_t)B * 3 * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int b = (int)(e / ((int64_t)3 * HW));
    const float v = img[e];
    float n = P[e] / vals[b] - quant255(v);
    if (Pg) {
      const float* s = img + (int64_t)b * 3 * HW;
      const float gq = quant255(0.2989f * s[p] + 0.587f * s[HW + p] + 0.114f * s[2 * HW + p]);
      const float ng = Pg[(int64_t)b * HW + p] / vals_g[b] - gq;
      const float g = gray[b];
      n = n * (1.f - g) + ng * g;
    }
    out[e] = fminf(fmaxf(v + n * scale[b], 0.f), 1.f);
  }
}

// --------------------------------------------------------------------------------- DiffJPEG
#include "diffjpeg_table.h"
// quantisation tables AS STORED by the reference (diffjpeg.py:16-38: the standard tables transposed)
__device__ const float c_ytab[8][8] = {
    {16, 12, 14, 14, 18, 24, 49, 72},  {11, 12, 13, 17, 22, 35, 64, 92},
    {10, 14, 16, 22, 37, 55, 78, 95},  {16, 19, 24, 29, 56, 64, 87, 98},
    {24, 26, 40, 51, 68, 81, 103, 112}, {40, 58, 57, 87, 109, 104, 121, 100},
    {51, 60, 69, 80, 103, 113, 120, 103}, {61, 55, 56, 62, 77, 92, 101, 99}};
__device__ const float c_ctab[8][8] = {
    {17, 18, 24, 47, 99, 99, 99, 99}, {18, 21, 26, 66, 99, 99, 99, 99},
    {24, 26, 56, 99, 99, 99, 99, 99}, {47, 66, 99, 99, 99, 99, 99, 99},
    {99, 99, 99, 99, 99, 99, 99, 99}, {99, 99, 99, 99, 99, 99, 99, 99},
    {99, 99, 99, 99, 99, 99, 99, 99}, {99, 99, 99, 99, 99, 99, 99, 99}};

// One wavefront per 16x16 MCU (4 MCUs per workgroup).  Lane l = (u, v) = (l>>3, l&7) is at once
// the DCT coefficient index and the pixel position inside an 8x8 block.  Everything between the
// single read and the single write of the image lives in LDS/registers.
//
// ROUNDING CONTRACT (round 6): every fp32 operation below is the reference's operation in the reference's
// order, so the quantised coefficients — and the output — are the reference's bit for bit (no coefficient "rounds the other
// way").  `torch.tensordot` (diffjpeg.py:93, 183, 374, 473) is a matrix product whose reduction the CPU library runs as ONE
// chain of fused multiply-adds in index order starting from 0 (checked against the reference run: tests/golden/
// degrade_prims.npz is reproduced exactly): fmaf chains here, never a product-then-add, never a reassociation; the DCT
// table holds the reference's float32(float64 product) values (diffjpeg_table.h), the scale / alpha factors are
// float32(outer(alpha, alpha) [* 0.25]) and the two divisions are correctly rounded.
__global__ __launch_bounds__(256) void diffjpeg_kernel(const float* __restrict__ img,
                                                       const float* __restrict__ quality,
                                                       float* __restrict__ out, int B, int H, int W,
                                                       int mcu_x, int mcu_y) {
  __shared__ float sY[4][16][17], sCb[4][16][17], sCr[4][16][17];
  __shared__ float sBlk[4][6][8][9];  // dequantised, alpha-scaled coefficients / reconstructed blocks
  __shared__ float sT[64 * 65];       // c_dct, rows padded to 65: conflict-free by column (forward) and by row (inverse)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = lane >> 3, v = lane & 7;
  const int64_t mcu = (int64_t)blockIdx.x * 4 + wave;
  const int64_t nmcu = (int64_t)B * mcu_y * mcu_x;
  const bool live = mcu < nmcu;
  const int mx = live ? (int)(mcu % mcu_x) : 0;
  const int my = live ? (int)((mcu / mcu_x) % mcu_y) : 0;
  const int b = live ? (int)(mcu / ((int64_t)mcu_x * mcu_y)) : 0;
  const int64_t HW = (int64_t)H * W;
  const float* src = img + (int64_t)b * 3 * HW;
  for (int i = threadIdx.x; i < 4096; i += 256) sT[(i >> 6) * 65 + (i & 63)] = c_dct[i >> 6][i & 63];

  // quality -> factor (quality_to_factor, diffjpeg.py:48-61), per sample, on the device
  const float q = quality[b];
  const float factor = __fdiv_rn(q < 50.f ? __fdiv_rn(5000.f, q) : 200.f - q * 2.f, 100.f);

  // 1) load 4 pixels per lane, RGB*255 -> YCbCr (zero padding outside the image); tensordot over the 3 channels = an
  //    fmaf chain from 0 (its first link is the plain product), then + shift
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = lane + 64 * i, py = p >> 4, px = p & 15;
    const int gy = my * 16 + py, gx = mx * 16 + px;
    float r = 0.f, g = 0.f, bl = 0.f;
    if (live && gy < H && gx < W) {
      const int64_t o = (int64_t)gy * W + gx;
      r = src[o] * 255.f;
      g = src[HW + o] * 255.f;
      bl = src[2 * HW + o] * 255.f;
    }
    sY[wave][py][px] = __fmaf_rn(bl, 0.114f, __fmaf_rn(g, 0.587f, __fmul_rn(r, 0.299f))) + 0.f;
    sCb[wave][py][px] = __fmaf_rn(bl, 0.5f, __fmaf_rn(g, -0.331264f, __fmul_rn(r, -0.168736f))) + 128.f;
    sCr[wave][py][px] = __fmaf_rn(bl, -0.081312f, __fmaf_rn(g, -0.418688f, __fmul_rn(r, 0.5f))) + 128.f;
  }
  __syncthreads();

  const bool u0 = u == 0, v0 = v == 0;
  const float dct_scale = (u0 && v0) ? DCT_SCALE_00 : (u0 || v0) ? DCT_SCALE_0X : DCT_SCALE_XX;
  const float idct_alpha = (u0 && v0) ? IDCT_ALPHA_00 : (u0 || v0) ? IDCT_ALPHA_0X : IDCT_ALPHA_XX;
  // 2) forward DCT + quantise + dequantise for the 6 blocks; lane = coefficient (u, v)
#pragma unroll
  for (int blk = 0; blk < 6; ++blk) {
    float s = 0.f;
    if (blk < 4) {
      const int by = (blk >> 1) * 8, bx = (blk & 1) * 8;
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y)
          s = __fmaf_rn(__fsub_rn(sY[wave][by + x][bx + y], 128.f), sT[(x * 8 + y) * 65 + lane], s);
    } else {
      float(*pl)[17] = blk == 4 ? sCb[wave] : sCr[wave];
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          // 2x2 average (avg_pool2d kernel 2 stride 2: the window summed row by row, then / 4)
          const float c = __fmul_rn(__fadd_rn(__fadd_rn(__fadd_rn(pl[2 * x][2 * y], pl[2 * x][2 * y + 1]), pl[2 * x + 1][2 * y]),
                                              pl[2 * x + 1][2 * y + 1]), 0.25f);
          s = __fmaf_rn(__fsub_rn(c, 128.f), sT[(x * 8 + y) * 65 + lane], s);
        }
    }
    const float coef = __fmul_rn(dct_scale, s);
    const float tab = __fmul_rn(blk < 4 ? c_ytab[u][v] : c_ctab[u][v], factor);
    const float deq = __fmul_rn(rintf(__fdiv_rn(coef, tab)), tab);  // torch.round = half-to-even
    sBlk[wave][blk][u][v] = __fmul_rn(deq, idct_alpha);              // iDCT8x8: image *= alpha
  }
  __syncthreads();
  // 3) inverse DCT; lane = pixel (u, v) of the block: tensor[x, y, u, v] of iDCT8x8 = c_dct[8 u + v][8 x + y]
  float rec[6];
#pragma unroll
  for (int blk = 0; blk < 6; ++blk) {
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) s = __fmaf_rn(sBlk[wave][blk][x][y], sT[lane * 65 + x * 8 + y], s);
    rec[blk] = __fadd_rn(__fmul_rn(0.25f, s), 128.f);
  }
  __syncthreads();
#pragma unroll
  for (int blk = 0; blk < 6; ++blk) sBlk[wave][blk][u][v] = rec[blk];
  __syncthreads();
  // 4) chroma x2 repeat, (+ shift) YCbCr -> RGB as the fmaf chain of its tensordot, clamp, / 255, store
  if (!live) return;
  float* dst = out + (int64_t)b * 3 * HW;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = lane + 64 * i, py = p >> 4, px = p & 15;
    const int gy = my * 16 + py, gx = mx * 16 + px;
    if (gy < H && gx < W) {
      const float yv = __fadd_rn(sBlk[wave][(py >> 3) * 2 + (px >> 3)][py & 7][px & 7], 0.f);
      const float cb = __fadd_rn(sBlk[wave][4][py >> 1][px >> 1], -128.f);
      const float cr = __fadd_rn(sBlk[wave][5][py >> 1][px >> 1], -128.f);
      const float y1 = __fmul_rn(yv, 1.f);
      const float r = __fmaf_rn(cr, 1.402f, __fmaf_rn(cb, 0.f, y1));
      const float g = __fmaf_rn(cr, -0.714136f, __fmaf_rn(cb, -0.344136f, y1));
      const float bl = __fmaf_rn(cr, 0.f, __fmaf_rn(cb, 1.772f, y1));
      const int64_t o = (int64_t)gy * W + gx;
      dst[o] = __fdiv_rn(fminf(fmaxf(r, 0.f), 255.f), 255.f);
      dst[HW + o] = __fdiv_rn(fminf(fmaxf(g, 0.f), 255.f), 255.f);
      dst[2 * HW + o] = __fdiv_rn(fminf(fmaxf(bl, 0.f), 255.f), 255.f);
    }
  }
}

// --------------------------------------------------------------------------------- misc
__global__ __launch_bounds__(256) void quantize_u8_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = quant255(in[e]);
}

__global__ __launch_bounds__(256) void clamp01_kernel(const float* __restrict__ in,
                                                      float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = fminf(fmaxf(in[e], 0.f), 1.f);
}

// out (planes, h, w) = in (planes, H, W)[top:top+h, left:left+w]
__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ in,
                                                   float* __restrict__ out, int planes, int H, int W,
                                                   int top, int left, int h, int w) {
  const int64_t total = (int64_t)planes * h * w;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int x = (int)(e % w);
    const int64_t t = e / w;
    const int y = (int)(t % h);
    const int pl = (int)(t / h);
    out[e] = in[((int64_t)pl * H + top + y) * W + left + x];
  }
}

// dst[i] = src[idx[i]] over rows of `row` floats (pool shuffle: queue = queue[randperm])
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                          const int64_t* __restrict__ idx,
                                                          float* __restrict__ dst, int nrows,
                                                          int64_t row) {
  const int64_t total = (int64_t)nrows * row;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / row, c = e - r * row;
    dst[e] = src[idx[r] * row + c];
  }
}

}  // namespace

extern "C" int neosr_filter2d(const float* img, const float* kernel, float* out, int32_t B,
                              int32_t C, int32_t H, int32_t W, int32_t k, int32_t kernel_batched,
                              void* stream) {
  NEOSR_CHECK(img && kernel && out && B > 0 && C > 0 && H > 0 && W > 0, "filter2d: bad args");
  NEOSR_CHECK(k % 2 == 1 && k >= 1 && k <= F_KMAX, "filter2d: kernel size must be odd and <= 21 (got %d)", k);
  NEOSR_CHECK(k / 2 < H && k / 2 < W, "filter2d: reflect padding needs k//2 < H, W");
  NEOSR_CHECK((int64_t)B * C <= 65535, "filter2d: too many planes");
  dim3 grid(ceil_div(W, F_T), ceil_div(H, F_T), B * C);
  hipLaunchKernelGGL(filter2d_kernel, grid, dim3(256), 0, (hipStream_t)stream, img, kernel, out, B,
                     C, H, W, k, kernel_batched);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_resize(const float* in, float* out, int32_t planes, int32_t Hin, int32_t Win,
                            int32_t Hout, int32_t Wout, int32_t mode, float rs_h, float rs_w,
                            void* stream) {
  NEOSR_CHECK(in && out && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resize: bad args");
  NEOSR_CHECK(mode >= 0 && mode <= 2, "resize: mode must be area|bilinear|bicubic");
  hipLaunchKernelGGL(resize_kernel, dim3(grid_for((int64_t)planes * Hout * Wout)), dim3(256), 0,
                     (hipStream_t)stream, in, out, planes, Hin, Win, Hout, Wout, mode, rs_h, rs_w);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_gaussian_noise(const float* img, const float* noise, const float* noise_gray,
                                    const float* sigma, const float* gray, float* out, int32_t B,
                                    int32_t C, int32_t H, int32_t W, void* stream) {
  NEOSR_CHECK(img && noise && sigma && out && B > 0 && C > 0 && H > 0 && W > 0, "gaussian_noise: bad args");
  NEOSR_CHECK(!noise_gray || gray, "gaussian_noise: gray flags missing");
  hipLaunchKernelGGL(gaussian_noise_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0,
                     (hipStream_t)stream, img, noise, noise_gray, sigma, gray, out, B, C, H * W);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_poisson_rate(const float* img, uint32_t* levels_ws, float* vals, float* rate,
                                  int32_t B, int32_t H, int32_t W, int32_t gray, void* stream) {
  NEOSR_CHECK(img && levels_ws && vals && rate && B > 0 && H > 0 && W > 0, "poisson_rate: bad args");
  hipStream_t st = (hipStream_t)stream;
  NEOSR_HIP(hipMemsetAsync(levels_ws, 0, (size_t)B * 8 * sizeof(uint32_t), st));
  const int HW = H * W;
  int gx = ceil_div((gray ? 1 : 3) * HW, 256 * 16);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(level_bitmap_kernel, dim3(gx, B), dim3(256), 0, st, img, levels_ws, HW, gray);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(poisson_rate_kernel, dim3(grid_for((int64_t)B * (gray ? 1 : 3) * HW)), dim3(256),
                     0, st, img, levels_ws, vals, rate, B, HW, gray);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_poisson_noise(const float* img, const float* P, const float* vals,
                                   const float* P_gray, const float* vals_gray, const float* scale,
                                   const float* gray, float* out, int32_t B, int32_t H, int32_t W,
                                   void* stream) {
  NEOSR_CHECK(img && P && vals && scale && out && B > 0 && H > 0 && W > 0, "poisson_noise: bad args");
  NEOSR_CHECK(!P_gray || (vals_gray && gray), "poisson_noise: gray inputs incomplete");
  hipLaunchKernelGGL(poisson_noise_kernel, dim3(grid_for((int64_t)B * 3 * H * W)), dim3(256), 0,
                     (hipStream_t)stream, img, P, vals, P_gray, vals_gray, scale, gray, out, B, H * W);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_diffjpeg(const float* img, const float* quality, float* out, int32_t B,
                              int32_t H, int32_t W, void* stream) {
  NEOSR_CHECK(img && quality && out && B > 0 && H > 0 && W > 0, "diffjpeg: bad args");
  const int mx = ceil_div(W, 16), my = ceil_div(H, 16);
  const int64_t nmcu = (int64_t)B * mx * my;
  hipLaunchKernelGGL(diffjpeg_kernel, dim3((unsigned)((nmcu + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, img, quality, out, B, H, W, mx, my);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_quantize_u8(const float* in, float* out, int64_t n, void* stream) {
  NEOSR_CHECK(in && out && n > 0, "quantize_u8: bad args");
  hipLaunchKernelGGL(quantize_u8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_clamp01(const float* in, float* out, int64_t n, void* stream) {
  NEOSR_CHECK(in && out && n > 0, "clamp01: bad args");
  hipLaunchKernelGGL(clamp01_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_crop(const float* in, float* out, int32_t planes, int32_t H, int32_t W,
                          int32_t top, int32_t left, int32_t h, int32_t w, void* stream) {
  NEOSR_CHECK(in && out && planes > 0 && top >= 0 && left >= 0 && h > 0 && w > 0 && top + h <= H &&
                  left + w <= W,
              "crop: window (%d,%d,%d,%d) outside %dx%d", top, left, h, w, H, W);
  hipLaunchKernelGGL(crop_kernel, dim3(grid_for((int64_t)planes * h * w)), dim3(256), 0,
                     (hipStream_t)stream, in, out, planes, H, W, top, left, h, w);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_gather_rows(const float* src, const int64_t* idx, float* dst, int32_t nrows,
                                 int64_t row_elems, void* stream) {
  NEOSR_CHECK(src && idx && dst && nrows > 0 && row_elems > 0, "gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)nrows * row_elems)), dim3(256), 0,
                     (hipStream_t)stream, src, idx, dst, nrows, row_elems);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ Poisson sampling (replaces torch.poisson)
// Counter-based Philox4x32-10 (Salmon et al. 2011): element e owns the counter stream (e, draw#, offset) under
// key = seed, so the field is a pure function of (seed, offset, rate) — no generator state, no host sync.
// Small rates: Knuth's product of uniforms; rate >= 10: Hoermann's PTRS transformed rejection (the
// algorithm behind numpy / ATen's CPU sampler), exact in distribution.
namespace {

struct Philox {
  uint32_t k0, k1, c0, c1, c2, c3;
  uint32_t out[4];
  int have = 0;
  uint32_t draw = 0;
  __device__ Philox(uint64_t seed, uint64_t offset, uint64_t elem)
      : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), c0((uint32_t)elem), c1((uint32_t)(elem >> 32)),
        c2((uint32_t)offset), c3((uint32_t)(offset >> 32)) {}
  __device__ void refill() {
    uint32_t a = c0, b = c1 ^ draw, c = c2, d = c3;  // the draw index perturbs the counter
    uint32_t x0 = k0, x1 = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const uint32_t hi0 = __umulhi(0xD2511F53u, a), lo0 = 0xD2511F53u * a;
      const uint32_t hi1 = __umulhi(0xCD9E8D57u, c), lo1 = 0xCD9E8D57u * c;
      a = hi1 ^ b ^ x0;
      b = lo1;
      c = hi0 ^ d ^ x1;
      d = lo0;
      x0 += 0x9E3779B9u;
      x1 += 0xBB67AE85u;
    }
    out[0] = a; out[1] = b; out[2] = c; out[3] = d;
    have = 4;
    ++draw;
  }
  __device__ float uniform() {  // (0, 1)
    if (!have) refill();
    const uint32_t x = out[--have];
    return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f);
  }
};

__global__ __launch_bounds__(256) void poisson_sample_kernel(const float* __restrict__ rate,
                                                             float* __restrict__ out, int64_t n,
                                                             uint64_t seed, uint64_t offset) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float lam = rate[e];
    Philox rng(seed, offset, (uint64_t)e);
    float k = 0.f;
    if (!(lam > 0.f)) {
      k = 0.f;
    } else if (lam < 10.f) {
      const float L = expf(-lam);
      float p = rng.uniform();
      while (p > L) {
        k += 1.f;
        p *= rng.uniform();
      }
    } else {
      const float slam = sqrtf(lam), loglam = logf(lam);
      const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
      const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
      for (;;) {
        const float U = rng.uniform() - 0.5f, V = rng.uniform();
        const float us = 0.5f - fabsf(U);
        k = floorf((2.f * a / us + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) break;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (logf(V) + logf(invalpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.f)) break;
      }
    }
    out[e] = k;
  }
}


// Standard normal field (replaces torch.randn in the live draw stream of the degradation bank, degradations.py:593-598):
// element pair (2i, 2i+1) = Box-Muller of two Philox uniforms of counter stream i -> a pure function of (seed, offset).
__global__ __launch_bounds__(256) void normal_sample_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                            uint64_t offset) {
  const int64_t npair = (n + 1) >> 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < npair; e += (int64_t)gridDim.x * 256) {
    Philox rng(seed, offset, (uint64_t)e);
    const float u1 = rng.uniform(), u2 = rng.uniform();
    const float r = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    out[2 * e] = r * cs;
    if (2 * e + 1 < n) out[2 * e + 1] = r * sn;
  }
}

// Blur / sinc kernel synthesis on the device (neosr/data/degradations.py:24-512 evaluated for a whole batch; the
// random PARAMETERS are drawn on the host in the reference's order, neosr_amd/data/degradations.py).  One workgroup per
// 21 x 21 kernel, float64 like the reference's numpy code, output float32 zero-padded around the k x k support.
//   params[i] = {type, k, sig_x, sig_y, theta, beta (or cutoff), isotropic, unused}
//   type 0 Gaussian exp(-q/2) | 1 generalized exp(-q^beta / 2) | 2 plateau 1 / (1 + q^beta) | 3 circular low-pass
//   cutoff * J1(cutoff r) / (2 pi r) (centre cutoff^2 / 4 pi) | 4 pulse;  q = g^T Sigma^-1 g on the centred grid
__global__ __launch_bounds__(512) void blur_kernels_kernel(const double* __restrict__ params, float* __restrict__ out) {
  __shared__ double red[8];
  const double* P = params + (int64_t)blockIdx.x * 8;
  const int type = (int)P[0], k = (int)P[1];
  const double sx = P[2], sy = P[3], th = P[4], beta = P[5];
  const bool iso = P[6] != 0.0;
  const int tid = threadIdx.x;
  const int py = tid / 21, px = tid - py * 21;
  const int pad = (21 - k) / 2;
  const int iy = py - pad, ix = px - pad;
  const bool in = tid < 441 && iy >= 0 && iy < k && ix >= 0 && ix < k;
  double v = 0.0;
  if (type == 4) {
    v = (tid == 10 * 21 + 10) ? 1.0 : 0.0;
  } else if (in) {
    const double c = (k - 1) * 0.5;
    const double gx = ix - c, gy = iy - c;
    if (type == 3) {
      const double r = sqrt(gx * gx + gy * gy);
      v = r == 0.0 ? beta * beta / (4.0 * 3.141592653589793) : beta * j1(beta * r) / (2.0 * 3.141592653589793 * r);
    } else {
      double a11, a12, a22;  // Sigma^-1
      if (iso) {
        a11 = a22 = 1.0 / (sx * sx);
        a12 = 0.0;
      } else {
        const double cs = cos(th), sn = sin(th), ix2 = 1.0 / (sx * sx), iy2 = 1.0 / (sy * sy);
        a11 = cs * cs * ix2 + sn * sn * iy2;
        a12 = cs * sn * (ix2 - iy2);
        a22 = sn * sn * ix2 + cs * cs * iy2;
      }
      const double q = a11 * gx * gx + 2.0 * a12 * gx * gy + a22 * gy * gy;
      v = type == 0 ? exp(-0.5 * q) : (type == 1 ? exp(-0.5 * pow(q, beta)) : 1.0 / (pow(q, beta) + 1.0));
    }
  }
  double s = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  double tot = 0.0;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[w];
  if (tid < 441) out[(int64_t)blockIdx.x * 441 + tid] = (float)(type == 4 ? v : v / tot);
}

}  // namespace

extern "C" int neosr_normal_sample(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  NEOSR_CHECK(out && n > 0, "normal_sample: bad args");
  hipLaunchKernelGGL(normal_sample_kernel, dim3(grid_for((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, out, n, seed,
                     offset);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_blur_kernels(const double* params, int32_t n, float* out, void* stream) {
  NEOSR_CHECK(params && out && n > 0, "blur_kernels: bad args");
  hipLaunchKernelGGL(blur_kernels_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, params, out);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_poisson_sample(const float* rate, float* out, int64_t n, uint64_t seed, uint64_t offset,
                                    void* stream) {
  NEOSR_CHECK(rate && out && n > 0, "poisson_sample: bad args");
  hipLaunchKernelGGL(poisson_sample_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, rate, out, n,
                     seed, offset);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
