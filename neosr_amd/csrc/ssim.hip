// ssim.hip — multi-scale SSIM loss (neosr/losses/ssim_loss.py:11-163) for gfx950.
//
// Per scale the reference runs five depthwise 11x11 Gaussian convolutions (zero padding 5) of x, y, x^2,
// y^2 and x*y, forms the cs / ssim maps and averages them; between scales both images are 2x2 average
// pooled; loss = loss_weight * (1 - prod_i cs_i^w_i * ssim_4^w_4).  Here one kernel per scale stages a
// 42x42 halo tile of x and y in LDS, applies the separable window to the five products at once and emits
// (a) per-tile partial sums of cs and ssim (fixed-order two-stage reduction) and (b) the three derivative
// maps d map / d(G*x), d map / d(G*x^2), d map / d(G*xy) the backward pass needs.  Backward per scale
// filters those three maps with the same window (the zero-padded Gaussian is self-adjoint) and combines
// them as g * (F_mu + 2 x F_xx + y F_xy), adding the gradient that arrives through the average pool from
// the coarser scale.  Planar NCHW fp32, HBM-bound.
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

constexpr int TS = 32, R = 5, TW = TS + 2 * R;  // 42
constexpr int LX = TW + 1;                      // LDS row stride of the staged tiles
constexpr int LT = TS + 1;

struct Win11 {
  float w[11];
};

__device__ __forceinline__ void tile_of(int bid, int H, int W, int& p, int& y0, int& x0) {
  const int tx = (W + TS - 1) / TS, ty = (H + TS - 1) / TS;
  x0 = (bid % tx) * TS;
  const int t = bid / tx;
  y0 = (t % ty) * TS;
  p = t / ty;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                       const Win11 g, float* __restrict__ dmaps,
                                                       float* __restrict__ partial, int P, int H, int W, float C1,
                                                       float C2, int want_ssim) {
  __shared__ float xs[TW * LX], ys[TW * LX];
  __shared__ float tmp[5][TW * LT];
  __shared__ float red[2][256];
  int p, y0, x0;
  tile_of(blockIdx.x, H, W, p, y0, x0);
  const int tid = threadIdx.x;
  const float* xp = x + (int64_t)p * H * W;
  const float* yp = y + (int64_t)p * H * W;
  for (int e = tid; e < TW * TW; e += 256) {
    const int r = e / TW, c = e - r * TW;
    const int yy = y0 + r - R, xx = x0 + c - R;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    xs[r * LX + c] = in ? xp[(int64_t)yy * W + xx] : 0.f;
    ys[r * LX + c] = in ? yp[(int64_t)yy * W + xx] : 0.f;
  }
  __syncthreads();
  // horizontal pass over all 42 rows
  for (int e = tid; e < TW * TS; e += 256) {
    const int r = e / TS, c = e - r * TS;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float a = xs[r * LX + c + k], b = ys[r * LX + c + k], wk = g.w[k];
      sx += wk * a;
      sy += wk * b;
      sxx += wk * a * a;
      syy += wk * b * b;
      sxy += wk * a * b;
    }
    tmp[0][r * LT + c] = sx;
    tmp[1][r * LT + c] = sy;
    tmp[2][r * LT + c] = sxx;
    tmp[3][r * LT + c] = syy;
    tmp[4][r * LT + c] = sxy;
  }
  __syncthreads();
  float acc_cs = 0.f, acc_ssim = 0.f;
  const int64_t plane = (int64_t)H * W, mapsz = (int64_t)P * plane;
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e / TS, c = e - r * TS;
    const int yy = y0 + r, xx = x0 + c;
    if (yy >= H || xx >= W) continue;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float wk = g.w[k];
#pragma unroll
      for (int q = 0; q < 5; ++q) v[q] += wk * tmp[q][(r + k) * LT + c];
    }
    const float mux = v[0], muy = v[1];
    const float sx2 = v[2] - mux * mux, sy2 = v[3] - muy * muy, sxy = v[4] - mux * muy;
    const float A1 = 2.f * mux * muy + C1, A2 = 2.f * sxy + C2;
    const float B1 = mux * mux + muy * muy + C1, B2 = sx2 + sy2 + C2;
    const float l = A1 / B1, cs = A2 / B2;
    acc_cs += cs;
    acc_ssim += l * cs;
    if (dmaps) {
      // d cs / d(mu_x, E[x^2], E[xy])
      float d_mu = -2.f * muy / B2 + 2.f * mux * A2 / (B2 * B2);
      float d_xx = -A2 / (B2 * B2);
      float d_xy = 2.f / B2;
      if (want_ssim) {  // ssim = l * cs
        const float dl_mu = 2.f * muy / B1 - 2.f * mux * A1 / (B1 * B1);
        d_mu = cs * dl_mu + l * d_mu;
        d_xx *= l;
        d_xy *= l;
      }
      const int64_t o = p * plane + (int64_t)yy * W + xx;
      dmaps[o] = d_mu;
      dmaps[mapsz + o] = d_xx;
      dmaps[2 * mapsz + o] = d_xy;
    }
  }
  red[0][tid] = acc_cs;
  red[1][tid] = acc_ssim;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    partial[2 * (int64_t)blockIdx.x] = red[0][0];
    partial[2 * (int64_t)blockIdx.x + 1] = red[1][0];
  }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ dmaps, const float* __restrict__ x,
                                                       const float* __restrict__ y, const Win11 g,
                                                       const float* __restrict__ gscal, const float* __restrict__ gout,
                                                       const float* __restrict__ coarse, float* __restrict__ dx, int P,
                                                       int H, int W) {
  __shared__ float ds[3][TW * LX];
  __shared__ float tmp[3][TW * LT];
  int p, y0, x0;
  tile_of(blockIdx.x, H, W, p, y0, x0);
  const int tid = threadIdx.x;
  const int64_t plane = (int64_t)H * W, mapsz = (int64_t)P * plane;
  for (int e = tid; e < TW * TW; e += 256) {
    const int r = e / TW, c = e - r * TW;
    const int yy = y0 + r - R, xx = x0 + c - R;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const int64_t o = p * plane + (int64_t)yy * W + xx;
#pragma unroll
    for (int q = 0; q < 3; ++q) ds[q][r * LX + c] = in ? dmaps[q * mapsz + o] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < TW * TS; e += 256) {
    const int r = e / TS, c = e - r * TS;
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) s[q] += g.w[k] * ds[q][r * LX + c + k];
#pragma unroll
    for (int q = 0; q < 3; ++q) tmp[q][r * LT + c] = s[q];
  }
  __syncthreads();
  const float gs = gscal[0] * (gout ? gout[0] : 1.f);
  for (int e = tid; e < TS * TS; e += 256) {
    const int r = e / TS, c = e - r * TS;
    const int yy = y0 + r, xx = x0 + c;
    if (yy >= H || xx >= W) continue;
    float f[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) f[q] += g.w[k] * tmp[q][(r + k) * LT + c];
    const int64_t o = p * plane + (int64_t)yy * W + xx;
    float v = gs * (f[0] + 2.f * x[o] * f[1] + y[o] * f[2]);
    if (coarse) v += 0.25f * coarse[(int64_t)p * (H / 2) * (W / 2) + (int64_t)(yy >> 1) * (W / 2) + (xx >> 1)];
    dx[o] = v;
  }
}

__global__ __launch_bounds__(256) void avgpool2_planes_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int64_t n, int Ho, int Wo) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int xo = (int)(e % Wo);
    const int64_t t = e / Wo;
    const int yo = (int)(t % Ho);
    const int64_t p = t / Ho;
    const float* s = in + (p * 2 * Ho + 2 * yo) * (2 * Wo) + 2 * xo;
    out[e] = (s[0] + s[1] + s[2 * Wo] + s[2 * Wo + 1]) * 0.25f;
  }
}

// single workgroup: per-scale means from the tile partials (fixed order), the loss, and per scale
// d loss / d map-pixel = -loss_weight * w_i * prod / m_i / npix_i
__global__ __launch_bounds__(256) void msssim_finalize_kernel(const neosr_msssim_desc d) {
  __shared__ double red[256];
  __shared__ double mean[NEOSR_MSSSIM_SCALES];
  const int tid = threadIdx.x;
  for (int s = 0; s < d.nscales; ++s) {
    const int col = s == d.nscales - 1 ? 1 : 0;  // ssim on the last scale, cs before
    double a = 0.0;
    for (int k = tid; k < d.nblk[s]; k += 256) a += (double)d.partial[s][2 * (int64_t)k + col];
    red[tid] = a;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h) red[tid] += red[tid + h];
      __syncthreads();
    }
    if (tid == 0) mean[s] = red[0] / (double)d.npix[s];
    __syncthreads();
  }
  if (tid == 0) {
    float prod = 1.f;
    for (int s = 0; s < d.nscales; ++s) prod *= powf((float)mean[s], d.weights[s]);
    d.loss[0] = d.loss_weight * (1.f - prod);
    for (int s = 0; s < d.nscales; ++s)
      d.gscal[s] = -d.loss_weight * d.weights[s] * prod / (float)mean[s] / (float)d.npix[s];
  }
}

int tiles(int P, int H, int W) { return P * ceil_div(H, TS) * ceil_div(W, TS); }

}  // namespace

extern "C" int64_t neosr_ssim_tiles(int32_t P, int32_t H, int32_t W) { return tiles(P, H, W); }

extern "C" int neosr_ssim_fwd(const float* x, const float* y, const float* window11, float* dmaps, float* partial,
                              int32_t P, int32_t H, int32_t W, float C1, float C2, int32_t want_ssim, void* stream) {
  NEOSR_CHECK(x && y && window11 && partial && P > 0 && H > 0 && W > 0, "ssim_fwd: bad args");
  Win11 g;
  for (int k = 0; k < 11; ++k) g.w[k] = window11[k];  // host array (11 floats)
  hipLaunchKernelGGL(ssim_fwd_kernel, dim3(tiles(P, H, W)), dim3(256), 0, (hipStream_t)stream, x, y, g, dmaps, partial,
                     P, H, W, C1, C2, want_ssim);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_ssim_bwd(const float* dmaps, const float* x, const float* y, const float* window11,
                              const float* gscal, const float* gout, const float* coarse, float* dx, int32_t P,
                              int32_t H, int32_t W, void* stream) {
  NEOSR_CHECK(dmaps && x && y && window11 && gscal && dx && P > 0 && H > 0 && W > 0, "ssim_bwd: bad args");
  NEOSR_CHECK(!coarse || (H % 2 == 0 && W % 2 == 0), "ssim_bwd: coarse gradient needs even H, W");
  Win11 g;
  for (int k = 0; k < 11; ++k) g.w[k] = window11[k];
  hipLaunchKernelGGL(ssim_bwd_kernel, dim3(tiles(P, H, W)), dim3(256), 0, (hipStream_t)stream, dmaps, x, y, g, gscal,
                     gout, coarse, dx, P, H, W);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_avgpool2_planes(const float* in, float* out, int32_t P, int32_t H, int32_t W, void* stream) {
  NEOSR_CHECK(in && out && P > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "avgpool2_planes: even H, W only");
  const int64_t n = (int64_t)P * (H / 2) * (W / 2);
  int gsz = (int)((n + 255) / 256);
  if (gsz > 4096) gsz = 4096;
  hipLaunchKernelGGL(avgpool2_planes_kernel, dim3(gsz), dim3(256), 0, (hipStream_t)stream, in, out, n, H / 2, W / 2);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_msssim_finalize(const neosr_msssim_desc* d, void* stream) {
  NEOSR_CHECK(d && d->nscales > 0 && d->nscales <= NEOSR_MSSSIM_SCALES && d->loss && d->gscal, "msssim_finalize: bad args");
  for (int s = 0; s < d->nscales; ++s)
    NEOSR_CHECK(d->partial[s] && d->nblk[s] > 0 && d->npix[s] > 0, "msssim_finalize: scale %d incomplete", s);
  hipLaunchKernelGGL(msssim_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *d);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
