// color.hip — the pieces of consistency_loss (neosr/losses/consistency_loss.py:14-192)
// for gfx950: clamp, torchvision GaussianBlur(21, sigma 3) with reflect padding and its adjoint,
// sRGB -> CIE L* luma, sRGB -> Oklab chroma, and the two cosine-similarity terms.  Planar NCHW fp32,
// one pass per op, all HBM-bound; every backward is the exact adjoint / Jacobian of its forward.
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t n, int cap = 8192) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------------------------- clamp
__global__ __launch_bounds__(256) void clamp_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                    float* __restrict__ out, int64_t n, float lo, float hi) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = x[e];
    out[e] = g ? ((v >= lo && v <= hi) ? g[e] : 0.f) : fminf(fmaxf(v, lo), hi);
  }
}

// ---------------------------------------------------------------------------------- gaussian blur
struct Taps {
  float w[NEOSR_BLUR_MAX_TAPS];
  int n;
};

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// one axis of the reflect-padded correlation: out[p] = sum_k w_k in[reflect(p + k - r)]
// axis stride `st`, axis length `len`; `adjoint` gathers the transpose instead (pad gradients fold back)
__global__ __launch_bounds__(256) void blur_axis_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        const Taps t, int64_t n, int len, int64_t st, int adjoint) {
  const int r = t.n / 2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int p = (int)((e / st) % len);
    const float* base = in + (e - (int64_t)p * st);
    float s = 0.f;
    if (!adjoint) {
      for (int k = 0; k < t.n; ++k) s += t.w[k] * base[(int64_t)reflect(p + k - r, len) * st];
    } else {
      // out[q] = sum_k w_k * sum over source positions i in {q, -q, 2(len-1)-q} of in[i - k + r] (valid targets)
      for (int k = 0; k < t.n; ++k) {
        float a = 0.f;
        int pp = p - k + r;
        if (pp >= 0 && pp < len) a += base[(int64_t)pp * st];
        if (p >= 1) {  // padded index i = -p
          pp = -p - k + r;
          if (pp >= 0 && pp < len) a += base[(int64_t)pp * st];
        }
        if (p <= len - 2) {  // padded index i = 2(len-1) - p
          pp = 2 * (len - 1) - p - k + r;
          if (pp >= 0 && pp < len) a += base[(int64_t)pp * st];
        }
        s += t.w[k] * a;
      }
    }
    out[e] = s;
  }
}

// ---------------------------------------------------------------------------------- colour maps
__device__ __forceinline__ float lin_rgb(float v) { return v <= 0.04045f ? v / 12.92f : powf((v + 0.055f) / 1.055f, 2.4f); }
__device__ __forceinline__ float lin_rgb_d(float v) {
  return v <= 0.04045f ? 1.f / 12.92f : 2.4f / 1.055f * powf((v + 0.055f) / 1.055f, 1.4f);
}
__device__ __forceinline__ float scbrt(float v) { return copysignf(powf(fabsf(v), 1.f / 3.f), v); }
// d/dv sign(v)|v|^(1/3) = |v|^(-2/3) / 3
__device__ __forceinline__ float scbrt_d(float v) { return powf(fabsf(v), -2.f / 3.f) / 3.f; }

// CIE L* / 100, clamped to [0, 1], times `mul` (consistency_loss.py:106-133; the low branch is the
// reference's Y * (Y * 24389/27))
__global__ __launch_bounds__(256) void luma_kernel(const float* __restrict__ rgb, const float* __restrict__ g,
                                                   float* __restrict__ out, int B, int64_t plane, float mul) {
  const int64_t n = (int64_t)B * plane;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / plane, p = e - b * plane;
    const float* px = rgb + b * 3 * plane + p;
    const float r = px[0], gg = px[plane], bb = px[2 * plane];
    const float Y = 0.2126f * lin_rgb(r) + 0.7152f * lin_rgb(gg) + 0.0722f * lin_rgb(bb);
    const bool low = Y <= (216.f / 24389.f);
    const float L = low ? Y * (Y * (24389.f / 27.f)) : scbrt(Y) * 116.f - 16.f;
    const float v = L / 100.f;
    if (!g) {
      out[e] = fminf(fmaxf(v, 0.f), 1.f) * mul;
    } else {
      const float dL = low ? 2.f * Y * (24389.f / 27.f) : 116.f * scbrt_d(Y);
      const float gv = (v >= 0.f && v <= 1.f) ? g[e] * mul * dL / 100.f : 0.f;
      float* o = out + b * 3 * plane + p;
      o[0] = gv * 0.2126f * lin_rgb_d(r);
      o[plane] = gv * 0.7152f * lin_rgb_d(gg);
      o[2 * plane] = gv * 0.0722f * lin_rgb_d(bb);
    }
  }
}

// Oklab (a, b) * mul + 0.5, clamped to [0, 1] (consistency_loss.py:63-104,159-165): (B,3,H,W) -> (B,2,H,W)
__global__ __launch_bounds__(256) void chroma_kernel(const float* __restrict__ rgb, const float* __restrict__ g,
                                                     float* __restrict__ out, int B, int64_t plane, float mul) {
  const int64_t n = (int64_t)B * plane;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / plane, p = e - b * plane;
    const float* px = rgb + b * 3 * plane + p;
    const float r0 = px[0], g0 = px[plane], b0 = px[2 * plane];
    const float r = lin_rgb(r0), gg = lin_rgb(g0), bb = lin_rgb(b0);
    const float l = 0.4122214708f * r + 0.5363325363f * gg + 0.0514459929f * bb;
    const float m = 0.2119034982f * r + 0.6806995451f * gg + 0.1073969566f * bb;
    const float s = 0.0883024619f * r + 0.2817188376f * gg + 0.6299787005f * bb;
    const float l_ = scbrt(l), m_ = scbrt(m), s_ = scbrt(s);
    const float ca = (1.9779984951f * l_ - 2.4285922050f * m_ + 0.4505937099f * s_) * mul + 0.5f;
    const float cb = (0.0259040371f * l_ + 0.7827717662f * m_ - 0.8086757660f * s_) * mul + 0.5f;
    if (!g) {
      out[b * 2 * plane + p] = fminf(fmaxf(ca, 0.f), 1.f);
      out[b * 2 * plane + plane + p] = fminf(fmaxf(cb, 0.f), 1.f);
    } else {
      const float ga = (ca >= 0.f && ca <= 1.f) ? g[b * 2 * plane + p] * mul : 0.f;
      const float gb = (cb >= 0.f && cb <= 1.f) ? g[b * 2 * plane + plane + p] * mul : 0.f;
      const float dl_ = 1.9779984951f * ga + 0.0259040371f * gb;
      const float dm_ = -2.4285922050f * ga + 0.7827717662f * gb;
      const float ds_ = 0.4505937099f * ga - 0.8086757660f * gb;
      const float dl = dl_ * scbrt_d(l), dm = dm_ * scbrt_d(m), ds = ds_ * scbrt_d(s);
      float* o = out + b * 3 * plane + p;
      o[0] = (0.4122214708f * dl + 0.2119034982f * dm + 0.0883024619f * ds) * lin_rgb_d(r0);
      o[plane] = (0.5363325363f * dl + 0.6806995451f * dm + 0.2817188376f * ds) * lin_rgb_d(g0);
      o[2 * plane] = (0.0514459929f * dl + 0.1073969566f * dm + 0.6299787005f * ds) * lin_rgb_d(b0);
    }
  }
}

// ---------------------------------------------------------------------------------- cosine similarity
// nn.CosineSimilarity(dim=1, eps) of (G, L, inner) tensors: vectors of length L with stride `inner`.
// fwd: per-vector cos into `cosv` (G*inner values) + per-block partial sums; bwd: d(mean cos)/da.
__global__ __launch_bounds__(256) void cos_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ stats, float* __restrict__ partial, int64_t nvec,
                                                      int L, int64_t inner, float eps) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    const int64_t gidx = v / inner, in = v - gidx * inner;
    const float* pa = a + gidx * L * inner + in;
    const float* pb = b + gidx * L * inner + in;
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int k = 0; k < L; ++k) {
      const float x = pa[(int64_t)k * inner], y = pb[(int64_t)k * inner];
      dot += x * y;
      na += x * x;
      nb += y * y;
    }
    // ATen: dot / sqrt(clamp_min(na * nb, eps^2))
    const float den = sqrtf(fmaxf(na * nb, eps * eps));
    const float c = dot / den;
    stats[3 * v] = dot;
    stats[3 * v + 1] = na;
    stats[3 * v + 2] = nb;
    acc += c;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out = 1 - sum(partial) / nvec
__global__ __launch_bounds__(256) void cos_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int64_t nvec) {
  __shared__ double red[256];
  double a = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) a += (double)partial[k];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(1.0 - red[0] / (double)nvec);
}

// da = gout * d(1 - mean cos)/da = -gout / nvec * (b / den - dot * nb * a / den^3)   (den^2 = na nb above eps^2)
__global__ __launch_bounds__(256) void cos_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ stats, const float* __restrict__ gout,
                                                      float* __restrict__ da, int64_t nvec, int L, int64_t inner,
                                                      float eps) {
  const float gs = -gout[0] / (float)nvec;
  const int64_t n = nvec * L;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t in = e % inner, k = (e / inner) % L, gidx = e / (inner * L);
    const int64_t v = gidx * inner + in;
    (void)k;
    const float dot = stats[3 * v], na = stats[3 * v + 1], nb = stats[3 * v + 2];
    const float prod = na * nb;
    float d;
    if (prod > eps * eps) {
      const float den = sqrtf(prod);
      d = b[e] / den - dot * nb * a[e] / (den * prod);
    } else {
      d = b[e] / eps;
    }
    da[e] = gs * d;
  }
}

}  // namespace

extern "C" int neosr_clamp(const float* x, const float* g, float* out, int64_t n, float lo, float hi, void* stream) {
  NEOSR_CHECK(x && out && n > 0, "clamp: bad args");
  hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g, out, n, lo, hi);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_gaussian_blur_reflect(const float* in, float* out, float* tmp, const float* taps, int32_t ntaps,
                                           int32_t P, int32_t H, int32_t W, int32_t adjoint, void* stream) {
  NEOSR_CHECK(in && out && tmp && taps && ntaps > 0 && ntaps <= NEOSR_BLUR_MAX_TAPS && (ntaps & 1) && P > 0,
              "gaussian_blur_reflect: bad args (odd taps <= 31)");
  NEOSR_CHECK(H > ntaps / 2 && W > ntaps / 2, "gaussian_blur_reflect: image smaller than the reflect padding");
  Taps t;
  t.n = ntaps;
  for (int k = 0; k < ntaps; ++k) t.w[k] = taps[k];
  const int64_t n = (int64_t)P * H * W;
  hipStream_t st = (hipStream_t)stream;
  // horizontal then vertical (the 2-D window is the outer product; order is irrelevant for the adjoint too)
  hipLaunchKernelGGL(blur_axis_kernel, dim3(grid_for(n)), dim3(256), 0, st, in, tmp, t, n, W, (int64_t)1, adjoint);
  hipLaunchKernelGGL(blur_axis_kernel, dim3(grid_for(n)), dim3(256), 0, st, tmp, out, t, n, H, (int64_t)W, adjoint);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_rgb_to_luma(const float* rgb, const float* g, float* out, int32_t B, int32_t H, int32_t W,
                                 float mul, void* stream) {
  NEOSR_CHECK(rgb && out && B > 0 && H > 0 && W > 0, "rgb_to_luma: bad args");
  hipLaunchKernelGGL(luma_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, rgb, g, out, B,
                     (int64_t)H * W, mul);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_rgb_to_oklab_chroma(const float* rgb, const float* g, float* out, int32_t B, int32_t H, int32_t W,
                                         float mul, void* stream) {
  NEOSR_CHECK(rgb && out && B > 0 && H > 0 && W > 0, "rgb_to_oklab_chroma: bad args");
  hipLaunchKernelGGL(chroma_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, rgb, g, out,
                     B, (int64_t)H * W, mul);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_cosine_dist_fwd(const float* a, const float* b, float* stats, float* partial, float* out,
                                     int64_t groups, int32_t L, int64_t inner, float eps, void* stream) {
  NEOSR_CHECK(a && b && stats && partial && out && groups > 0 && L > 0 && inner > 0, "cosine_dist_fwd: bad args");
  const int64_t nvec = groups * inner;
  int nblk = grid_for(nvec, 1024);
  hipLaunchKernelGGL(cos_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, stats, partial, nvec, L, inner,
                     eps);
  hipLaunchKernelGGL(cos_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, out, nblk, nvec);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_cosine_dist_bwd(const float* a, const float* b, const float* stats, const float* gout, float* da,
                                     int64_t groups, int32_t L, int64_t inner, float eps, void* stream) {
  NEOSR_CHECK(a && b && stats && gout && da && groups > 0 && L > 0 && inner > 0, "cosine_dist_bwd: bad args");
  const int64_t nvec = groups * inner;
  hipLaunchKernelGGL(cos_bwd_kernel, dim3(grid_for(nvec * L)), dim3(256), 0, (hipStream_t)stream, a, b, stats, gout, da,
                     nvec, L, inner, eps);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
