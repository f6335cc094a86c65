// elementwise.hip — HBM-bound layout, index, loss and optimizer kernels for gfx950.
// All reductions are two-stage with a fixed summation order (no float atomics) so that results
// are run-to-run deterministic.  Reference call sites: see include/neosr_amd.h.
#include <cstring>

#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

constexpr int RED_BLOCKS = 1024;  // stage-1 partial count for flat reductions (<= 4096 ws floats)

__device__ __forceinline__ float block_reduce_sum_256(float v, float* sm /*4 floats*/) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------- layout
// NCHW -> NHWC through an LDS transpose tile: 64 pixels x C (C small) per block iteration
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int B, int C,
                                                           int HW, int out_cs) {
  // generic, coalesced on the read side (pixels contiguous per plane); C is small (3..64) for the
  // call sites on the path, so writes are short runs of C floats.
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int64_t t = e / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    out[((int64_t)b * HW + p) * out_cs + c] = in[e];
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int B, int C,
                                                           int HW, int in_cs) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int64_t t = e / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    out[e] = in[((int64_t)b * HW + p) * in_cs + c];
  }
}

// 2x2 sum pool, channels-last, float4 along channels
__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const float* __restrict__ in,
                                                          float* __restrict__ out, int B, int H,
                                                          int W, int C4, int in_cs, int out_cs,
                                                          int accumulate, const float* __restrict__ mask = nullptr,
                                                          int mask_cs = 0, float mask_slope = 1.f) {
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    int64_t t = e / C4;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int64_t W2 = 2 * W;
    const float* p = in + (((int64_t)b * 2 * H + 2 * y) * W2 + 2 * x) * in_cs + c4 * 4;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 bq = *reinterpret_cast<const float4*>(p + in_cs);
    const float4 c = *reinterpret_cast<const float4*>(p + W2 * in_cs);
    const float4 dq = *reinterpret_cast<const float4*>(p + W2 * in_cs + in_cs);
    float4 r;
    r.x = (a.x + bq.x) + (c.x + dq.x);
    r.y = (a.y + bq.y) + (c.y + dq.y);
    r.z = (a.z + bq.z) + (c.z + dq.z);
    r.w = (a.w + bq.w) + (c.w + dq.w);
    if (mask) {  // derivative of the LeakyReLU whose output is `mask` (the pooled gradient's own activation)
      const float4 m = *reinterpret_cast<const float4*>(mask + (((int64_t)b * H + y) * W + x) * mask_cs + c4 * 4);
      r.x = m.x > 0.f ? r.x : r.x * mask_slope;
      r.y = m.y > 0.f ? r.y : r.y * mask_slope;
      r.z = m.z > 0.f ? r.z : r.z * mask_slope;
      r.w = m.w > 0.f ? r.w : r.w * mask_slope;
    }
    float4* o = reinterpret_cast<float4*>(out + (((int64_t)b * H + y) * W + x) * out_cs + c4 * 4);
    if (accumulate) {
      const float4 old = *o;
      r.x += old.x;
      r.y += old.y;
      r.z += old.z;
      r.w += old.w;
    }
    *o = r;
  }
}

// PixelShuffle: one thread per output element (coalesced NCHW writes)
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ base,
                                                            float* __restrict__ out, int B, int C,
                                                            int H, int W, int r, int in_cs) {
  const int Ho = H * r, Wo = W * r;
  const int64_t total = (int64_t)B * C * Ho * Wo;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int xo = (int)(e % Wo);
    int64_t t = e / Wo;
    const int yo = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int h = yo / r, i = yo - h * r, w = xo / r, j = xo - w * r;
    float v = in[(((int64_t)b * H + h) * W + w) * in_cs + c * r * r + i * r + j];
    if (base) v += base[(((int64_t)b * C + c) * H + h) * W + w];
    out[e] = v;
  }
}

__global__ __launch_bounds__(256) void pixel_unshuffle_kernel(const float* __restrict__ gout,
                                                              float* __restrict__ gin, int B, int C,
                                                              int H, int W, int r, int gin_cs) {
  const int Ho = H * r, Wo = W * r;
  const int64_t total = (int64_t)B * C * Ho * Wo;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int xo = (int)(e % Wo);
    int64_t t = e / Wo;
    const int yo = (int)(t % Ho);
    t /= Ho;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int h = yo / r, i = yo - h * r, w = xo / r, j = xo - w * r;
    gin[(((int64_t)b * H + h) * W + w) * gin_cs + c * r * r + i * r + j] = gout[e];
  }
}

// PReLU slope grad, stage 1: block handles a slab of pixels, lanes along channels
__global__ __launch_bounds__(256) void prelu_dslope_stage1(const float* __restrict__ dA,
                                                           const float* __restrict__ z,
                                                           float* __restrict__ part, int64_t npix,
                                                           int C, int da_cs, int z_cs,
                                                           int pix_per_block) {
  // thread t -> channel (t % CPAD), pixel lane (t / CPAD)
  __shared__ float sm[256];
  const int cpad = C <= 64 ? 64 : (C <= 128 ? 128 : 256);
  const int c = threadIdx.x % cpad, pl = threadIdx.x / cpad, npl = 256 / cpad;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  float s = 0.f;
  if (c < C)
    for (int64_t p = p0 + pl; p < p1; p += npl) {
      const float zz = z[p * z_cs + c];
      s += dA[p * da_cs + c] * fminf(zz, 0.f);
    }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    float tot = 0.f;
    for (int k = 0; k < npl; ++k) tot += sm[k * cpad + c];
    part[(int64_t)blockIdx.x * C + c] = tot;
  }
}

// stage 2: 64 channels per workgroup, four block lanes walk the partial rows k = lane, lane + 4, ...
// (fixed combination order)
__global__ __launch_bounds__(256) void prelu_dslope_stage2(const float* __restrict__ part,
                                                           float* __restrict__ dslope, int nblk,
                                                           int C, int accumulate) {
  __shared__ float red[4][64];
  const int o = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int k = kl;
    for (; k + 12 < nblk; k += 16) {
      s0 += part[(int64_t)k * C + c];
      s1 += part[(int64_t)(k + 4) * C + c];
      s2 += part[(int64_t)(k + 8) * C + c];
      s3 += part[(int64_t)(k + 12) * C + c];
    }
    for (; k < nblk; k += 4) s0 += part[(int64_t)k * C + c];
  }
  red[kl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < C) {
    const float s = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    dslope[c] = accumulate ? dslope[c] + s : s;
  }
}

// ---------------------------------------------------------------- L1 loss
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b, int64_t n,
                                                         float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = a4[i], y = b4[i];
    s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += fabsf(a[i] - b[i]);
  const float r = block_reduce_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void finalize_sum_kernel(const float* __restrict__ part, int np,
                                                           float scale, float* __restrict__ out,
                                                           int do_sqrt) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  const float r = block_reduce_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = do_sqrt ? sqrtf(r * scale) : r * scale;
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ b,
                                                     const float* __restrict__ gout, int64_t n,
                                                     float scale, float* __restrict__ ga) {
  const float g = gout[0] * scale;
  const int64_t n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* g4 = reinterpret_cast<float4*>(ga);
  auto sgn = [g](float d) { return d > 0.f ? g : (d < 0.f ? -g : 0.f); };
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = a4[i], y = b4[i];
    float4 r;
    r.x = sgn(x.x - y.x);
    r.y = sgn(x.y - y.y);
    r.z = sgn(x.z - y.z);
    r.w = sgn(x.w - y.w);
    g4[i] = r;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) ga[i] = sgn(a[i] - b[i]);
}

// ---------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n,
                                                            float gscale, float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 x = g4[i];
    x.x *= gscale;
    x.y *= gscale;
    x.z *= gscale;
    x.w *= gscale;
    s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      const float x = g[i] * gscale;
      s += x * x;
    }
  const float r = block_reduce_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

struct AdamArgs {
  neosr_adamw_desc d;
  float bc1, bc2_sqrt;  // 1 - beta1^t,  sqrt(1 - beta2^t)
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float* ema,
                                          const AdamArgs& a, float clip) {
  const neosr_adamw_desc& d = a.d;
  g *= clip;
  // torch.optim.AdamW (single-tensor path): p *= 1 - lr*wd; m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g^2
  p *= 1.f - d.lr * d.weight_decay;
  m = m + (g - m) * (1.f - d.beta1);
  v = v * d.beta2 + (1.f - d.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + d.eps;
  p = p - (d.lr / a.bc1) * (m / denom);
  if (ema) {
    if (d.ema_decay < 0.f)
      *ema = p;
    else
      *ema = *ema + (p - *ema) * (1.f - d.ema_decay);  // _foreach_lerp_(ema, p, 1-decay)
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(const AdamArgs a) {
  const neosr_adamw_desc& d = a.d;
  float clip = d.grad_scale;
  if (d.max_norm > 0.f) {
    const float total = d.norm_ws[0];
    const float coef = d.max_norm / (total + 1e-6f);
    clip *= fminf(coef, 1.f);
  }
  const int64_t n4 = d.n >> 2;
  float4* p4 = reinterpret_cast<float4*>(d.param);
  const float4* g4 = reinterpret_cast<const float4*>(d.grad);
  float4* m4 = reinterpret_cast<float4*>(d.exp_avg);
  float4* v4 = reinterpret_cast<float4*>(d.exp_avg_sq);
  float4* e4 = reinterpret_cast<float4*>(d.ema);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 p = p4[i], m = m4[i], v = v4[i];
    const float4 g = g4[i];
    float4 e = d.ema ? e4[i] : make_float4(0, 0, 0, 0);
    adamw_one(p.x, g.x, m.x, v.x, d.ema ? &e.x : nullptr, a, clip);
    adamw_one(p.y, g.y, m.y, v.y, d.ema ? &e.y : nullptr, a, clip);
    adamw_one(p.z, g.z, m.z, v.z, d.ema ? &e.z : nullptr, a, clip);
    adamw_one(p.w, g.w, m.w, v.w, d.ema ? &e.w : nullptr, a, clip);
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
    if (d.ema) e4[i] = e;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < d.n; i += 256)
      adamw_one(d.param[i], d.grad[i], d.exp_avg[i], d.exp_avg_sq[i], d.ema ? d.ema + i : nullptr,
                a, clip);
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    p[i] = v;
}

// out[p, c] += alpha * in[p, c] over a channels-last slice
__global__ __launch_bounds__(256) void axpy_slice_kernel(float* __restrict__ out,
                                                         const float* __restrict__ in, int64_t npix,
                                                         int C, int out_cs, int in_cs, float alpha) {
  const int64_t total = npix * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const int64_t p = e / C;
    out[p * out_cs + c] += alpha * in[p * in_cs + c];
  }
}

inline int grid_for(int64_t work_items, int cap = 2048) {
  int64_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

}  // namespace

extern "C" int neosr_fill(float* p, int64_t n, float v, void* stream) {
  NEOSR_CHECK(p && n > 0, "fill: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, v);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_axpy_slice(float* out, const float* in, int64_t npix, int32_t C,
                                int32_t out_cs, int32_t in_cs, float alpha, void* stream) {
  NEOSR_CHECK(out && in && npix > 0 && C > 0, "axpy_slice: bad args");
  hipLaunchKernelGGL(axpy_slice_kernel, dim3(grid_for(npix * C)), dim3(256), 0, (hipStream_t)stream,
                     out, in, npix, C, out_cs, in_cs, alpha);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_nchw_to_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H,
                                  int32_t W, int32_t out_cs, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && C > 0 && H > 0 && W > 0 && out_cs >= C, "nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, C, H * W, out_cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_nhwc_to_nchw(const float* in, float* out, int32_t B, int32_t C, int32_t H,
                                  int32_t W, int32_t in_cs, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && C > 0 && H > 0 && W > 0 && in_cs >= C, "nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, C, H * W, in_cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_pool2x2_sum(const float* in, float* out, int32_t B, int32_t H, int32_t W,
                                 int32_t C, int32_t in_cs, int32_t out_cs, int32_t accumulate,
                                 void* stream) {
  NEOSR_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0, "pool2x2: bad args");
  NEOSR_CHECK(C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && (uintptr_t)in % 16 == 0 &&
                  (uintptr_t)out % 16 == 0,
              "pool2x2: needs 4-channel aligned tensors");
  hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, H, W, C / 4, in_cs, out_cs, accumulate);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_pool2x2_sum_masked(const float* in, float* out, const float* mask, int32_t B, int32_t H,
                                        int32_t W, int32_t C, int32_t in_cs, int32_t out_cs, int32_t mask_cs,
                                        float mask_slope, void* stream) {
  NEOSR_CHECK(in && out && mask && B > 0 && H > 0 && W > 0 && C > 0, "pool2x2_masked: bad args");
  NEOSR_CHECK(C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && mask_cs % 4 == 0 && (uintptr_t)in % 16 == 0 &&
                  (uintptr_t)out % 16 == 0 && (uintptr_t)mask % 16 == 0,
              "pool2x2_masked: needs 4-channel aligned tensors");
  hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, in, out, B, H, W, C / 4, in_cs, out_cs, 0, mask, mask_cs, mask_slope);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_pixel_shuffle_nhwc_to_nchw(const float* in, const float* base, float* out,
                                                int32_t B, int32_t C, int32_t H, int32_t W,
                                                int32_t r, int32_t in_cs, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && C > 0 && H > 0 && W > 0 && r > 0 && in_cs >= C * r * r,
              "pixel_shuffle: bad args");
  hipLaunchKernelGGL(pixel_shuffle_kernel, dim3(grid_for((int64_t)B * C * H * W * r * r)),
                     dim3(256), 0, (hipStream_t)stream, in, base, out, B, C, H, W, r, in_cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_pixel_unshuffle_nchw_to_nhwc(const float* gout, float* gin, int32_t B,
                                                  int32_t C, int32_t H, int32_t W, int32_t r,
                                                  int32_t gin_cs, void* stream) {
  NEOSR_CHECK(gout && gin && B > 0 && C > 0 && H > 0 && W > 0 && r > 0 && gin_cs >= C * r * r,
              "pixel_unshuffle: bad args");
  hipLaunchKernelGGL(pixel_unshuffle_kernel, dim3(grid_for((int64_t)B * C * H * W * r * r)),
                     dim3(256), 0, (hipStream_t)stream, gout, gin, B, C, H, W, r, gin_cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// (slabs of 32 pixels: at compact's batch 2 — 8192 pixels — 256 workgroups of 8 dependent loads per thread instead of 32
// workgroups of 64: the pass was latency-bound at 20 us, 18 % of a configs[0] step)
static int prelu_blocks(int64_t npix) {
  int64_t b = (npix + 31) / 32;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

// the same two stages for MANY (dA, z, dslope) triples of one geometry in two launches: blockIdx.y = job, each job's
// partial rows in its own slice of the workspace — same sums in the same order as neosr_prelu_dslope
constexpr int DS_MAX = 32;
struct DslopeBatch {
  neosr_dslope_item it[DS_MAX];
};
__global__ __launch_bounds__(256) void prelu_dslope_many_stage1(const DslopeBatch bt, float* __restrict__ part, int64_t npix,
                                                                int C, int da_cs, int z_cs, int pix_per_block, int nblk) {
  __shared__ float sm[256];
  const neosr_dslope_item& j = bt.it[blockIdx.y];
  const float* __restrict__ dA = j.dA;
  const float* __restrict__ z = j.z;
  part += (int64_t)blockIdx.y * nblk * C;
  const int cpad = C <= 64 ? 64 : (C <= 128 ? 128 : 256);
  const int c = threadIdx.x % cpad, pl = threadIdx.x / cpad, npl = 256 / cpad;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
  const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
  float s = 0.f;
  if (c < C)
    for (int64_t p = p0 + pl; p < p1; p += npl) {
      const float zz = z[p * z_cs + c];
      s += dA[p * da_cs + c] * fminf(zz, 0.f);
    }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (pl == 0 && c < C) {
    float tot = 0.f;
    for (int k = 0; k < npl; ++k) tot += sm[k * cpad + c];
    part[(int64_t)blockIdx.x * C + c] = tot;
  }
}
__global__ __launch_bounds__(256) void prelu_dslope_many_stage2(const DslopeBatch bt, const float* __restrict__ part, int nblk,
                                                                int C) {
  __shared__ float red[4][64];
  float* __restrict__ dslope = bt.it[blockIdx.y].dslope;
  part += (int64_t)blockIdx.y * nblk * C;
  const int o = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int k = kl;
    for (; k + 12 < nblk; k += 16) {
      s0 += part[(int64_t)k * C + c];
      s1 += part[(int64_t)(k + 4) * C + c];
      s2 += part[(int64_t)(k + 8) * C + c];
      s3 += part[(int64_t)(k + 12) * C + c];
    }
    for (; k < nblk; k += 4) s0 += part[(int64_t)k * C + c];
  }
  red[kl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kl == 0 && c < C) dslope[c] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
}

extern "C" int64_t neosr_prelu_dslope_workspace_bytes(int64_t npix, int32_t C) {
  return (int64_t)prelu_blocks(npix) * C * 4;
}

extern "C" int neosr_prelu_dslope(const float* dA, const float* z, float* dslope, float* workspace,
                                  int64_t npix, int32_t C, int32_t da_cs, int32_t z_cs,
                                  int32_t accumulate, void* stream) {
  NEOSR_CHECK(dA && z && dslope && workspace && npix > 0 && C > 0 && C <= 256,
              "prelu_dslope: bad args (C<=256)");
  const int nblk = prelu_blocks(npix);
  const int ppb = (int)((npix + nblk - 1) / nblk);
  hipLaunchKernelGGL(prelu_dslope_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dA, z,
                     workspace, npix, C, da_cs, z_cs, ppb);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(prelu_dslope_stage2, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     workspace, dslope, nblk, C, accumulate);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_prelu_dslope_many(const neosr_dslope_item* items, int32_t n, float* workspace, int64_t npix,
                                       int32_t C, int32_t da_cs, int32_t z_cs, void* stream) {
  NEOSR_CHECK(items && n > 0 && workspace && npix > 0 && C > 0 && C <= 256, "prelu_dslope_many: bad args (C<=256)");
  const int nblk = prelu_blocks(npix);
  const int ppb = (int)((npix + nblk - 1) / nblk);
  for (int i0 = 0; i0 < n; i0 += DS_MAX) {
    const int cnt = n - i0 < DS_MAX ? n - i0 : DS_MAX;
    DslopeBatch bt;
    memset(&bt, 0, sizeof(bt));
    for (int i = 0; i < cnt; ++i) {
      NEOSR_CHECK(items[i0 + i].dA && items[i0 + i].z && items[i0 + i].dslope, "prelu_dslope_many: null tensor");
      bt.it[i] = items[i0 + i];
    }
    float* part = workspace + (int64_t)i0 * nblk * C;
    hipLaunchKernelGGL(prelu_dslope_many_stage1, dim3(nblk, cnt), dim3(256), 0, (hipStream_t)stream, bt, part, npix, C,
                       da_cs, z_cs, ppb, nblk);
    hipLaunchKernelGGL(prelu_dslope_many_stage2, dim3((C + 63) / 64, cnt), dim3(256), 0, (hipStream_t)stream, bt, part,
                       nblk, C);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_l1_loss_fwd(const float* pred, const float* target, int64_t n,
                                 float loss_weight, float* loss_out, float* workspace,
                                 void* stream) {
  NEOSR_CHECK(pred && target && loss_out && workspace && n > 0, "l1_loss_fwd: bad args");
  NEOSR_CHECK((uintptr_t)pred % 16 == 0 && (uintptr_t)target % 16 == 0, "l1_loss_fwd: unaligned");
  const int nb = grid_for((n >> 2) + 1, RED_BLOCKS);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target,
                     n, workspace);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nb,
                     loss_weight / (float)n, loss_out, 0);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_l1_loss_bwd(const float* pred, const float* target, const float* grad_out,
                                 int64_t n, float loss_weight, float* grad_pred, void* stream) {
  NEOSR_CHECK(pred && target && grad_out && grad_pred && n > 0, "l1_loss_bwd: bad args");
  NEOSR_CHECK((uintptr_t)pred % 16 == 0 && (uintptr_t)target % 16 == 0 &&
                  (uintptr_t)grad_pred % 16 == 0,
              "l1_loss_bwd: unaligned");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for((n >> 2) + 1)), dim3(256), 0, (hipStream_t)stream,
                     pred, target, grad_out, n, loss_weight / (float)n, grad_pred);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_grad_norm(const float* grad, int64_t n, float grad_scale, float* norm_ws,
                               void* stream) {
  NEOSR_CHECK(grad && norm_ws && n > 0, "grad_norm: bad args");
  NEOSR_CHECK((uintptr_t)grad % 16 == 0, "grad_norm: unaligned");
  const int nb = grid_for((n >> 2) + 1, RED_BLOCKS);
  // partials live at ws[4..], result at ws[0]
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, grad, n,
                     grad_scale, norm_ws + 4);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, norm_ws + 4,
                     nb, 1.0f, norm_ws, 1);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_adamw_step(const neosr_adamw_desc* dp, void* stream) {
  const neosr_adamw_desc& d = *dp;
  NEOSR_CHECK(d.param && d.grad && d.exp_avg && d.exp_avg_sq && d.n > 0 && d.step >= 1,
              "adamw_step: bad args");
  NEOSR_CHECK((uintptr_t)d.param % 16 == 0 && (uintptr_t)d.grad % 16 == 0 &&
                  (uintptr_t)d.exp_avg % 16 == 0 && (uintptr_t)d.exp_avg_sq % 16 == 0 &&
                  (uintptr_t)d.ema % 16 == 0,
              "adamw_step: unaligned arena");
  if (d.max_norm > 0.f) {
    NEOSR_CHECK(d.norm_ws, "adamw_step: clipping needs norm_ws");
    if (int rc = neosr_grad_norm(d.grad, d.n, d.grad_scale, d.norm_ws, stream)) return rc;
  }
  AdamArgs a;
  a.d = d;
  a.bc1 = (float)(1.0 - pow((double)d.beta1, (double)d.step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)d.beta2, (double)d.step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((d.n >> 2) + 1)), dim3(256), 0,
                     (hipStream_t)stream, a);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
