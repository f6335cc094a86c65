// layers.hip — the non-conv layers and losses of the GAN / perceptual branch of the hot path
// (U-Net-SN discriminator, VGG19 feature extractor, chc / BCE losses) for gfx950.  All HBM-bound:
// channels-last activations, float4 along channels where the shape allows, fixed-order two-stage
// reductions.  Reference call sites: neosr/archs/unet_arch.py:36-67 (bilinear x2, skip adds,
// 4x4/s2 convs via space-to-depth), torch.nn.utils.spectral_norm (unet_arch.py:21-34),
// neosr/archs/vgg_arch.py:159-199 (input norm, max-pool), neosr/losses/basic_loss.py:132-219 (chc),
// neosr/losses/gan_loss.py:45-82 (BCEWithLogits vs a constant label).
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t work_items, int cap = 4096) {
  int64_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_reduce_sum(v);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int np,
                                                           float scale, float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  const float r = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = r * scale;
}

// ------------------------------------------------------------------ space-to-depth (r = 2), NHWC
// out[b, Y, X, (dy*2+dx)*C + c] = in[b, 2Y+dy, 2X+dx, c]   (dir = 0), inverse for dir = 1
template <int V>
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                  int B, int H2, int W2, int C, int dir) {
  const int Cv = C / V;
  const int64_t total = (int64_t)B * H2 * W2 * 4 * Cv;  // H2, W2 = low-res size
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % Cv) * V;
    int64_t t = e / Cv;
    const int q = (int)(t & 3);
    t >>= 2;
    const int X = (int)(t % W2);
    t /= W2;
    const int Y = (int)(t % H2);
    const int b = (int)(t / H2);
    const int64_t hi = (((int64_t)b * 2 * H2 + 2 * Y + (q >> 1)) * 2 * W2 + 2 * X + (q & 1)) * C + c;
    const int64_t lo = e * V;
    const int64_t src = dir == 0 ? hi : lo, dst = dir == 0 ? lo : hi;
    if (V == 4) *reinterpret_cast<float4*>(out + dst) = *reinterpret_cast<const float4*>(in + src);
    else out[dst] = in[src];
  }
}

// depth-to-space of a gradient fused with what follows it in the U-Net's backward pass (unet_arch.py:36-60: x0 / x1 / x2
// feed a 4x4 / stride-2 convolution AND a skip addition): out = (d2s(g) + skip) * (y > 0 ? 1 : slope) — the sum autograd
// would form in a pass of its own and the LeakyReLU derivative of the producing layer, in the pass that re-lays the
// gradient out anyway.  Same expressions as the separate passes: bit-identical.
template <int V>
__global__ __launch_bounds__(256) void d2s_fused_kernel(const float* __restrict__ g, const float* __restrict__ skip,
                                                        const float* __restrict__ y, float slope, float* __restrict__ out,
                                                        int B, int H2, int W2, int C) {
  const int Cv = C / V;
  const int64_t total = (int64_t)B * H2 * W2 * 4 * Cv;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % Cv) * V;
    int64_t t = e / Cv;
    const int q = (int)(t & 3);
    t >>= 2;
    const int X = (int)(t % W2);
    t /= W2;
    const int Y = (int)(t % H2);
    const int b = (int)(t / H2);
    const int64_t hi = (((int64_t)b * 2 * H2 + 2 * Y + (q >> 1)) * 2 * W2 + 2 * X + (q & 1)) * C + c;
    const int64_t lo = e * V;
    if (V == 4) {
      float4 v = *reinterpret_cast<const float4*>(g + lo);
      if (skip) {
        const float4 a = *reinterpret_cast<const float4*>(skip + hi);
        v.x = a.x + v.x; v.y = a.y + v.y; v.z = a.z + v.z; v.w = a.w + v.w;
      }
      if (y) {
        const float4 m = *reinterpret_cast<const float4*>(y + hi);
        v.x = m.x > 0.f ? v.x : v.x * slope; v.y = m.y > 0.f ? v.y : v.y * slope;
        v.z = m.z > 0.f ? v.z : v.z * slope; v.w = m.w > 0.f ? v.w : v.w * slope;
      }
      *reinterpret_cast<float4*>(out + hi) = v;
    } else {
      float v = g[lo];
      if (skip) v = skip[hi] + v;
      if (y) v = y[hi] > 0.f ? v : v * slope;
      out[hi] = v;
    }
  }
}

// ------------------------------------------------------------------ bilinear x2, align_corners=False
// src = (dst + 0.5) / 2 - 0.5 clamped at 0: taps {i0, i1} with weights {1-l, l}
__device__ __forceinline__ void bil_tap(int o, int n_in, int& i0, int& i1, float& l) {
  float s = 0.5f * (o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l = s - i0;
}

// V = 4: one thread per channel quad (16-byte accesses; C % 4 == 0 and 16-byte aligned buffers), else V = 1
template <int V>
struct VecF {
  float v[V];
  __device__ __forceinline__ static VecF load(const float* p) {
    VecF r;
    if (V == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      r.v[0] = t.x; r.v[1 % V] = t.y; r.v[2 % V] = t.z; r.v[3 % V] = t.w;
    } else {
      r.v[0] = *p;
    }
    return r;
  }
  __device__ __forceinline__ void store(float* p) const {
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
    else *p = v[0];
  }
};

template <int V>
__global__ __launch_bounds__(256) void bilinear_up2_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int B, int H,
                                                           int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, Cv = C / V;
  const int64_t total = (int64_t)B * Ho * Wo * Cv;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % Cv) * V;
    int64_t t = e / Cv;
    const int ox = (int)(t % Wo);
    t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bil_tap(oy, H, y0, y1, ly);
    bil_tap(ox, W, x0, x1, lx);
    const float* s = in + (int64_t)b * H * W * C + c;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const VecF<V> a00 = VecF<V>::load(s + ((int64_t)y0 * W + x0) * C), a01 = VecF<V>::load(s + ((int64_t)y0 * W + x1) * C);
    const VecF<V> a10 = VecF<V>::load(s + ((int64_t)y1 * W + x0) * C), a11 = VecF<V>::load(s + ((int64_t)y1 * W + x1) * C);
    VecF<V> o;
#pragma unroll
    for (int k = 0; k < V; ++k)
      o.v[k] = hy * (hx * a00.v[k] + lx * a01.v[k]) + ly * (hx * a10.v[k] + lx * a11.v[k]);
    o.store(out + e * V);
  }
}

// adjoint, gather form (deterministic): each input pixel sums the <= 4x4 outputs that read it
template <int V>
__global__ __launch_bounds__(256) void bilinear_up2_bwd_kernel(const float* __restrict__ gout,
                                                               float* __restrict__ gin, int B, int H,
                                                               int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, Cv = C / V;
  const int64_t total = (int64_t)B * H * W * Cv;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % Cv) * V;
    int64_t t = e / Cv;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const float* g = gout + (int64_t)b * Ho * Wo * C + c;
    VecF<V> acc;
#pragma unroll
    for (int k = 0; k < V; ++k) acc.v[k] = 0.f;
    for (int oy = max(2 * y - 2, 0); oy <= min(2 * y + 2, Ho - 1); ++oy) {
      int y0, y1;
      float ly;
      bil_tap(oy, H, y0, y1, ly);
      const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = max(2 * x - 2, 0); ox <= min(2 * x + 2, Wo - 1); ++ox) {
        int x0, x1;
        float lx;
        bil_tap(ox, W, x0, x1, lx);
        const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
        if (wx != 0.f) {
          const VecF<V> gv = VecF<V>::load(g + ((int64_t)oy * Wo + ox) * C);
#pragma unroll
          for (int k = 0; k < V; ++k) acc.v[k] += wy * wx * gv.v[k];
        }
      }
    }
    acc.store(gin + e * V);
  }
}

// ------------------------------------------------------------------ max-pool 2x2/s2, NHWC
__global__ __launch_bounds__(256) void maxpool2_kernel(const float* __restrict__ in,
                                                       float* __restrict__ out, int B, int Ho, int Wo,
                                                       int C) {
  const int64_t total = (int64_t)B * Ho * Wo * C;
  const int W = 2 * Wo;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int x = (int)(t % Wo);
    t /= Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float* s = in + (((int64_t)b * 2 * Ho + 2 * y) * W + 2 * x) * C + c;
    out[e] = fmaxf(fmaxf(s[0], s[C]), fmaxf(s[(int64_t)W * C], s[(int64_t)W * C + C]));
  }
}

// gradient goes to the first window element (row-major) that equals the max, like ATen's indices
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ in,
                                                           const float* __restrict__ gout,
                                                           float* __restrict__ gin, int B, int Ho,
                                                           int Wo, int C) {
  const int64_t total = (int64_t)B * Ho * Wo * C;
  const int W = 2 * Wo;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int x = (int)(t % Wo);
    t /= Wo;
    const int y = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const int64_t base = (((int64_t)b * 2 * Ho + 2 * y) * W + 2 * x) * C + c;
    const int64_t off[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
    float m = in[base];
    int am = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float v = in[base + off[k]];
      if (v > m) { m = v; am = k; }
    }
    const float g = gout[e];
#pragma unroll
    for (int k = 0; k < 4; ++k) gin[base + off[k]] = (k == am) ? g : 0.f;
  }
}

// ------------------------------------------------------------------ elementwise
// VEC: 16-byte accesses (n a multiple of 4, 16-byte aligned pointers) — the activation maps of the U-Net discriminator and
// the VGG taps at B = 32 are 100-500 MB: four bytes per lane and trip left these passes at half the HBM rate
template <bool VEC>
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a,
                                                  const float* __restrict__ b, float* __restrict__ out,
                                                  int64_t n) {
  if (VEC) {
    const int64_t nq = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
      const float4 u = reinterpret_cast<const float4*>(a)[q], v = reinterpret_cast<const float4*>(b)[q];
      reinterpret_cast<float4*>(out)[q] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = a[e] + b[e];
}

// out = x > 0 ? x : x*slope (fwd);  gin = y_or_x > 0 ? g : g*slope (bwd)
template <bool VEC>
__global__ __launch_bounds__(256) void lrelu_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ g, float slope,
                                                    float* __restrict__ out, int64_t n) {
  if (VEC) {
    const int64_t nq = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
      const float4 xv = reinterpret_cast<const float4*>(x)[q];
      const float4 v = g ? reinterpret_cast<const float4*>(g)[q] : xv;
      reinterpret_cast<float4*>(out)[q] = make_float4(xv.x > 0.f ? v.x : v.x * slope, xv.y > 0.f ? v.y : v.y * slope,
                                                      xv.z > 0.f ? v.z : v.z * slope, xv.w > 0.f ? v.w : v.w * slope);
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const float v = g ? g[e] : x[e];
    out[e] = x[e] > 0.f ? v : v * slope;
  }
}
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// NCHW image -> NHWC with per-tensor affine (x - mean) / std  (VGG input norm) ; bwd: /std
__global__ __launch_bounds__(256) void norm_nchw_to_nhwc_kernel(const float* __restrict__ in,
                                                                float* __restrict__ out, int B, int C,
                                                                int HW, int out_cs, float mean,
                                                                float inv_std, int dir) {
  const int64_t total = (int64_t)B * C * HW;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int p = (int)(e % HW);
    const int64_t t = e / HW;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int64_t o = ((int64_t)b * HW + p) * out_cs + c;
    if (dir == 0) out[o] = (in[e] - mean) * inv_std;  // in = NCHW image, out = NHWC
    else out[e] = in[o] * inv_std;                     // in = NHWC grad,  out = NCHW grad
  }
}

// ------------------------------------------------------------------ chc loss (lambda = 0)
// t = |d| (l1) or sqrt(d^2 + 1e-12) (huber); loss = w * mean(clamp(t, lo, hi)); d = (a - b) * pre
__global__ __launch_bounds__(256) void chc_partial_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, int64_t n,
                                                          float pre, int huber, float lo, float hi,
                                                          float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] * pre - b[i] * pre;
    const float t = huber ? sqrtf(d * d + 1e-12f) : fabsf(d);
    s += fminf(fmaxf(t, lo), hi);
  }
  const float r = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void chc_bwd_kernel(const float* __restrict__ a,
                                                      const float* __restrict__ b,
                                                      const float* __restrict__ gout, int64_t n,
                                                      float pre, int huber, float lo, float hi,
                                                      float scale, float* __restrict__ ga,
                                                      int accumulate) {
  const float gs = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] * pre - b[i] * pre;
    float t, dt;
    if (huber) {
      t = sqrtf(d * d + 1e-12f);
      dt = d / t;
    } else {
      t = fabsf(d);
      dt = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    const float g = (t >= lo && t <= hi) ? gs * dt * pre : 0.f;
    ga[i] = accumulate ? ga[i] + g : g;
  }
}


// ------------------------------------------------------------------ chc loss with the cosine term (lambda != 0)
// basic_loss.py:192-219 on NCHW tensors: c = mean over pixels of (1 - cos_sim over channels), then
// loss = w * mean(clamp(t + lambda * c, lo, hi)).  aux[0] = c, aux[1] = number of elements inside the clamp range
// (both stay on the device: the backward needs them, the host never does).
__device__ __forceinline__ void pixel_cos(const float* __restrict__ a, const float* __restrict__ b, int C, int64_t hw,
                                          int64_t base, float eps, float& na, float& nb, float& cs) {
  float saa = 0.f, sbb = 0.f;
  for (int c = 0; c < C; ++c) {
    const float x = a[base + c * hw], y = b[base + c * hw];
    saa += x * x;
    sbb += y * y;
  }
  na = fmaxf(sqrtf(saa), eps);  // ATen cosine_similarity: sum((x / max(|x|, eps)) * (y / max(|y|, eps)))
  nb = fmaxf(sqrtf(sbb), eps);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += (a[base + c * hw] / na) * (b[base + c * hw] / nb);
  cs = s;
}

__global__ __launch_bounds__(256) void cos_nchw_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                               int64_t npix, int C, int64_t hw, float eps,
                                                               float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / hw, r = i - n * hw;
    float na, nb, cs;
    pixel_cos(a, b, C, hw, n * C * hw + r, eps, na, nb, cs);
    s += 1.f - cs;
  }
  const float rsum = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = rsum;
}

__global__ __launch_bounds__(256) void chc_cos_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              int64_t n, int huber, float lo, float hi, float lambda,
                                                              const float* __restrict__ aux, float* __restrict__ part,
                                                              float* __restrict__ cnt_part) {
  __shared__ float sm[4];
  const float off = lambda * aux[0];
  float s = 0.f, k = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    const float t = (huber ? sqrtf(d * d + 1e-12f) : fabsf(d)) + off;
    s += fminf(fmaxf(t, lo), hi);
    k += (t >= lo && t <= hi) ? 1.f : 0.f;
  }
  const float rs = block_sum_256(s, sm);
  const float rk = block_sum_256(k, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = rs;
    cnt_part[blockIdx.x] = rk;
  }
}

// d loss / d a: the elementwise part where the clamp passes, plus lambda * (#elements inside the clamp) * d c / d a
__global__ __launch_bounds__(256) void chc_cos_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ gout, int64_t npix, int C,
                                                          int64_t hw, int huber, float lo, float hi, float lambda,
                                                          float scale, float eps, const float* __restrict__ aux,
                                                          float* __restrict__ ga) {
  const float gs = gout[0] * scale;
  const float off = lambda * aux[0];
  const float kc = -gs * lambda * aux[1] / (float)npix;  // d/d cos of this pixel
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / hw, r = i - n * hw, base = n * C * hw + r;
    float na, nb, cs;
    pixel_cos(a, b, C, hw, base, eps, na, nb, cs);
    for (int c = 0; c < C; ++c) {
      const float x = a[base + c * hw], y = b[base + c * hw];
      const float d = x - y;
      float t, dt;
      if (huber) {
        t = sqrtf(d * d + 1e-12f);
        dt = d / t;
      } else {
        t = fabsf(d);
        dt = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      }
      t += off;
      float g = (t >= lo && t <= hi) ? gs * dt : 0.f;
      // cos = sum_c (x_c / na)(y_c / nb); for na above eps: d cos / d x_c = y_c / (na nb) - cos * x_c / na^2
      const float dcos = na > eps ? (y / (na * nb) - cs * x / (na * na)) : y / (na * nb);
      ga[base + c * hw] = g + kc * dcos;
    }
  }
}

// ------------------------------------------------------------------ BCE-with-logits vs constant t
// mean(max(x,0) - x*t + log1p(exp(-|x|)))
__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ x, int64_t n,
                                                          float tval, float* __restrict__ part,
                                                          float* __restrict__ xsum_part) {
  __shared__ float sm[4];
  float s = 0.f, sx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * tval + log1pf(expf(-fabsf(v)));
    sx += v;
  }
  const float r = block_sum_256(s, sm);
  const float rx = block_sum_256(sx, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = r;
    xsum_part[blockIdx.x] = rx;
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ gout, int64_t n,
                                                      float tval, float scale, float* __restrict__ gx) {
  const float gs = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    gx[i] = gs * (1.f / (1.f + expf(-v)) - tval);
  }
}

// ------------------------------------------------------------------ spectral norm
// torch.nn.utils.spectral_norm, one power iteration, W (rows, cols):
//   v = normalize(W^T u); u = normalize(W v); sigma = u . (W v); w = W / sigma
// Single workgroup per stage is enough (<= 512 x 4608).  Stage kernels:
// v = W^T u: 64 columns x 4 row lanes per workgroup (a thread walks rows ty, ty + 4, ... of its column with eight loads in
// flight; the four lanes of a column are summed through LDS in a fixed order).  The one-thread-per-column form kept
// 16 KB in flight on a 512 x 4096 matrix and took 70-200 us per call.
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ W,
                                                      const float* __restrict__ u,
                                                      float* __restrict__ v, int rows, int cols) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;  // column
  float s = 0.f;
  if (j < cols) {
#pragma unroll 8
    for (int i = ty; i < rows; i += 4) s += W[(int64_t)i * cols + j] * u[i];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && j < cols) v[j] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W,
                                                     const float* __restrict__ v,
                                                     float* __restrict__ u, int rows, int cols) {
  __shared__ float sm[4];
  const int i = blockIdx.x;  // row
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) s += W[(int64_t)i * cols + j] * v[j];
  const float r = block_sum_256(s, sm);
  if (threadIdx.x == 0) u[i] = r;
}

// x /= max(||x||, eps); optionally also emit dot = x_raw . y  (single workgroup)
__global__ __launch_bounds__(256) void sn_normalize_kernel(float* __restrict__ x, int n, float eps,
                                                           float* __restrict__ norm_out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  const float nrm = sqrtf(block_sum_256(s, sm));
  const float d = fmaxf(nrm, eps);
  for (int i = threadIdx.x; i < n; i += 256) x[i] = x[i] / d;
  if (norm_out && threadIdx.x == 0) norm_out[0] = nrm;
}

// sigma = u . t (t = W v un-normalised), single workgroup
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ b, int n,
                                                     float* __restrict__ out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += a[i] * b[i];
  const float r = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = r;
}

__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ W,
                                                       const float* __restrict__ sigma,
                                                       float* __restrict__ out, int64_t n) {
  const float inv = 1.f / sigma[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = W[e] * inv;
}

// <gw, w> partials
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a,
                                                          const float* __restrict__ b, int64_t n,
                                                          float* __restrict__ part) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += a[i] * b[i];
  const float r = block_sum_256(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}

// gW_orig = (gw - <gw, w> u v^T) / sigma
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ gw,
                                                     const float* __restrict__ u,
                                                     const float* __restrict__ v,
                                                     const float* __restrict__ sigma,
                                                     const float* __restrict__ dot,
                                                     float* __restrict__ gorig, int rows, int cols) {
  const int64_t n = (int64_t)rows * cols;
  const float inv = 1.f / sigma[0], dt = dot[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int i = (int)(e / cols), j = (int)(e - (int64_t)i * cols);
    gorig[e] = (gw[e] - dt * u[i] * v[j]) * inv;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int neosr_space_to_depth2(const float* in, float* out, int32_t B, int32_t Hlo,
                                     int32_t Wlo, int32_t C, int32_t inverse, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && Hlo > 0 && Wlo > 0 && C > 0, "space_to_depth2: bad args");
  if (C % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 16 == 0)
    hipLaunchKernelGGL(s2d_kernel<4>, dim3(grid_for((int64_t)B * Hlo * Wlo * C)), dim3(256), 0, ST, in, out, B,
                       Hlo, Wlo, C, inverse);
  else
    hipLaunchKernelGGL(s2d_kernel<1>, dim3(grid_for((int64_t)B * Hlo * Wlo * 4 * C)), dim3(256), 0, ST, in, out,
                       B, Hlo, Wlo, C, inverse);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_depth_to_space2_fused(const float* g, const float* skip, const float* y, float slope, float* out,
                                           int32_t B, int32_t Hlo, int32_t Wlo, int32_t C, void* stream) {
  NEOSR_CHECK(g && out && B > 0 && Hlo > 0 && Wlo > 0 && C > 0, "depth_to_space2_fused: bad args");
  const bool v4 = C % 4 == 0 && (uintptr_t)g % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)skip % 16 == 0 &&
                  (uintptr_t)y % 16 == 0;
  if (v4)
    hipLaunchKernelGGL(d2s_fused_kernel<4>, dim3(grid_for((int64_t)B * Hlo * Wlo * C)), dim3(256), 0, ST, g, skip, y, slope,
                       out, B, Hlo, Wlo, C);
  else
    hipLaunchKernelGGL(d2s_fused_kernel<1>, dim3(grid_for((int64_t)B * Hlo * Wlo * 4 * C)), dim3(256), 0, ST, g, skip, y,
                       slope, out, B, Hlo, Wlo, C);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_bilinear_up2(const float* in, float* out, int32_t B, int32_t H, int32_t W,
                                  int32_t C, int32_t backward, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0, "bilinear_up2: bad args");
  const bool v4 = (C % 4 == 0) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (!backward) {
    if (v4) hipLaunchKernelGGL(bilinear_up2_kernel<4>, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, ST, in, out, B, H, W, C);
    else hipLaunchKernelGGL(bilinear_up2_kernel<1>, dim3(grid_for((int64_t)B * 4 * H * W * C)), dim3(256), 0, ST, in, out, B, H, W, C);
  } else {
    if (v4) hipLaunchKernelGGL(bilinear_up2_bwd_kernel<4>, dim3(grid_for((int64_t)B * H * W * C / 4)), dim3(256), 0, ST, in, out, B, H, W, C);
    else hipLaunchKernelGGL(bilinear_up2_bwd_kernel<1>, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, ST, in, out, B, H, W, C);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_maxpool2(const float* in, const float* gout, float* out, int32_t B, int32_t Ho,
                              int32_t Wo, int32_t C, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && Ho > 0 && Wo > 0 && C > 0, "maxpool2: bad args");
  if (!gout)
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C)), dim3(256), 0, ST,
                       in, out, B, Ho, Wo, C);
  else
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * C)), dim3(256), 0,
                       ST, in, gout, out, B, Ho, Wo, C);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  NEOSR_CHECK(a && b && out && n > 0, "add: bad args");
  if (n % 4 == 0 && al16(a) && al16(b) && al16(out))
    hipLaunchKernelGGL(add_kernel<true>, dim3(grid_for(n / 4, 8192)), dim3(256), 0, ST, a, b, out, n);
  else
    hipLaunchKernelGGL(add_kernel<false>, dim3(grid_for(n)), dim3(256), 0, ST, a, b, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_leaky_relu(const float* x, const float* g, float slope, float* out, int64_t n,
                                void* stream) {
  NEOSR_CHECK(x && out && n > 0, "leaky_relu: bad args");
  if (n % 4 == 0 && al16(x) && al16(g) && al16(out))
    hipLaunchKernelGGL(lrelu_kernel<true>, dim3(grid_for(n / 4, 8192)), dim3(256), 0, ST, x, g, slope, out, n);
  else
    hipLaunchKernelGGL(lrelu_kernel<false>, dim3(grid_for(n)), dim3(256), 0, ST, x, g, slope, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_norm_nchw_nhwc(const float* in, float* out, int32_t B, int32_t C, int32_t H,
                                    int32_t W, int32_t cs, float mean, float std, int32_t backward,
                                    void* stream) {
  NEOSR_CHECK(in && out && B > 0 && C > 0 && H > 0 && W > 0 && cs >= C && std != 0.f, "norm: bad args");
  hipLaunchKernelGGL(norm_nchw_to_nhwc_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0,
                     ST, in, out, B, C, H * W, cs, mean, 1.f / std, backward);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_chc_loss_fwd(const float* a, const float* b, int64_t n, float pre, int32_t huber,
                                  float clip_min, float clip_max, float loss_weight, float* loss_out,
                                  float* workspace, void* stream) {
  NEOSR_CHECK(a && b && loss_out && workspace && n > 0, "chc_loss_fwd: bad args");
  const int nb = grid_for(n, 1024);
  hipLaunchKernelGGL(chc_partial_kernel, dim3(nb), dim3(256), 0, ST, a, b, n, pre, huber, clip_min,
                     clip_max, workspace);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace, nb,
                     loss_weight / (float)n, loss_out);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_chc_loss_bwd(const float* a, const float* b, const float* grad_out, int64_t n,
                                  float pre, int32_t huber, float clip_min, float clip_max,
                                  float loss_weight, float* grad_a, int32_t accumulate, void* stream) {
  NEOSR_CHECK(a && b && grad_out && grad_a && n > 0, "chc_loss_bwd: bad args");
  hipLaunchKernelGGL(chc_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, a, b, grad_out, n, pre,
                     huber, clip_min, clip_max, loss_weight / (float)n, grad_a, accumulate);
  NEOSR_LAUNCH_CHECK();
  return 0;
}


extern "C" int neosr_chc_cos_loss_fwd(const float* a, const float* b, int32_t N, int32_t C, int64_t hw, int32_t huber,
                                      float clip_min, float clip_max, float loss_lambda, float loss_weight,
                                      float cos_eps, float* loss_out, float* aux, float* workspace, void* stream) {
  NEOSR_CHECK(a && b && loss_out && aux && workspace && N > 0 && C > 0 && hw > 0, "chc_cos_loss_fwd: bad args");
  const int64_t npix = (int64_t)N * hw, n = npix * C;
  const int nbp = grid_for(npix, 1024), nb = grid_for(n, 1024);
  hipLaunchKernelGGL(cos_nchw_partial_kernel, dim3(nbp), dim3(256), 0, ST, a, b, npix, C, hw, cos_eps, workspace);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace, nbp, 1.f / (float)npix, aux);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(chc_cos_partial_kernel, dim3(nb), dim3(256), 0, ST, a, b, n, huber, clip_min, clip_max,
                     loss_lambda, aux, workspace + 1024, workspace + 2048);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace + 1024, nb, loss_weight / (float)n,
                     loss_out);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace + 2048, nb, 1.f, aux + 1);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_chc_cos_loss_bwd(const float* a, const float* b, const float* grad_out, int32_t N, int32_t C,
                                      int64_t hw, int32_t huber, float clip_min, float clip_max, float loss_lambda,
                                      float loss_weight, float cos_eps, const float* aux, float* grad_a,
                                      void* stream) {
  NEOSR_CHECK(a && b && grad_out && aux && grad_a && N > 0 && C > 0 && hw > 0, "chc_cos_loss_bwd: bad args");
  const int64_t npix = (int64_t)N * hw;
  hipLaunchKernelGGL(chc_cos_bwd_kernel, dim3(grid_for(npix)), dim3(256), 0, ST, a, b, grad_out, npix, C, hw, huber,
                     clip_min, clip_max, loss_lambda, loss_weight / (float)(npix * C), cos_eps, aux, grad_a);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_bce_logits_fwd(const float* x, int64_t n, float target, float loss_weight,
                                    float* loss_out, float* mean_out, float* workspace, void* stream) {
  NEOSR_CHECK(x && loss_out && workspace && n > 0, "bce_logits_fwd: bad args");
  const int nb = grid_for(n, 1024);
  hipLaunchKernelGGL(bce_partial_kernel, dim3(nb), dim3(256), 0, ST, x, n, target, workspace,
                     workspace + 1024);
  NEOSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace, nb,
                     loss_weight / (float)n, loss_out);
  NEOSR_LAUNCH_CHECK();
  if (mean_out) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace + 1024, nb,
                       1.f / (float)n, mean_out);
    NEOSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int neosr_bce_logits_bwd(const float* x, const float* grad_out, int64_t n, float target,
                                    float loss_weight, float* grad_x, void* stream) {
  NEOSR_CHECK(x && grad_out && grad_x && n > 0, "bce_logits_bwd: bad args");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, x, grad_out, n, target,
                     loss_weight / (float)n, grad_x);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_out,
                                       float* sigma, float* scratch_rows, int32_t rows, int32_t cols,
                                       int32_t update_uv, float eps, void* stream) {
  NEOSR_CHECK(w_orig && u && v && w_out && sigma && scratch_rows && rows > 0 && cols > 0,
              "spectral_norm_fwd: bad args");
  if (update_uv) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(ceil_div(cols, 64)), dim3(256), 0, ST, w_orig, u, v, rows, cols);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, ST, v, cols, eps, (float*)nullptr);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, ST, w_orig, v, u, rows, cols);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, ST, u, rows, eps, (float*)nullptr);
  }
  // sigma = u . (W v)
  hipLaunchKernelGGL(sn_w_v_kernel, dim3(rows), dim3(256), 0, ST, w_orig, v, scratch_rows, rows, cols);
  hipLaunchKernelGGL(sn_dot_kernel, dim3(1), dim3(256), 0, ST, u, scratch_rows, rows, sigma);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, ST, w_orig,
                     sigma, w_out, (int64_t)rows * cols);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_spectral_norm_bwd(const float* gw, const float* w, const float* u, const float* v,
                                       const float* sigma, float* g_orig, float* workspace,
                                       int32_t rows, int32_t cols, void* stream) {
  NEOSR_CHECK(gw && w && u && v && sigma && g_orig && workspace && rows > 0 && cols > 0,
              "spectral_norm_bwd: bad args");
  const int64_t n = (int64_t)rows * cols;
  const int nb = grid_for(n, 1024);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, ST, gw, w, n, workspace);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, ST, workspace, nb, 1.f, workspace + 1024);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, gw, u, v, sigma,
                     workspace + 1024, g_orig, rows, cols);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
