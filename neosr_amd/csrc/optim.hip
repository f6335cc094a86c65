// optim.hip — fused Schedule-Free Adan step on flat HBM arenas for gfx950.
//
// Replaces, for `optim_g.type = "adan_sf"` (neosr/models/base.py:164-165), the 17 `_foreach_*`
// sweeps of `_multi_tensor_adan` (neosr/optimizers/adan_sf.py:261-330) + `clip_grad_norm_`
// (models/image.py:533-544) + `AveragedModel.update_parameters` (image.py:661-662) by
// `neosr_grad_norm` + ONE sweep: 7 arenas read, 6 written per element.
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

struct AdanArgs {
  neosr_adan_desc d;
  float decay;           // 1 - lr * weight_decay
  float step_size;       // sf: lr * bc1 * (1 - ckp1)      else lr / bc1
  float step_size_diff;  // sf: lr * beta2 / bc2 * (1 - ckp1)  else lr * beta2 / bc2
  float bc3_sqrt;
};

// torch.lerp(a, b, w): a + w (b - a) for w < 0.5, else b - (b - a)(1 - w)   (ATen lerp formula)
__device__ __forceinline__ float lerp_aten(float a, float b, float w) {
  const float diff = b - a;
  return w < 0.5f ? a + w * diff : b - diff * (1.f - w);
}

__device__ __forceinline__ void adan_one(float& p, float g, float& m, float& n, float& df, float* z,
                                         float& npg, float* ema, const AdanArgs& a, float clip) {
  const neosr_adan_desc& d = a.d;
  g *= clip;
  if (d.first_step) npg = -g;  // adan_sf.py:225-226
  npg = npg + g;               // g_t - g_{t-1}
  m = m * d.beta1 + g * (1.f - d.beta1);
  df = df * d.beta2 + npg * (1.f - d.beta2);
  npg = npg * d.beta2 + g;
  n = n * d.beta3 + (1.f - d.beta3) * npg * npg;
  const float denom = sqrtf(n) / a.bc3_sqrt + d.eps;
  p *= a.decay;
  if (d.schedule_free) p = lerp_aten(p, *z, d.ckp1);
  p = p - a.step_size * (m / denom);
  p = p - a.step_size_diff * (df / denom);
  if (d.schedule_free) *z = *z - d.lr * g;
  npg = -g;
  if (ema) {
    if (d.ema_decay < 0.f)
      *ema = p;
    else
      *ema = *ema + (p - *ema) * (1.f - d.ema_decay);
  }
}

__global__ __launch_bounds__(256) void adan_sf_kernel(const AdanArgs a) {
  const neosr_adan_desc& d = a.d;
  float clip = d.grad_scale;
  if (d.max_norm > 0.f) clip *= fminf(d.max_norm / (d.norm_ws[0] + 1e-6f), 1.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    float p = d.param[i], m = d.exp_avg[i], n = d.exp_avg_sq[i], df = d.exp_avg_diff[i];
    float npg = d.neg_pre_grad[i];
    float z = d.schedule_free ? d.z[i] : 0.f;
    float e = d.ema ? d.ema[i] : 0.f;
    adan_one(p, d.grad[i], m, n, df, &z, npg, d.ema ? &e : nullptr, a, clip);
    d.param[i] = p;
    d.exp_avg[i] = m;
    d.exp_avg_sq[i] = n;
    d.exp_avg_diff[i] = df;
    d.neg_pre_grad[i] = npg;
    if (d.schedule_free) d.z[i] = z;
    if (d.ema) d.ema[i] = e;
  }
}

__global__ __launch_bounds__(256) void lerp_kernel(float* __restrict__ p, const float* __restrict__ end,
                                                   int64_t n, float w) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    p[i] = lerp_aten(p[i], end[i], w);
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int neosr_adan_sf_step(const neosr_adan_desc* dp, void* stream) {
  NEOSR_CHECK(dp, "adan_sf_step: null descriptor");
  const neosr_adan_desc& d = *dp;
  NEOSR_CHECK(d.param && d.grad && d.exp_avg && d.exp_avg_sq && d.exp_avg_diff && d.neg_pre_grad && d.n > 0 &&
                  d.step >= 1, "adan_sf_step: bad args");
  NEOSR_CHECK(!d.schedule_free || d.z, "adan_sf_step: schedule_free needs the z arena");
  if (d.max_norm > 0.f) {
    NEOSR_CHECK(d.norm_ws, "adan_sf_step: clipping needs norm_ws");
    if (int rc = neosr_grad_norm(d.grad, d.n, d.grad_scale, d.norm_ws, stream)) return rc;
  }
  // scalar coefficients in double, as the reference's Python does (adan_sf.py:183-185,309-322)
  const double b1 = d.beta1, b2 = d.beta2, b3 = d.beta3, lr = d.lr;
  const double bc1 = 1.0 - pow(b1, (double)d.step), bc2 = 1.0 - pow(b2, (double)d.step);
  const double bc3 = 1.0 - pow(b3, (double)d.step);
  AdanArgs a;
  a.d = d;
  a.d.first_step = d.step == 1 || d.first_step;
  a.decay = (float)(1.0 - lr * (double)d.weight_decay);
  a.bc3_sqrt = (float)sqrt(bc3);
  if (d.schedule_free) {
    a.step_size = (float)(lr * (bc1 * (1.0 - (double)d.ckp1)));
    a.step_size_diff = (float)(lr * (b2 / bc2 * (1.0 - (double)d.ckp1)));
  } else {
    a.step_size = (float)(lr / bc1);
    a.step_size_diff = (float)(lr * b2 / bc2);
  }
  hipLaunchKernelGGL(adan_sf_kernel, dim3(grid_for(d.n)), dim3(256), 0, (hipStream_t)stream, a);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_lerp(float* p, const float* end, int64_t n, float weight, void* stream) {
  NEOSR_CHECK(p && end && n > 0, "lerp: bad args");
  hipLaunchKernelGGL(lerp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, end, n, weight);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
