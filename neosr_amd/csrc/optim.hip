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

// ------------------------------------------------------------------------------------------------
// The other optimizers of base.get_optimizer (neosr/models/base.py:151-172) as ONE elementwise kernel:
// torch.optim.Adam / NAdam, adan (neosr/optimizers/adan.py), adamw_sf (adamw_sf.py), adamw_win
// (adamw_win.py).  All scalar coefficients are computed by the host in double (as the Python originals
// do) and passed in c[]; the model-level clip and the EMA update are fused exactly as in adamw / adan_sf.
namespace {

__global__ __launch_bounds__(256) void optim_kernel(const neosr_optim_desc d) {
  float clip = d.grad_scale;
  if (d.max_norm > 0.f) clip *= fminf(d.max_norm / (d.norm_ws[0] + 1e-6f), 1.f);
  const float* c = d.c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    float p = d.param[i];
    float g = d.grad[i] * clip;
    switch (d.kind) {
      case NEOSR_OPT_ADAM:     // c: b1, b2, eps, wd, lr/bc1, sqrt(bc2)
      case NEOSR_OPT_NADAM: {  // c: b1, b2, eps, wd, cg, sqrt(bc2), cm
        g += c[3] * p;
        float m = d.s0[i], v = d.s1[i];
        m = m + (g - m) * (1.f - c[0]);
        v = v * c[1] + (1.f - c[1]) * g * g;
        const float denom = sqrtf(v) / c[5] + c[2];
        if (d.kind == NEOSR_OPT_ADAM) {
          p -= c[4] * (m / denom);
        } else {
          p -= c[4] * (g / denom);
          p -= c[6] * (m / denom);
        }
        d.s0[i] = m;
        d.s1[i] = v;
        break;
      }
      case NEOSR_OPT_ADAN: {  // s0 m, s1 n, s2 diff, s3 neg_pre_grad; c: b1,b2,b3,eps, lr*wd, ss, ssd, sqrt(bc3), no_prox
        float m = d.s0[i], n = d.s1[i], df = d.s2[i], npg = d.s3[i];
        if (d.flags & 1) npg = -g;  // first step
        npg += g;
        m = m * c[0] + g * (1.f - c[0]);
        df = df * c[1] + npg * (1.f - c[1]);
        npg = npg * c[1] + g;
        n = n * c[2] + (1.f - c[2]) * npg * npg;
        const float denom = sqrtf(n) / c[7] + c[3];
        if (c[8] != 0.f) {
          p *= 1.f - c[4];
          p -= c[5] * (m / denom);
          p -= c[6] * (df / denom);
        } else {
          p -= c[5] * (m / denom);
          p -= c[6] * (df / denom);
          p /= 1.f + c[4];
        }
        d.s0[i] = m;
        d.s1[i] = n;
        d.s2[i] = df;
        d.s3[i] = -g;
        break;
      }
      case NEOSR_OPT_ADAMW_SF: {  // s0 exp_avg_sq, s1 z; c: b2, eps, decay, ckp1, lr_t, lr_t*(b1*(1-ckp1)-1)
        float v = d.s0[i], z = d.s1[i];
        v = v * c[0] + (1.f - c[0]) * g * g;
        float gn = g / (sqrtf(v) + c[1]);
        if (c[2] != 0.f) gn += c[2] * p;
        p = lerp_aten(p, z, c[3]);
        p += c[5] * gn;
        z -= c[4] * gn;
        d.s0[i] = v;
        d.s1[i] = z;
        break;
      }
      default: {  // NEOSR_OPT_ADAMW_WIN: s0 m, s1 v, s2 x, s3 y; c: b1,b2,eps,wd,lr,bc1,sqrt(bc2),beta3,beta4; flags: 0 none 1 win 2 win2
        float m = d.s0[i], v = d.s1[i];
        m = m * c[0] + g * (1.f - c[0]);
        v = v * c[1] + (1.f - c[1]) * g * g;
        const float denom = sqrtf(v) / c[6] + c[2];
        if (d.flags == 0) {
          p *= 1.f - c[4] * c[3];
          p -= (c[4] / c[5]) * (m / denom);
        } else {
          const float upd = (m / denom) / c[5];
          const float lr_x = c[4], lr_y = c[7] * c[4];
          float x = d.s2[i];
          x = (x - lr_x * upd) * (1.f / (1.f + lr_x * c[3]));
          float gamma = 1.f / (1.f + lr_y / lr_x + lr_y * c[3]);
          if (d.flags == 1) {
            p = p * gamma + (lr_y / lr_x) * gamma * x - lr_y * gamma * upd;
          } else {
            float y = d.s3[i];
            y = y * gamma + (lr_y / lr_x) * gamma * x - lr_y * gamma * upd;
            const float lr_z = c[8] * c[4];
            gamma = 1.f / (1.f + lr_z / lr_x + lr_z / lr_y + lr_z * c[3]);
            p = p * gamma - lr_z * gamma * upd;
            p = p + (lr_z / lr_x) * gamma * x + (lr_z / lr_y) * gamma * y;
            d.s3[i] = y;
          }
          d.s2[i] = x;
        }
        d.s0[i] = m;
        d.s1[i] = v;
      }
    }
    d.param[i] = p;
    if (d.ema) {
      const float e = d.ema[i];
      d.ema[i] = d.ema_decay < 0.f ? p : e + (p - e) * (1.f - d.ema_decay);
    }
  }
}

}  // namespace

extern "C" int neosr_optim_step(const neosr_optim_desc* dp, void* stream) {
  NEOSR_CHECK(dp, "optim_step: null descriptor");
  const neosr_optim_desc& d = *dp;
  NEOSR_CHECK(d.param && d.grad && d.s0 && d.s1 && d.n > 0, "optim_step: bad args");
  NEOSR_CHECK(d.kind >= NEOSR_OPT_ADAM && d.kind <= NEOSR_OPT_ADAMW_WIN, "optim_step: unknown kind %d", d.kind);
  NEOSR_CHECK(d.kind != NEOSR_OPT_ADAN || (d.s2 && d.s3), "optim_step: adan needs 4 state arenas");
  NEOSR_CHECK(d.kind != NEOSR_OPT_ADAMW_WIN || d.flags == 0 || (d.s2 && (d.flags == 1 || d.s3)),
              "optim_step: adamw_win needs the x (and y) arenas");
  if (d.max_norm > 0.f) {
    NEOSR_CHECK(d.norm_ws, "optim_step: clipping needs norm_ws");
    if (int rc = neosr_grad_norm(d.grad, d.n, d.grad_scale, d.norm_ws, stream)) return rc;
  }
  hipLaunchKernelGGL(optim_kernel, dim3(grid_for(d.n)), dim3(256), 0, (hipStream_t)stream, d);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------- Friendly SAM
namespace {

constexpr int FSAM_BLOCKS = 1024;

// pass 1: g' = g - sigma * m (not on the first call), m <- lmbda * m + (1 - lmbda) * g, g' written over
// the gradient; per-workgroup partial of sum (|w| g')^2 (adaptive) in a fixed order
__global__ __launch_bounds__(256) void fsam_momentum_kernel(const neosr_fsam_desc d) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    const float g = d.grad[i] * d.grad_scale;
    float gp = g, m = g;
    if (!d.first) {
      const float m0 = d.momentum[i];
      gp = g - m0 * d.sigma;
      m = m0 * d.lmbda + g * (1.f - d.lmbda);
    }
    d.momentum[i] = m;
    d.grad[i] = gp;
    const float t = d.adaptive ? fabsf(d.param[i]) * gp : gp;
    acc += t * t;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) d.norm_ws[4 + blockIdx.x] = red[0];
}

// pass 2: every workgroup re-sums the partials in the same order (identical scale everywhere), keeps w
// in old_p and climbs to w + rho * w^2 * g' / (norm + 1e-12)
__global__ __launch_bounds__(256) void fsam_perturb_kernel(const neosr_fsam_desc d, int nparts) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += d.norm_ws[4 + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float norm = sqrtf(red[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) d.norm_ws[0] = norm;
  const float scale = d.rho / (norm + 1e-12f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (int64_t)gridDim.x * 256) {
    const float w = d.param[i];
    d.old_p[i] = w;
    d.param[i] = w + (d.adaptive ? w * w : 1.f) * d.grad[i] * scale;
  }
}

}  // namespace

extern "C" int neosr_fsam_first_step(const neosr_fsam_desc* dp, void* stream) {
  NEOSR_CHECK(dp, "fsam_first_step: null descriptor");
  const neosr_fsam_desc& d = *dp;
  NEOSR_CHECK(d.param && d.grad && d.momentum && d.old_p && d.norm_ws && d.n > 0, "fsam_first_step: bad args");
  NEOSR_CHECK(d.rho >= 0.f, "fsam_first_step: rho must be non-negative");
  int nb = (int)((d.n + 255) / 256);
  if (nb > FSAM_BLOCKS) nb = FSAM_BLOCKS;
  hipLaunchKernelGGL(fsam_momentum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d);
  hipLaunchKernelGGL(fsam_perturb_kernel, dim3(grid_for(d.n)), dim3(256), 0, (hipStream_t)stream, d, nb);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
