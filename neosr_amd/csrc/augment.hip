// augment.hip — batch augmentations of neosr/data/augmentations.py for gfx950:
// the antialiased bilinear / bicubic resizes that bracket every call of apply_augment
// (augmentations.py:258-308: LQ is up-sampled x scale before and down-sampled after, even for "none"),
// resizemix's antialiased bicubic resize into a box, and the box / batch-permutation blends of mixup,
// cutmix and cutblur.  Planar NCHW fp32 like the reference's batch tensors; HBM-bound.
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t n, int cap = 8192) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ATen's antialias filters (aten/src/ATen/native/UpSample.h: aa_filter for bilinear, bicubic a = -0.5)
__device__ __forceinline__ float aa_filter(float x, int mode) {
  x = fabsf(x);
  if (mode == NEOSR_RESIZE_BILINEAR) return x < 1.f ? 1.f - x : 0.f;
  const float a = -0.5f;
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}

// one output sample along one axis: sum_j w_j in[(xmin + j) * stride] with ATen's window
// (_compute_indices_min_size_weights_aa): support = interp/2 * max(scale, 1), weights normalised
__device__ __forceinline__ float aa_sample(const float* in, int64_t stride, int in_size, int o, float scale,
                                           int mode) {
  const float interp = mode == NEOSR_RESIZE_BILINEAR ? 2.f : 4.f;
  const float support = scale >= 1.f ? interp * 0.5f * scale : interp * 0.5f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * (o + 0.5f);
  int xmin = (int)(center - support + 0.5f);
  if (xmin < 0) xmin = 0;
  int xend = (int)(center + support + 0.5f);
  if (xend > in_size) xend = in_size;
  const int xsize = xend - xmin;
  float total = 0.f, acc = 0.f;
  for (int j = 0; j < xsize; ++j) {
    const float w = aa_filter((j + xmin - center + 0.5f) * invscale, mode);
    total += w;
    acc += w * in[(int64_t)(xmin + j) * stride];
  }
  return total != 0.f ? acc / total : acc;
}

// horizontal pass: tmp[b, c, y, xo] from in[perm[b], c, y, :]
__global__ __launch_bounds__(256) void resize_aa_h_kernel(const float* __restrict__ in, float* __restrict__ tmp,
                                                          const int32_t* __restrict__ perm, int B, int C, int Hin,
                                                          int Win, int Wout, float scale_w, int mode) {
  const int64_t total = (int64_t)B * C * Hin * Wout;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int xo = (int)(e % Wout);
    int64_t t = e / Wout;
    const int y = (int)(t % Hin);
    t /= Hin;
    const int c = (int)(t % C), b = (int)(t / C);
    const int sb = perm ? perm[b] : b;
    tmp[e] = aa_sample(in + (((int64_t)sb * C + c) * Hin + y) * Win, 1, Win, xo, scale_w, mode);
  }
}

// vertical pass into the box (y0, x0) of an (Hfull, Wfull) image, optional clamp to [0, 1]
__global__ __launch_bounds__(256) void resize_aa_v_kernel(const float* __restrict__ tmp, float* __restrict__ out,
                                                          int B, int C, int Hin, int Hout, int Wout, int Hfull,
                                                          int Wfull, int y0, int x0, float scale_h, int mode,
                                                          int clamp01) {
  const int64_t total = (int64_t)B * C * Hout * Wout;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int xo = (int)(e % Wout);
    int64_t t = e / Wout;
    const int yo = (int)(t % Hout);
    const int64_t p = t / Hout;  // b * C + c
    float v = aa_sample(tmp + (p * Hin) * Wout + xo, Wout, Hin, yo, scale_h, mode);
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    out[(p * Hfull + y0 + yo) * Wfull + x0 + xo] = v;
  }
}

// out = inside the box ? lam * x[b] + (1 - lam) * src[perm[b]] : x[b]
__global__ __launch_bounds__(256) void box_blend_kernel(const float* __restrict__ x, const float* __restrict__ src,
                                                        const int32_t* __restrict__ perm, float* __restrict__ out,
                                                        int B, int C, int H, int W, int y0, int y1, int x0, int x1,
                                                        float lam) {
  const int64_t total = (int64_t)B * C * H * W, plane = (int64_t)H * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int xx = (int)(e % W), yy = (int)((e / W) % H);
    float v = x[e];
    if (yy >= y0 && yy < y1 && xx >= x0 && xx < x1) {
      const int64_t bc = e / plane;
      const int b = (int)(bc / C), c = (int)(bc % C);
      const int sb = perm ? perm[b] : b;
      const float s = src[((int64_t)sb * C + c) * plane + (int64_t)yy * W + xx];
      v = lam * v + (1.f - lam) * s;
    }
    out[e] = v;
  }
}

}  // namespace

extern "C" int neosr_resize_aa(const float* in, float* out, float* tmp, const int32_t* perm, int32_t B, int32_t C,
                               int32_t Hin, int32_t Win, int32_t Hout, int32_t Wout, int32_t Hfull, int32_t Wfull,
                               int32_t y0, int32_t x0, int32_t mode, int32_t clamp01, void* stream) {
  NEOSR_CHECK(in && out && tmp && B > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0,
              "resize_aa: bad args");
  NEOSR_CHECK(mode == NEOSR_RESIZE_BILINEAR || mode == NEOSR_RESIZE_BICUBIC, "resize_aa: bilinear or bicubic only");
  NEOSR_CHECK(y0 >= 0 && x0 >= 0 && y0 + Hout <= Hfull && x0 + Wout <= Wfull, "resize_aa: box outside the output");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(resize_aa_h_kernel, dim3(grid_for((int64_t)B * C * Hin * Wout)), dim3(256), 0, st, in, tmp, perm,
                     B, C, Hin, Win, Wout, (float)Win / (float)Wout, mode);
  hipLaunchKernelGGL(resize_aa_v_kernel, dim3(grid_for((int64_t)B * C * Hout * Wout)), dim3(256), 0, st, tmp, out, B,
                     C, Hin, Hout, Wout, Hfull, Wfull, y0, x0, (float)Hin / (float)Hout, mode, clamp01);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_box_blend(const float* x, const float* src, const int32_t* perm, float* out, int32_t B,
                               int32_t C, int32_t H, int32_t W, int32_t y0, int32_t y1, int32_t x0, int32_t x1,
                               float lam, void* stream) {
  NEOSR_CHECK(x && src && out && B > 0 && C > 0 && H > 0 && W > 0, "box_blend: bad args");
  hipLaunchKernelGGL(box_blend_kernel, dim3(grid_for((int64_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream, x,
                     src, perm, out, B, C, H, W, y0, y1, x0, x1, lam);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
