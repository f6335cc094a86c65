// gelu.h — GELU / GELU' as every kernel of the library evaluates them (the CAB elementwise pass of cab.hip and, fused, the
// F(4x4,3x3) convolution epilogue of conv_wino4.hip: the same expressions, so the fused and the separate forms agree bit for
// bit).  erf: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7, from one v_rcp, one v_exp and five FMAs; the Gaussian
// exp(-z^2 / 2) is shared with GELU'.  (csrc/gemm_mfma.hip carries its own copy of the same formula for the Linear epilogues.)
#pragma once

namespace {
__device__ __forceinline__ float gauss_half(float z) { return __expf(-0.5f * z * z); }
__device__ __forceinline__ float erf_from_gauss(float z, float e) {  // erf(z / sqrt 2), e = exp(-z^2 / 2)
  const float x = fabsf(z) * 0.70710678118654752f;
  const float t = __frcp_rn(fmaf(0.3275911f, x, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return copysignf(fmaf(-p * t, e, 1.f), z);
}
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erf_from_gauss(z, gauss_half(z))); }
__device__ __forceinline__ float gelu_d(float z) {
  const float e = gauss_half(z);
  return 0.5f * (1.f + erf_from_gauss(z, e)) + z * 0.3989422804014327f * e;
}
}  // namespace
