// bf16x3.h — an fp32 value as three bf16 pieces and the six-term product on the bf16 MFMA (shared by gemm_mfma.hip and the
// direct-form convolution weight gradient of wgrad.hip).  x = p0 + p1 + p2 with p0 = the upper 16 bits of the pattern, p1 =
// the upper 16 bits of the exact remainder x - p0 and p2 = bf16_rne(x - p0 - p1): 8 + 8 + 8 significant bits.  A product
// from the six leading cross terms (fp32 accumulate) differs from the fp32 product by ~2^-24 of it (tools/micro/split_err.py).
// Internal to libneosr_amd.
#pragma once
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the 8 reduction values of a lane (two 16-byte LDS quads) -> the three 8 x bf16 MFMA operands
__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, bf16x8 (&P)[3]) {
  unsigned b[8];
  float r1[8], r2[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b[e] = __float_as_uint(lo4[e]);
    b[4 + e] = __float_as_uint(hi4[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    r1[e] = __uint_as_float(b[e]) - __uint_as_float(b[e] & 0xffff0000u);
    r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
  }
  u32x4 p0, p1, p2;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p0[i] = __builtin_amdgcn_perm(b[2 * i + 1], b[2 * i], 0x07060302u);
    p1[i] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * i + 1]), __float_as_uint(r1[2 * i]), 0x07060302u);
    // (a C conversion, not inline assembly: hipcc emits v_cvt_pk_bf16_f32 for it AND knows that a vector instruction wrote the
    // register — behind an `asm` it pads no wait states, and an MFMA issued right after may read the operand's OLD contents.
    // Seen in a bf16x3 form of the register-fed TN kernel: the last token pair of the last column came out wrong by ~1e-4;
    // that form was measured at the fp32 kernel's speed once correct — 79.9 vs 79.7 us for qkv at M = 32 768, every value
    // feeds one MFMA there, so the split is not amortised — and is not kept.)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t t = {(__bf16)r2[2 * i], (__bf16)r2[2 * i + 1]};
    p2[i] = __builtin_bit_cast(unsigned, t);
  }
  P[0] = __builtin_bit_cast(bf16x8, p0);
  P[1] = __builtin_bit_cast(bf16x8, p1);
  P[2] = __builtin_bit_cast(bf16x8, p2);
}
// the same for 8 values that live in 8 separate registers
__device__ __forceinline__ void split3v(const float (&x)[8], bf16x8 (&P)[3]) {
  split3((f32x4){x[0], x[1], x[2], x[3]}, (f32x4){x[4], x[5], x[6], x[7]}, P);
}
// acc += sum over the six leading cross terms of (weight pieces W) x (activation pieces X), smallest terms first.
// `fast` (wave-uniform; the `fast_matmul` tier, neosr_set_fast_matmul): only p0q0 + p0q1 + p1q0 — the three terms of
// relative size 2^-16 are dropped, a product is then good to ~2e-5 instead of 2^-24 (train.py:168-173 ships a far looser
// mode behind the same option); never the default.
__device__ __forceinline__ f32x16 mac6(const bf16x8 (&W)[3], const bf16x8 (&X)[3], f32x16 acc, bool fast = false) {
  if (!fast) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[2], X[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], X[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1], X[1], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[1], X[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], X[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[0], X[0], acc, 0, 0, 0);
  return acc;
}

