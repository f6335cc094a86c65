// conv_wino4.h — geometry shared by the two F(4x4,3x3) kernels: conv3x3_wino4_kernel (conv_wino4.hip, one layer per
// launch) and conv3x3_wino4_chain_kernel (conv_wino4_chain.hip, a chain of layers per launch).  Raw-tile LDS image,
// accumulator-exchange strides, compile-time per-thread tables.  Internal to libneosr_amd.
#pragma once
#include "conv_common.h"
#include "conv_pack.h"

namespace {

constexpr int QT = 16;                 // output pixels per workgroup side
constexpr int QR = QT + 2;             // raw tile side 18
constexpr int QSLOTS = 360;            // pixel slots of a raw buffer: 180 even (rows 0-7, 16, 17) + 180 odd (rows 8-15; 36 unused)
constexpr int QGRAN = QSLOTS * 8;      // 16-byte granules per 32-channel chunk = 2880 = 45 wave-level DMA instructions
constexpr int QES = 36;                // tile stride (floats) of the accumulator exchange image
constexpr int QEXF = 6 * 4 * 16 * QES; // one k-parity half of the exchange image: [row 6][b 4][tile 16][cout 32 (+4)] = 13824 floats
constexpr int QBUF = QEXF;             // floats per LDS object (raw buffer needs 360 * 32 = 11520)
static_assert(QSLOTS * 32 <= QBUF, "raw buffer must fit its LDS object");
constexpr int QU_CHUNK = neosr_pack::WINO4_IMG_FLOATS;  // 36 pos x 2 kp x 2 cout blocks x 256 floats = 144 KB per 32 channels

typedef __attribute__((address_space(3))) void* lds_void;

__device__ __forceinline__ f32x4 ld4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 splat(float v) { return (f32x4){v, v, v, v}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }

// ---- the `fast_matmul` tier (neosr_set_fast_matmul; reference: train.py:168-173 turns TF32 on): every fp32 operand of the
// F(4x4,3x3) products as TWO bf16 pieces hi + lo (16 significant bits instead of TF32's 11), all four cross products on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  The K = 32 slots of one MFMA hold {4 channels x (w_hi, w_lo)} of the
// lane's k quad on the weight side — exactly the 16 bytes the lane loads anyway, so the weight image keeps its size and
// the kernels their loads (bytes per MAC is what bounds this loop: profiles/NEGATIVE_RESULTS.md 5.1) — against
// {a_hi, a_hi} in the first MFMA and {a_lo, a_lo} in the second: 2 bf16 MFMAs of 16 cycles instead of 4 f32 MFMAs of 32.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_bits __attribute__((ext_vector_type(4)));

// hi = the upper 16 bits of the fp32 pattern (truncation: hi is a bf16 number, x - hi is exact), lo = bf16_rne(x - hi)
__device__ __forceinline__ void split_hi_lo(const f32x4 v, bf16x8& hi2, bf16x8& lo2) {
  const unsigned b0 = __float_as_uint(v[0]), b1 = __float_as_uint(v[1]), b2 = __float_as_uint(v[2]), b3 = __float_as_uint(v[3]);
  const unsigned h01 = __builtin_amdgcn_perm(b1, b0, 0x07060302u), h23 = __builtin_amdgcn_perm(b3, b2, 0x07060302u);
  const float r0 = v[0] - __uint_as_float(b0 & 0xffff0000u), r1 = v[1] - __uint_as_float(b1 & 0xffff0000u);
  const float r2 = v[2] - __uint_as_float(b2 & 0xffff0000u), r3 = v[3] - __uint_as_float(b3 & 0xffff0000u);
  // (C conversions, not inline assembly: behind an `asm` hipcc pads no wait states between the vector write and an MFMA that
  // reads the register — see split3 in gemm_mfma.hip)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const bf16x2_t t01 = {(__bf16)r0, (__bf16)r1}, t23 = {(__bf16)r2, (__bf16)r3};
  const unsigned l01 = __builtin_bit_cast(unsigned, t01), l23 = __builtin_bit_cast(unsigned, t23);
  hi2 = __builtin_bit_cast(bf16x8, (u32x4_bits){h01, h23, h01, h23});
  lo2 = __builtin_bit_cast(bf16x8, (u32x4_bits){l01, l23, l01, l23});
}

// Raw tile image: pixel (y, x) of the 18 x 18 tile, channel quad q (0..7) of the chunk lives at float offset
//   32 * slot(y, x) + 4 * (q ^ swz(y, x)),   slot = 2 * (yy * 18 + x) + odd,  odd = 1 for rows 8..15 (yy = y - 8),
//   else 0 with yy = y (rows 0..7) or y - 8 (rows 16, 17);  swz = ((x >> 2) & 3) | (((y >> 2) & 1) << 2).
// A ds_read_b128 lane group holds the 16 tiles once each (k quad fixed per tile row); their pixels (4 ty + r, 4 tx + c)
// differ in (slot parity, (y >> 2) & 1, (x >> 2) & 3) = 16 distinct 16-byte bank groups for every patch position.
constexpr __host__ __device__ int q_slot(int y, int x) {
  const int odd = (y >= 8 && y < 16) ? 1 : 0;
  const int yy = y < 8 ? y : y - 8;
  return 2 * (yy * QR + x) + odd;
}
constexpr __host__ __device__ int q_swz(int y, int x) { return ((x >> 2) & 3) | (((y >> 2) & 1) << 2); }
constexpr __host__ __device__ int q_off(int y, int x, int q) { return 32 * q_slot(y, x) + 4 * (q ^ q_swz(y, x)); }
// rows of the 6 x 6 patch that row ti of B^T d reads (see the column pass in the kernel); k = 2 is unused by rows 0 / 5
constexpr __host__ __device__ int q_row(int ti, int k) {
  return k == 0 ? (ti == 0 ? 0 : 1) : k == 1 ? (ti == 5 ? 3 : 2) : k == 2 ? 3 : (ti == 5 ? 5 : 4);
}

// Launch-invariant per-thread geometry, evaluated at COMPILE time (with twelve waves per CU the address set-up of the
// prologue was issue-bound: ~450 vector instructions per wave before the first load could leave):
//   gran[tid][r]   DMA granule G = r * 768 + tid -> y | x << 8 | channel quad << 16 | (slot exists) << 24
//   pa[wave][lane] LDS float offsets of (patch row k, column block cs) of the lane's tile, k-quad and the wave's parity
struct QTables {
  int gran[768][4];
  int pa[12][64][8];
};
constexpr QTables q_make_tables() {
  QTables t{};
  for (int tid = 0; tid < 768; ++tid)
    for (int r = 0; r < 4; ++r) {
      const int G = r * 768 + tid;
      const int P = G >> 3, sl = G & 7;
      const int odd = P & 1, idx = P >> 1;
      const int yy = idx / QR, x = idx - yy * QR;
      const int y = odd ? yy + 8 : (yy < 8 ? yy : yy + 8);
      const bool used = G < QGRAN && (odd ? yy < 8 : yy < 10);
      const int q = sl ^ q_swz(y, x);
      t.gran[tid][r] = used ? (y | (x << 8) | (q << 16) | (1 << 24)) : 0;
    }
  for (int w = 0; w < 12; ++w)
    for (int l = 0; l < 64; ++l) {
      const int ti = w >> 1, kp = w & 1, t16 = l & 15, kq = l >> 4;
      for (int k = 0; k < 4; ++k)
        for (int cs = 0; cs < 2; ++cs)
          t.pa[w][l][k * 2 + cs] = q_off(4 * (t16 >> 2) + q_row(ti, k), 4 * (t16 & 3) + 4 * cs, 4 * kp + kq);
    }
  return t;
}
static __device__ const QTables g_qt = q_make_tables();

}  // namespace
