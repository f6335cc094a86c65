// gemm_mfma.hip — fp32 GEMMs of the transformer generators (SwinIR / HAT `nn.Linear` layers) on
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate) for gfx950.
//
//   NT  C[M,N] = A[M,K] * B[N,K]^T          Linear forward   (B = weight (out,in))
//   NN  C[M,N] = A[M,K] * B[K,N]            Linear backward-data (A = dY, B = weight)
//   TN  C[M,N] = A[K,M]^T * B[K,N]          Linear backward-weight (A = dY, B = X), split over K
// 128x64 output tile per 256-thread workgroup (each wave 32 rows x 64 cols = 2 MFMA tiles), K in
// chunks of 32 staged through LDS ([row][32+1] layouts -> conflict-free fragment reads) with the
// next chunk's global loads in flight in registers.  D is formed as B_frag x A_frag so each lane
// owns one output row and runs of 4 consecutive columns -> 16-byte epilogue stores.
// Epilogue (NT/NN): + bias[n]; erf-form GELU (A&S 7.1.26 erf, |err| <= 1.5e-7) with the pre-activation kept in `aux_out`;
// multiply by GELU'(aux_in) (backward through the activation); per-sample row scale (DropPath);
// + residual.  TN writes split-K partial slabs that `colsum_kernel` sums in a fixed order.
//
//
// Default since round 5 (neosr_set_gemm_x3): the same three products from bf16x3 pieces — every fp32 operand as
// p0 + p1 + p2 (8 + 8 + 8 significant bits), the six leading cross terms on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: as accurate as the fp32 MFMA (split3 / mac6 below), 6 x 32 cycles per 16 reduction indices instead of
// 8 x 64.  NT / NN: gemm_nt_glds_x3_kernel & co (the fp32 kernels' tiles and epilogues, weight tile pre-split into LDS
// planes); TN: gemm_tn_lds_x3_kernel / _group_kernel (192 x 192 tiles, both operands split once at staging).  Under
// neosr_set_fast_matmul the three 2^-16 terms are dropped (the labelled reduced-precision tier).  The fp32 kernels stay
// selectable (neosr_set_gemm_x3(0)) and take the shapes the bf16x3 forms do not.
//
// Reference call sites: neosr/archs/swinir_arch.py:15-38 (Mlp), :139-143,150-156,209-210
// (qkv / proj Linears), and the same layers of neosr/archs/hat_arch.py.
#include <cstring>
#include <type_traits>

#include <atomic>
#include "common.h"
#include "conv_common.h"
#include "../../include/neosr_amd.h"
#include "prof.h"
#include "bf16x3.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32, LDK = BK + 1;
// operands that are contiguous along the reduction index (NT: A and B; NN: A) are staged [row][k]
// with the odd stride LDK; operands contiguous along the output index (NN: B; TN: A and B) are
// staged [k][row] with 16-byte row stores (strides LDM / LDN) — the MFMA fragment reads then walk
// consecutive addresses per lane, so neither layout needs a transposing scatter into LDS
constexpr int LDM = BM + 4, LDN = BN + 4;
constexpr int A_LDS = BK * LDM, B_LDS = BK * LDN;
static_assert(A_LDS >= BM * LDK && B_LDS >= BN * LDK, "LDS carve-up");

__device__ __attribute__((aligned(256))) float gm_zero_page[64];
__device__ __attribute__((aligned(256))) float gm_trash[1024];

// erf for the GELU epilogues: Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 (the fp32 resolution of erf near 1), from one
// v_rcp, one v_exp and five FMAs — the library erff costs ~3x the VALU work, and the fc1 epilogue (32 values per lane
// behind a 6-chunk K loop) is VALU-bound: 72.8 us with erff against 53.8 us for the bare product at M = 32768.
// The Gaussian exp(-z^2/2) it needs is shared with GELU'.
__device__ __forceinline__ float gauss_half(float z) { return __expf(-0.5f * z * z); }  // exp(-z^2 / 2)
__device__ __forceinline__ float erf_from_gauss(float z, float e) {  // erf(z / sqrt 2), e = exp(-z^2 / 2)
  const float x = fabsf(z) * 0.70710678118654752f;
  const float t = __frcp_rn(fmaf(0.3275911f, x, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  return copysignf(fmaf(-p * t, e, 1.f), z);
}
__device__ __forceinline__ float gelu_exact(float z) { return 0.5f * z * (1.f + erf_from_gauss(z, gauss_half(z))); }
__device__ __forceinline__ float gelu_grad(float z) {
  const float e = gauss_half(z);
  return 0.5f * (1.f + erf_from_gauss(z, e)) + z * 0.3989422804014327f * e;
}

struct GemmArgs {
  neosr_gemm_desc d;
  int ksplit_len;  // TN: K range per split
  int tiles_m, tiles_n, nsplit;
  float* colsum_part;  // TN with d.colsum_a: per-split partial column sums of A, row stride slab
  int64_t slab;        // TN: floats per split-K slab (M*N, + M when the column sums ride behind it)
  int fast3;       // bf16x3 kernels: the three-product fast_matmul tier (mac6)
  int b_vec;       // B (and bias) 16-byte aligned -> float4 loads; else dword loads (weights that sit
                   // at a 4-byte-aligned offset of a packed parameter arena)
};

__device__ __forceinline__ f32x4 ld4(const float* p, int vec) {
  if (vec) return *reinterpret_cast<const f32x4*>(p);
  f32x4 r = {p[0], p[1], p[2], p[3]};
  return r;
}

// ---- bf16x3 operands (round 5).  An fp32 value as three bf16 pieces, x = p0 + p1 + p2 with p0 = the upper 16 bits of the
// pattern, p1 = the upper 16 bits of the exact remainder x - p0 and p2 = bf16_rne(x - p0 - p1): 8 + 8 + 8 significant
// bits.  A product of two such values from the six leading cross terms p0q0, p0q1, p1q0, p0q2, p1q1, p2q0 on the bf16 MFMA
// (fp32 accumulate) differs from the fp32 product by ~2^-24 of it — the dropped terms are 2^-24 and below — so a GEMM built
// from them is as accurate as the fp32 MFMA one (tools/micro/split_err.py: 1.5e-6 vs 2.8e-6 rel. L2 on a K = 192 Winograd
// product against float64) while v_mfma_f32_32x32x16_bf16 retires 16 K-values per 32 cycles against 2 per 64 for
// v_mfma_f32_32x32x2_f32: 6 x 32 cycles per 16 reduction indices instead of 8 x 64.

// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KB at `lds_addr`) as inline assembly.  hipcc waits `vmcnt(0)` in front of
// any ds_read that MAY alias the target of an LDS-DMA it knows about — with both chunk buffers in one __shared__ array that
// is every fragment read, i.e. the DMA of chunk c + 1 was waited for BEFORE chunk c was computed (the ISA of the round 2-4
// kernels shows the wait right behind the loads).  Hidden from the wait-count pass, the DMA runs under the chunk's MFMAs;
// the explicit `s_waitcnt vmcnt(0)` in front of each chunk barrier is the only wait it needs.  (Counted waits the compiler
// places for its own register loads only become stronger: vmcnt counts the hidden loads too.)
typedef int gm_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gm_dma16(gm_i32x4 rsrc, const float* lds_ptr, int voff, int soff) {
  // (wave-uniform: the wave index inside it comes from the thread id)
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)lds_ptr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory", "m0");
}

// epilogue shared by all GEMM kernels: lane owns C row m = m0 + 32*wave + l31 and the column quads
// 32t + 8g + 4lh + {0..3} (D was formed as Bfrag x Afrag)
struct NoPref {};
// PREF = float4[8]: residual quads already in registers (index 4 t + g), bias tile in LDS (`bias_lds`)
// NTILE 32-column accumulator tiles per wave; row_off / col_off: position of the wave's tile inside the workgroup tile
// (default: wave w owns rows 32 w.., columns 0..)
// APREF = float4[..]: the pre-activation quads of `aux_in` already in registers too (same index)
template <int MODE, typename PREF = NoPref, int NTILE = 2, typename APREF = NoPref>
__device__ __forceinline__ void epilogue(const GemmArgs& args, const f32x16 (&acc)[NTILE], int m0, int n0, int split,
                                         const float* bias_lds = nullptr, const PREF& res_pref = PREF{},
                                         int row_off = -1, int col_off = 0, const APREF& aux_pref = APREF{}) {
  constexpr bool HAS_PREF = !std::is_same<PREF, NoPref>::value;
  constexpr bool HAS_APREF = !std::is_same<APREF, NoPref>::value;
  const neosr_gemm_desc& d = args.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int M = d.M, N = d.N;
  const int m = m0 + (row_off < 0 ? wave * 32 : row_off) + l31;
  const bool m_ok = m < M;
  const int64_t mrow = m_ok ? m : 0;
  float* Cbase = d.C;
  if (MODE == 2) Cbase = d.C + (int64_t)split * args.slab;  // split-K partial slab
  const float rs = (MODE != 2 && d.row_scale && m_ok) ? d.row_scale[m / d.rows_per_scale] : 1.f;
#pragma unroll
  for (int t = 0; t < NTILE; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + col_off + 32 * t + 8 * g + 4 * lh;
      const bool ok = m_ok && n < N;
      float v[4] = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
      if (MODE != 2) {
        const int ns = n < N ? n : 0;
        if (d.bias) {
          const f32x4 b = bias_lds ? *reinterpret_cast<const f32x4*>(bias_lds + col_off + 32 * t + 8 * g + 4 * lh)
                                   : ld4(d.bias + ns, args.b_vec);
          v[0] += b[0]; v[1] += b[1]; v[2] += b[2]; v[3] += b[3];
        }
        if (d.aux_out)  // keep the pre-activation for the backward pass
          *reinterpret_cast<float4*>(ok ? d.aux_out + mrow * d.ldaux + n : gm_trash + tid * 4) =
              make_float4(v[0], v[1], v[2], v[3]);
        if (d.gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
        }
        if (d.aux_in) {  // dz = da * GELU'(z)
          float4 z;
          if constexpr (HAS_APREF)
            z = aux_pref[4 * t + g];
          else
            z = *reinterpret_cast<const float4*>(ok ? d.aux_in + mrow * d.ldaux + n : gm_zero_page);
          v[0] *= gelu_grad(z.x); v[1] *= gelu_grad(z.y); v[2] *= gelu_grad(z.z); v[3] *= gelu_grad(z.w);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= rs;
        if (d.res) {
          float4 r;
          if constexpr (HAS_PREF)
            r = res_pref[4 * t + g];
          else
            r = *reinterpret_cast<const float4*>(ok ? d.res + mrow * d.ldres + n : gm_zero_page);
          v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
      }
      *reinterpret_cast<float4*>(ok ? Cbase + mrow * d.ldc + n : gm_trash + tid * 4) =
          make_float4(v[0], v[1], v[2], v[3]);
    }
}

// MODE 0 = NT, 1 = NN, 2 = TN.  BMV = 64 (NT / NN only): 64-row tiles for launches that leave the last round of 128-row
// tiles mostly empty (M = 16 384 tokens); the four waves then form a 2 x 2 grid of 32 x 32 tiles, one accumulator each.
template <int MODE, int BMV = BM>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(const GemmArgs args) {
  static_assert(BMV == BM || (BMV == 64 && MODE != 2), "64-row tiles: NT / NN");
  constexpr int NT = BMV == BM ? 2 : 1, AI = BMV / 32;
  const neosr_gemm_desc& d = args.d;
  __shared__ float lds[A_LDS + B_LDS];
  float* As = lds;          // [m][k]
  float* Bs = lds + A_LDS;  // [n][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs, each with a private 4 MB L2.
  // NT/NN: every XCD gets a contiguous run of tiles, n-tile fastest, so the workgroups sharing one
  // 128-row A panel run back to back on one L2.  TN: all output tiles of one K-split run on ONE XCD
  // (split s -> XCD s % 8), so that split's rows of dY and X are fetched from HBM once, not once per
  // tile — K = B*H*W is huge and the (M x N) output tiny, so this is the traffic that matters.
  const int tiles = args.tiles_m * args.tiles_n;
  int logical, split = 0;
  if (MODE == 2) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    split = (j / tiles) * 8 + xcd;
    logical = j % tiles;
    if (split >= args.nsplit) return;
  } else {
    const int chunk = gridDim.x >> 3;
    logical = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (logical >= tiles) return;
  }
  const int m0 = (logical / args.tiles_n) * BMV, n0 = (logical % args.tiles_n) * BN;
  const int M = d.M, N = d.N;
  int k_lo = 0, k_hi = d.K;
  if (MODE == 2) {
    k_lo = split * args.ksplit_len;
    k_hi = min(d.K, k_lo + args.ksplit_len);
  }

  // staging registers: A tile 128x32 floats = 4 float4 / thread, B tile 64x32 = 2 float4 / thread
  f32x4 ra[4], rb[2];
  float rsc[4] = {1.f, 1.f, 1.f, 1.f};  // TN: row_scale[k / rows_per_scale] of the staged dY rows (DropPath)
  auto gload = [&](int k0) {
    if (MODE != 2) {
      // A rows m (contiguous along k): thread -> (row = tid/8 + 32 i, k4 = (tid%8)*4)
      const int k4 = (tid & 7) << 2;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
        const bool ok = m < M && k0 + k4 < k_hi;
        ra[i] = *reinterpret_cast<const f32x4*>(ok ? d.A + (int64_t)m * d.lda + k0 + k4 : gm_zero_page);
      }
    } else {
      // A = dY[K rows = samples][M cols]: rows kk (chunk of 32), cols m tile (128): thread ->
      // (kk = tid/32 + 8 i, m4 = (tid%32)*4)
      const int m4 = (tid & 31) << 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = k0 + (tid >> 5) + 8 * i;
        const bool ok = kk < k_hi && m0 + m4 < M;
        ra[i] = *reinterpret_cast<const f32x4*>(ok ? d.A + (int64_t)kk * d.lda + m0 + m4 : gm_zero_page);
        if (d.row_scale) rsc[i] = kk < k_hi ? d.row_scale[kk / d.rows_per_scale] : 0.f;
      }
    }
    if (MODE == 0) {
      // B = W[N rows][K cols]: thread -> (row = tid/8 + 32 i, k4)
      const int k4 = (tid & 7) << 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        const bool ok = n < N && k0 + k4 < k_hi;
        rb[i] = ld4(ok ? d.B + (int64_t)n * d.ldb + k0 + k4 : gm_zero_page, args.b_vec);
      }
    } else {
      // B[K rows][N cols] (contiguous along n): thread -> (kk = tid/16 + 16 i, n4 = (tid%16)*4)
      const int n4 = (tid & 15) << 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = k0 + (tid >> 4) + 16 * i;
        const bool ok = kk < k_hi && n0 + n4 < N;
        rb[i] = ld4(ok ? d.B + (int64_t)kk * d.ldb + n0 + n4 : gm_zero_page, args.b_vec);
      }
    }
  };
  auto sstore = [&]() {
    if (MODE != 2) {
      const int k4 = (tid & 7) << 2;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        float* p = As + ((tid >> 3) + 32 * i) * LDK + k4;
        p[0] = ra[i][0]; p[1] = ra[i][1]; p[2] = ra[i][2]; p[3] = ra[i][3];
      }
    } else {
      const int m4 = (tid & 31) << 2;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(As + ((tid >> 5) + 8 * i) * LDM + m4) = d.row_scale ? ra[i] * rsc[i] : ra[i];
    }
    if (MODE == 0) {
      const int k4 = (tid & 7) << 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float* p = Bs + ((tid >> 3) + 32 * i) * LDK + k4;
        p[0] = rb[i][0]; p[1] = rb[i][1]; p[2] = rb[i][2]; p[3] = rb[i][3];
      }
    } else {
      const int n4 = (tid & 15) << 2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<f32x4*>(Bs + ((tid >> 4) + 16 * i) * LDN + n4) = rb[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int wrow = BMV == BM ? wave * 32 : (wave & 1) * 32, wcol = BMV == BM ? 0 : (wave >> 1) * 32;

  // TN: the n-tile-0 workgroups also sum the columns of their A tile (= the bias gradient sum_rows dY)
  const bool do_colsum = MODE == 2 && args.colsum_part && n0 == 0 && tid < BM;
  float csum = 0.f;
  const int nchunks = (k_hi - k_lo + BK - 1) / BK;
  if (nchunks > 0) gload(k_lo);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (do_colsum) {
#pragma unroll 8
      for (int kk = 0; kk < BK; ++kk) csum += As[kk * LDM + tid];
    }
#ifndef GEMM_NO_GLOAD
    if (c + 1 < nchunks) gload(k_lo + (c + 1) * BK);
#endif
    // element (row r, k) of the staged A / B tile; strides are compile-time per MODE
    constexpr int a_rs = MODE == 2 ? 1 : LDK, a_ks = MODE == 2 ? LDM : 1;
    constexpr int b_rs = MODE == 0 ? LDK : 1, b_ks = MODE == 0 ? 1 : LDN;
    const float* ap = As + (wrow + l31) * a_rs + lh * a_ks;
    const float* bp = Bs + (wcol + l31) * b_rs + lh * b_ks;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      const float a = ap[ks * 2 * a_ks];
      const float b0 = bp[ks * 2 * b_ks];
      // D = Bfrag x Afrag: rows i = column n of C, cols j = row m of C
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a, acc[0], 0, 0, 0);
      if (NT == 2) acc[NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bp[32 * b_rs + ks * 2 * b_ks], a, acc[NT - 1], 0, 0, 0);
    }
  }

  if (do_colsum && m0 + tid < M) args.colsum_part[(int64_t)split * args.slab + m0 + tid] = csum;
  epilogue<MODE, NoPref, NT>(args, acc, m0, n0, split, nullptr, NoPref{}, BMV == BM ? -1 : wrow, wcol);
}

// NT GEMM with direct-to-LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass, two
// LDS buffers and ONE barrier per K chunk; MFMA fragments come from 16-byte LDS reads (one ds_read_b128
// feeds 4 MFMAs).  LDS image per operand: [row][8 quads of 4 k] with quad q of row r stored at slot
// q ^ ((r >> 1) & 7): a wave's glds instruction still fetches full 128-byte rows (8 lanes per row, permuted
// within the line).  ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same + 32: MI355X_MICROARCH.md, LDS) against 64 banks = sixteen 16-byte slots; a 128-byte row covers half a
// bank row, so lane (row r, quad q) sits on slot 8 (r & 1) + (q ^ g(r)).  With g = r & 7 (rounds 1-3) rows r and
// r + 24 of a group met on one slot — a 2-way conflict on EVERY fragment read (SQ_LDS_BANK_CONFLICT = 0.47 of the
// LDS cycles); g = (r >> 1) & 7 maps the eight even and the eight odd rows of either group onto eight
// distinct slots each: conflict-free.
// Fragment convention: lanes with lh = 0 read quad 2s, lanes with lh = 1 quad 2s+1; MFMA e of step s then
// multiplies k = 8s + e (lh 0) and k = 8s + 4 + e (lh 1) — the same pairing for both operands.
// X3: the products on the bf16 MFMA from bf16x3 operands (see split3 / mac6), everything else unchanged.
// BT (X3 only): B is given as [K][N] (the NN form: Linear backward-data, B = weight (out, in) with K = out) — only the
// register staging of the B tile differs: a thread gathers its 8 reduction values of column n with 8 dword loads.
template <bool X3, bool BT = false>
__device__ __forceinline__ void gemm_nt_glds_body(const GemmArgs& args) {
  static_assert(X3 || !BT, "the [K][N] form of B needs the register-staged (X3) B tile");
  const neosr_gemm_desc& d = args.d;
  // X3: the B (weight) tile lives in LDS as three bf16 planes — 12 slots of 16 bytes per row (slot = 4 piece + 2 pair-step
  // + lane half: the 8 reduction indices one lane half feeds to one MFMA), split ONCE per workgroup when the tile is staged
  // through registers, instead of by every wave at every fragment read; slot s of row n sits at s ^ ((n >> 2) & 3)
  // (a ds_read_b128 lane group then covers sixteen distinct 16-byte bank slots).  The A tile stays an fp32 LDS-DMA image:
  // every A element is read by exactly one wave, so splitting it at the fragment read costs the same as at staging.
  constexpr int BSZ = X3 ? BN * 48 : BN * BK;   // floats per B buffer (X3: 64 rows x 192 bytes)
  __shared__ __attribute__((aligned(1024))) float lds[2 * (BM * BK + BSZ)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int tiles = args.tiles_m * args.tiles_n;
  const int chunk = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (logical >= tiles) return;
  const int m0 = (logical / args.tiles_n) * BM, n0 = (logical % args.tiles_n) * BN;
  const int K = d.K;
  const bool fast3 = __builtin_amdgcn_readfirstlane(args.fast3) != 0;
  // per-lane source rows: instruction i of this wave covers tile rows (4 i + wave) * 8 + lane / 8
  const int rsub = lane >> 3, slot = lane & 7;
  const float* zp = gm_zero_page;  // pinned in SGPRs (else its address is re-read through the GOT in every chunk)
  asm volatile("" : "+s"(zp));
  // epilogue operands off the critical path: the 64 bias values of this column tile wait in LDS (one load per lane at
  // the start instead of 8 exposed L2 round trips per lane at the end), the residual quads are requested under the
  // last chunk's MFMAs
  __shared__ __attribute__((aligned(16))) float bias_s[BN];
  // (requested here, stored to LDS behind the first chunk's loads: written `bias_s[tid] = d.bias[..]` in one statement the
  // store waited vmcnt(0) for the load right here — one exposed L2 round trip per workgroup in front of the first DMA;
  // round 6, found in the ISA)
  float bias_v = 0.f;
  if (d.bias && tid < BN) bias_v = n0 + tid < d.N ? d.bias[n0 + tid] : 0.f;
  // DMA addressing (round 4): a lane's source row and k quad never change — only the chunk does — so the byte offsets are
  // computed ONCE (rows past M / N: an out-of-range offset = zeros; a second set for the last chunk drops the quads past
  // K) and a chunk adds its 128-byte step as the SCALAR offset of `buffer_load ... lds`.  Rounds 2-3 rebuilt a 64-bit
  // address with two range checks and a zero-page select for each of the six loads of every chunk: ~70 of the ~160 vector
  // instructions a wave issued per chunk beside its 32 MFMAs (SQ counters: 0.64 vector instructions per MFMA op against
  // 0.34 in the staged NN kernel), on a SIMD where vector and matrix instructions do not overlap.
  const int kc_last = (K + BK - 1) / BK - 1;
  constexpr int OOB = 0x7ffffff0;
  const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, 0x7ffffff0, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, 0x7ffffff0, 0x00020000);
  const gm_i32x4 rA_w = {__builtin_amdgcn_readfirstlane((int)(uintptr_t)d.A),
                         __builtin_amdgcn_readfirstlane((int)(((uintptr_t)d.A >> 32) & 0xffff)), 0x7ffffff0, 0x00020000};
  int offA[4], offA_l[4], offB[2], offB_l[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (4 * i + wave) * 8 + rsub;
    const int kq = 4 * (slot ^ ((r >> 1) & 7));
    const bool ok = m0 + r < d.M;
    offA[i] = ok ? ((m0 + r) * d.lda + kq) * 4 : OOB;
    offA_l[i] = (ok && kc_last * BK + kq < K) ? offA[i] : OOB;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (4 * i + wave) * 8 + rsub;
    const int kq = 4 * (slot ^ ((r >> 1) & 7));
    const bool ok = n0 + r < d.N;
    offB[i] = ok ? ((n0 + r) * d.ldb + kq) * 4 : OOB;
    offB_l[i] = (ok && kc_last * BK + kq < K) ? offB[i] : OOB;
  }
  typedef __attribute__((address_space(3))) void* lds_vp;
  auto issue = [&](int k0, int buf) {
    float* abuf = lds + buf * (BM * BK + BSZ);
    float* bbuf = abuf + BM * BK;
    const bool last = k0 == kc_last * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (X3)
        gm_dma16(rA_w, abuf + (4 * i + wave) * 256, last ? offA_l[i] : offA[i], k0 * 4);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vp)(abuf + (4 * i + wave) * 256), 16, last ? offA_l[i] : offA[i],
                                                 k0 * 4, 0, 0);
    }
    if constexpr (!X3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vp)(bbuf + (4 * i + wave) * 256), 16, last ? offB_l[i] : offB[i],
                                                 k0 * 4, 0, 0);
    }
  };
  // X3: thread (row tid >> 2, octet tid & 3) carries 8 consecutive reduction values of the NEXT B chunk in registers
  const int bn = tid >> 2, bo = tid & 3;
  const bool bn_ok = n0 + bn < d.N;
  const int offB3 = bn_ok ? ((n0 + bn) * d.ldb + 8 * bo) * 4 : OOB;
  const int offB3_l0 = (bn_ok && kc_last * BK + 8 * bo < K) ? offB3 : OOB;
  const int offB3_l1 = (bn_ok && kc_last * BK + 8 * bo + 4 < K) ? offB3 + 16 : OOB;
  f32x4 breg0 = {0.f, 0.f, 0.f, 0.f}, breg1 = {0.f, 0.f, 0.f, 0.f};
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(rB, 0, 0, 0)) rawq_t;
  // BT: rows of B are reduction indices: the resource ends behind row K - 1 (rows past K read zeros), the row is the
  // scalar offset, the column the lane offset
  const auto rBt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, BT ? (unsigned)(((int64_t)K * d.ldb) * 4) : 0u, 0x00020000);
  const int offBt = bn_ok ? (8 * bo * d.ldb + n0 + bn) * 4 : OOB;
  auto bload = [&](int k0) {
    if constexpr (BT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        breg0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBt, offBt, (k0 + e) * d.ldb * 4, 0));
        breg1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBt, offBt, (k0 + 4 + e) * d.ldb * 4, 0));
      }
      return;
    }
    const bool last = k0 == kc_last * BK;
    breg0 = __builtin_bit_cast(f32x4, (rawq_t)__builtin_amdgcn_raw_buffer_load_b128(rB, last ? offB3_l0 : offB3, k0 * 4, 0));
    breg1 = __builtin_bit_cast(f32x4, (rawq_t)__builtin_amdgcn_raw_buffer_load_b128(rB, last ? offB3_l1 : (offB3 == OOB ? OOB : offB3 + 16), k0 * 4, 0));
  };
  auto bstore = [&](int buf) {
    float* bbuf = lds + buf * (BM * BK + BSZ) + BM * BK;
    bf16x8 P[3];
    split3(breg0, breg1, P);
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3)
      *reinterpret_cast<bf16x8*>(bbuf + bn * 48 + 4 * ((4 * p3 + bo) ^ ((bn >> 2) & 3))) = P[p3];
  };
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int nchunks = (K + BK - 1) / BK;
  issue(0, 0);
  if constexpr (X3) {
    bload(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    bstore(0);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  if (d.bias && tid < BN) bias_s[tid] = bias_v;
  __syncthreads();
  const int arow = wave * 32 + l31;
  // K = 180 / 360 end in a chunk with 20 / 8 valid columns: the last chunk runs only the 8-column steps it needs
  // (the DMA zero-fills the rest of a started step); a column tile that lies wholly past N (N = 540: the last
  // 64-wide tile holds 28 columns) skips its MFMAs
  const int last_steps = (K - (nchunks - 1) * BK + 7) >> 3;
  const bool two = n0 + 32 < d.N;
  float4 resq[8], auxq[8];
  auto run = [&](auto two_tag) {
    constexpr bool TWO = decltype(two_tag)::value;
    auto mac = [&](int c, int nsteps) {
      const float* abuf = lds + (c & 1) * (BM * BK + BSZ);
      const float* bbuf = abuf + BM * BK;
      if constexpr (X3) {
        // pair-step ps = the 16 reduction indices 16 ps .. 16 ps + 15; lane half lh owns 8 consecutive ones (two A quads,
        // one 16-byte slot of each B plane)
        const int sa = (arow >> 1) & 7, sb = (l31 >> 2) & 3;
#pragma unroll
        for (int ps = 0; ps < BK / 16; ++ps) {
          if (2 * ps < nsteps) {
            const int q = 4 * ps + 2 * lh;
            bf16x8 X[3], W0[3], W1[3];
#ifdef GEMM_FAKE_PRESPLIT   // timing probe only: three plane reads instead of two fp32 reads + split3; WRONG numbers
            X[0] = *reinterpret_cast<const bf16x8*>(abuf + arow * BK + 4 * (q ^ sa));
            X[1] = *reinterpret_cast<const bf16x8*>(abuf + arow * BK + 4 * ((q + 1) ^ sa));
            X[2] = *reinterpret_cast<const bf16x8*>(abuf + arow * BK + 4 * (((q + 2) & 7) ^ sa));
#else
            split3(*reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * (q ^ sa)),
                   *reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * ((q + 1) ^ sa)), X);
#endif
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) {
              const int sl = 4 * ((4 * p3 + 2 * ps + lh) ^ sb);
              W0[p3] = *reinterpret_cast<const bf16x8*>(bbuf + l31 * 48 + sl);
              if (TWO) W1[p3] = *reinterpret_cast<const bf16x8*>(bbuf + (32 + l31) * 48 + sl);
            }
            acc[0] = mac6(W0, X, acc[0], fast3);
            if (TWO) acc[1] = mac6(W1, X, acc[1], fast3);
          }
        }
        return;
      }
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        if (s < nsteps) {
          const int q = 2 * s + lh;
          const f32x4 a = *reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * (q ^ ((arow >> 1) & 7)));
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bbuf + l31 * BK + 4 * (q ^ ((l31 >> 1) & 7)));
          f32x4 b1 = {0.f, 0.f, 0.f, 0.f};
          if (TWO) b1 = *reinterpret_cast<const f32x4*>(bbuf + (32 + l31) * BK + 4 * (q ^ ((l31 >> 1) & 7)));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[e], a[e], acc[0], 0, 0, 0);
            if (TWO) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[e], a[e], acc[1], 0, 0, 0);
          }
        }
      }
    };
    for (int c = 0; c + 1 < nchunks; ++c) {
      issue((c + 1) * BK, (c + 1) & 1);
      if constexpr (X3) bload((c + 1) * BK);
      mac(c, BK / 8);
      __builtin_amdgcn_s_waitcnt(0x0f70);  // the next chunk has landed ...
      if constexpr (X3) bstore((c + 1) & 1);   // (... its B values in registers: split and written as planes)
      __syncthreads();                      // ... for every wave, and this buffer is free to overwrite
    }
    if (d.res) {
      const int m = m0 + wave * 32 + l31;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 32 * t + 8 * g + 4 * lh;
          resq[4 * t + g] = *reinterpret_cast<const float4*>(m < d.M && n < d.N ? d.res + (int64_t)m * d.ldres + n : zp);
        }
    }
    if (d.aux_in) {   // (the GELU' operand of the data-gradient epilogue, like the residual: requested under the last chunk)
      const int m = m0 + wave * 32 + l31;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + 32 * t + 8 * g + 4 * lh;
          auxq[4 * t + g] = *reinterpret_cast<const float4*>(m < d.M && n < d.N ? d.aux_in + (int64_t)m * d.ldaux + n : zp);
        }
    }
    mac(nchunks - 1, last_steps);
  };
  if (two)
    run(std::true_type{});
  else
    run(std::false_type{});
  epilogue<0, float4[8], 2, float4[8]>(args, acc, m0, n0, 0, bias_s, resq, -1, 0, auxq);
}
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(const GemmArgs args) { gemm_nt_glds_body<false>(args); }
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_x3_kernel(const GemmArgs args) { gemm_nt_glds_body<true>(args); }
__global__ __launch_bounds__(256, 2) void gemm_nn_glds_x3_kernel(const GemmArgs args) { gemm_nt_glds_body<true, true>(args); }

// 64-row variant of gemm_nt_glds_kernel for launches that would not fill the chip with 128-row tiles (M = 16 384 tokens:
// 384 tiles of 128 x 64 for N = 180 on 768 resident slots).  The 4 waves form a 2 x 2 grid of 32 x 32 tiles (one
// accumulator each: one A and one B ds_read_b128 per 4 MFMAs), everything else — DMA staging, XOR-swizzled images, one
// barrier per chunk, trimmed last chunk, bias tile in LDS, residual quads requested under the last chunk — as above.
constexpr int BM2 = 64;
// BT (X3 only): B is given as [K][N] (the NN form: Linear backward-data, B = weight (out, in) with K = out) — only the
// register staging of the B tile differs: a thread gathers its 8 reduction values of column n with 8 dword loads.
template <bool X3, bool BT = false>
__device__ __forceinline__ void gemm_nt_glds64_body(const GemmArgs& args) {
  static_assert(X3 || !BT, "the [K][N] form of B needs the register-staged (X3) B tile");
  const neosr_gemm_desc& d = args.d;
  constexpr int BSZ = X3 ? BN * 48 : BN * BK;   // (X3: the B tile as three bf16 planes, see gemm_nt_glds_body)
  __shared__ __attribute__((aligned(1024))) float lds[2 * (BM2 * BK + BSZ)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int tiles = args.tiles_m * args.tiles_n;
  const int chunk = gridDim.x >> 3;
  const int logical = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (logical >= tiles) return;
  const int m0 = (logical / args.tiles_n) * BM2, n0 = (logical % args.tiles_n) * BN;
  const int K = d.K;
  const bool fast3 = __builtin_amdgcn_readfirstlane(args.fast3) != 0;
  const int rsub = lane >> 3, slot = lane & 7;
  const float* zp = gm_zero_page;
  asm volatile("" : "+s"(zp));
  __shared__ __attribute__((aligned(16))) float bias_s[BN];
  // (requested here, stored to LDS behind the first chunk's loads: written `bias_s[tid] = d.bias[..]` in one statement the
  // store waited vmcnt(0) for the load right here — one exposed L2 round trip per workgroup in front of the first DMA;
  // round 6, found in the ISA)
  float bias_v = 0.f;
  if (d.bias && tid < BN) bias_v = n0 + tid < d.N ? d.bias[n0 + tid] : 0.f;
  // (DMA addressing as in gemm_nt_glds_kernel: per-lane byte offsets once, the chunk as the scalar offset)
  const int kc_last = (K + BK - 1) / BK - 1;
  constexpr int OOB = 0x7ffffff0;
  const auto rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, 0x7ffffff0, 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, 0x7ffffff0, 0x00020000);
  const gm_i32x4 rA_w = {__builtin_amdgcn_readfirstlane((int)(uintptr_t)d.A),
                         __builtin_amdgcn_readfirstlane((int)(((uintptr_t)d.A >> 32) & 0xffff)), 0x7ffffff0, 0x00020000};
  int offA[2], offA_l[2], offB[2], offB_l[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (4 * i + wave) * 8 + rsub;
    const int kq = 4 * (slot ^ ((r >> 1) & 7));
    const bool oka = m0 + r < d.M, okb = n0 + r < d.N;
    offA[i] = oka ? ((m0 + r) * d.lda + kq) * 4 : OOB;
    offA_l[i] = (oka && kc_last * BK + kq < K) ? offA[i] : OOB;
    offB[i] = okb ? ((n0 + r) * d.ldb + kq) * 4 : OOB;
    offB_l[i] = (okb && kc_last * BK + kq < K) ? offB[i] : OOB;
  }
  typedef __attribute__((address_space(3))) void* lds_vp;
  auto issue = [&](int k0, int buf) {
    float* abuf = lds + buf * (BM2 * BK + BSZ);
    float* bbuf = abuf + BM2 * BK;
    const bool last = k0 == kc_last * BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (X3)
        gm_dma16(rA_w, abuf + (4 * i + wave) * 256, last ? offA_l[i] : offA[i], k0 * 4);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vp)(abuf + (4 * i + wave) * 256), 16, last ? offA_l[i] : offA[i],
                                                 k0 * 4, 0, 0);
      if constexpr (!X3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vp)(bbuf + (4 * i + wave) * 256), 16, last ? offB_l[i] : offB[i],
                                                 k0 * 4, 0, 0);
    }
  };
  const int bn = tid >> 2, bo = tid & 3;
  const bool bn_ok = n0 + bn < d.N;
  const int offB3 = bn_ok ? ((n0 + bn) * d.ldb + 8 * bo) * 4 : OOB;
  const int offB3_l0 = (bn_ok && kc_last * BK + 8 * bo < K) ? offB3 : OOB;
  const int offB3_l1 = (bn_ok && kc_last * BK + 8 * bo + 4 < K) ? offB3 + 16 : OOB;
  f32x4 breg0 = {0.f, 0.f, 0.f, 0.f}, breg1 = {0.f, 0.f, 0.f, 0.f};
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(rB, 0, 0, 0)) rawq_t;
  // BT: rows of B are reduction indices: the resource ends behind row K - 1 (rows past K read zeros), the row is the
  // scalar offset, the column the lane offset
  const auto rBt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, BT ? (unsigned)(((int64_t)K * d.ldb) * 4) : 0u, 0x00020000);
  const int offBt = bn_ok ? (8 * bo * d.ldb + n0 + bn) * 4 : OOB;
  auto bload = [&](int k0) {
    if constexpr (BT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        breg0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBt, offBt, (k0 + e) * d.ldb * 4, 0));
        breg1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBt, offBt, (k0 + 4 + e) * d.ldb * 4, 0));
      }
      return;
    }
    const bool last = k0 == kc_last * BK;
    breg0 = __builtin_bit_cast(f32x4, (rawq_t)__builtin_amdgcn_raw_buffer_load_b128(rB, last ? offB3_l0 : offB3, k0 * 4, 0));
    breg1 = __builtin_bit_cast(f32x4, (rawq_t)__builtin_amdgcn_raw_buffer_load_b128(rB, last ? offB3_l1 : (offB3 == OOB ? OOB : offB3 + 16), k0 * 4, 0));
  };
  auto bstore = [&](int buf) {
    float* bbuf = lds + buf * (BM2 * BK + BSZ) + BM2 * BK;
    bf16x8 P[3];
    split3(breg0, breg1, P);
#pragma unroll
    for (int p3 = 0; p3 < 3; ++p3)
      *reinterpret_cast<bf16x8*>(bbuf + bn * 48 + 4 * ((4 * p3 + bo) ^ ((bn >> 2) & 3))) = P[p3];
  };
  f32x16 acc[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  const int nchunks = (K + BK - 1) / BK;
  issue(0, 0);
  if constexpr (X3) {
    bload(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    bstore(0);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  if (d.bias && tid < BN) bias_s[tid] = bias_v;
  __syncthreads();
  const int arow = wm * 32 + l31, brow = wn * 32 + l31;
  const int last_steps = (K - (nchunks - 1) * BK + 7) >> 3;
  auto mac = [&](int c, int nsteps) {
    const float* abuf = lds + (c & 1) * (BM2 * BK + BSZ);
    const float* bbuf = abuf + BM2 * BK;
    if constexpr (X3) {
      const int sa = (arow >> 1) & 7, sb = (brow >> 2) & 3;
#pragma unroll
      for (int ps = 0; ps < BK / 16; ++ps) {
        if (2 * ps < nsteps) {
          const int q = 4 * ps + 2 * lh;
          bf16x8 X[3], W[3];
          split3(*reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * (q ^ sa)),
                 *reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * ((q + 1) ^ sa)), X);
#pragma unroll
          for (int p3 = 0; p3 < 3; ++p3)
            W[p3] = *reinterpret_cast<const bf16x8*>(bbuf + brow * 48 + 4 * ((4 * p3 + 2 * ps + lh) ^ sb));
          acc[0] = mac6(W, X, acc[0], fast3);
        }
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      if (s < nsteps) {
        const int q = 2 * s + lh;
        const f32x4 a = *reinterpret_cast<const f32x4*>(abuf + arow * BK + 4 * (q ^ ((arow >> 1) & 7)));
        const f32x4 b = *reinterpret_cast<const f32x4*>(bbuf + brow * BK + 4 * (q ^ ((brow >> 1) & 7)));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[e], a[e], acc[0], 0, 0, 0);
      }
    }
  };
  for (int c = 0; c + 1 < nchunks; ++c) {
    issue((c + 1) * BK, (c + 1) & 1);
    if constexpr (X3) bload((c + 1) * BK);
    mac(c, BK / 8);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if constexpr (X3) bstore((c + 1) & 1);
    __syncthreads();
  }
  float4 resq[4], auxq[4];
  if (d.res) {
    const int m = m0 + arow;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 32 + 8 * g + 4 * lh;
      resq[g] = *reinterpret_cast<const float4*>(m < d.M && n < d.N ? d.res + (int64_t)m * d.ldres + n : zp);
    }
  }
  if (d.aux_in) {   // (the GELU' operand, like the residual: requested under the last chunk)
    const int m = m0 + arow;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * 32 + 8 * g + 4 * lh;
      auxq[g] = *reinterpret_cast<const float4*>(m < d.M && n < d.N ? d.aux_in + (int64_t)m * d.ldaux + n : zp);
    }
  }
  mac(nchunks - 1, last_steps);
  epilogue<0, float4[4], 1, float4[4]>(args, acc, m0, n0, 0, bias_s, resq, wm * 32, wn * 32, auxq);
}
__global__ __launch_bounds__(256, 2) void gemm_nt_glds64_kernel(const GemmArgs args) { gemm_nt_glds64_body<false>(args); }
__global__ __launch_bounds__(256, 2) void gemm_nt_glds64_x3_kernel(const GemmArgs args) { gemm_nt_glds64_body<true>(args); }
__global__ __launch_bounds__(256, 2) void gemm_nn_glds64_x3_kernel(const GemmArgs args) { gemm_nt_glds64_body<true, true>(args); }


// (Round 4 tried a whole-K-panel variant for the K = 180 Linears — a 32 x 64 tile whose six chunks are all requested at
// once behind one wait and one barrier, four waves = 2 column tiles x 2 k-halves, two workgroups per CU — on the theory
// that a 0.5 us chunk cannot hide the DMA it waits for: 52 TF against 78 at M = 32 768 (qkv), 50 against 69 at M = 16 384;
// swinir_medium 38.4 -> 41.0 ms per step.  One accumulator chain per wave and two LDS reads per four MFMAs lose more than
// the barriers cost; the chunked kernels above stay.)

// TN GEMM fed from registers (weight gradients dW[m][n] = sum_t dY[t][m] X[t][n]): both operands are
// contiguous along their OUTPUT index, so an MFMA fragment is a plain coalesced row load — lane (l31, lh) reads
// 3 consecutive dY columns and 2 consecutive X columns of token 2s + lh (buffer_load_dwordx3 / dwordx2; a
// half-wave covers 384 / 256 contiguous bytes of one token row) and feeds them to 3 x 2 MFMAs whose row /
// column index is interleaved (m = m0 + 3 j + e1, n = n0 + 2 i + e2).  No LDS, no barrier: every WAVE is an
// independent task — one 96 x 64 output tile (6 accumulators) over its own run of tokens — with the next batch of
// RT_P token pairs in flight while the current one multiplies (one wave per SIMD, so the lookahead is in
// registers, not in a second wave).  The 96 / 64 granularity pads the transformer shapes (180, 360, 540) by
// 6.7 % per dimension where the 128 x 64 LDS tile pads 180 -> 256.  All tiles of a token split run on one XCD
// (its rows are fetched from HBM once and re-read from that L2); the number of splits is chosen so that the
// tasks just fill the 128 SIMDs of an XCD.  Partial tiles go to per-split slabs that `colsum_kernel` sums in
// split order, as before.
#ifndef RT_P
#define RT_P 16
#endif
constexpr int RT_M = 96, RT_N = 64;

#ifndef RT_OCC
#define RT_OCC 1
#endif
// RS: row_scale[token / rows_per_scale] multiplies the dY rows (the DropPath scale of the incoming gradient, per sample):
// rows_per_scale is a multiple of 32 and so is every run start, so a batch of 32 tokens has ONE scale — a scalar kept
// beside each fragment batch, three v_mul per token pair.
// one task = one wave: output tile `tile` over the token run of split `split`.  RS with d.row_scale == nullptr scales by 1
// (exact): the grouped launch below runs scaled and unscaled problems with one instantiation.
template <bool RS>
__device__ __forceinline__ void tn_reg_task(const GemmArgs& args, int split, int tile) {
  const neosr_gemm_desc& d = args.d;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int m0 = (tile / args.tiles_n) * RT_M, n0 = (tile % args.tiles_n) * RT_N;
  const int M = d.M, N = d.N;
  const int t_lo = split * args.ksplit_len;
  const int t_hi = min(d.K, t_lo + args.ksplit_len);
  // Operands come through buffer loads: a per-wave resource descriptor (base = the matrix, extent = the end of
  // THIS wave's token run) + 32-bit lane offset + scalar row offset, so the address math is scalar.  Lanes whose
  // columns fall outside the matrix read column 0 instead: their products only reach accumulator rows / columns
  // that are never stored.  Tokens past the run (last batch) get an out-of-range lane offset and read 0.
  const bool a_ok = m0 + 3 * l31 + 2 < M, b_ok = n0 + 2 * l31 + 1 < N;
  const int oa = ((a_ok ? m0 + 3 * l31 : 0) + lh * d.lda) * 4, ob = ((b_ok ? n0 + 2 * l31 : 0) + lh * d.ldb) * 4;
  const __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, d.K * d.lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, d.K * d.ldb * 4, 0x00020000);
  const int sa = 8 * d.lda, sb = 8 * d.ldb;  // bytes per token pair
  int ta = t_lo * d.lda * 4, tb = t_lo * d.ldb * 4;  // scalar offsets of the next batch

  f32x16 acc[3][2];
#pragma unroll
  for (int e1 = 0; e1 < 3; ++e1)
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e1][e2][r] = 0.f;
  float cs[3] = {0.f, 0.f, 0.f};

  // fragments stay in the loads' own 3- / 2-dword register tuples (re-packing them into one array makes the
  // compiler build a wide tuple out of moves that wait for the batch just issued)
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b96(ra, 0, 0, 0)) frag3;
  typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(rb, 0, 0, 0)) frag2;
  frag3 a0[RT_P], a1[RT_P];
  frag2 b0[RT_P], b1[RT_P];
  const int ntok = t_hi - t_lo;
  const int nfull = ntok / (2 * RT_P);
  const int nb = nfull + (ntok % (2 * RT_P) ? 1 : 0);  // batches of RT_P token pairs, the last one ragged
  // one pair of the batch: its two loads / its 6 MFMAs
  auto ld = [&](frag3& av, frag2& bv, int p, bool full, int t0) {
    const bool ok = full || t0 + 2 * p < t_hi;  // tokens past the run: out-of-range lane offset -> 0
    av = __builtin_amdgcn_raw_buffer_load_b96(ra, ok ? oa : 0x7ffffff0, ta + p * sa, 0);
    bv = __builtin_amdgcn_raw_buffer_load_b64(rb, ok ? ob : 0x7ffffff0, tb + p * sb, 0);
  };
  // scale of the batch in a0 / a1 (sc0 / sc1); grp / left walk the scale groups without a division per batch
  float sc0 = 1.f, sc1 = 1.f;
  const bool rs_on = RS && d.row_scale != nullptr;
  int grp = rs_on ? t_lo / d.rows_per_scale : 0, left = rs_on ? d.rows_per_scale - (t_lo - grp * d.rows_per_scale) : 0;
  auto next_scale = [&]() {
    float v = 1.f;
    if (rs_on) {
      v = d.row_scale[grp];
      left -= 2 * RT_P;
      if (left <= 0) {
        ++grp;
        left += d.rows_per_scale;
      }
    }
    return v;
  };
  auto mm = [&](const frag3& av, const frag2& bv, float sc) {
#pragma unroll
    for (int e1 = 0; e1 < 3; ++e1) {
      const float a = RS ? __uint_as_float(av[e1]) * sc : __uint_as_float(av[e1]);
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2)
        acc[e1][e2] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(bv[e2]), a, acc[e1][e2], 0, 0, 0);
      cs[e1] += a;  // column sums of dY (kept only by the n-tile-0 tasks)
    }
  };
  int ib = 0;  // index of the next batch to load
  auto load = [&](frag3 (&ab)[RT_P], frag2 (&bb)[RT_P], float& sc) {
    sc = next_scale();
    const bool full = ib < nfull;
    const int t0 = t_lo + ib * 2 * RT_P + lh;
#pragma unroll
    for (int p = 0; p < RT_P; ++p) ld(ab[p], bb[p], p, full, t0);
    ta += RT_P * sa;
    tb += RT_P * sb;
    ++ib;
  };
  auto mac = [&](const frag3 (&ab)[RT_P], const frag2 (&bb)[RT_P], float sc) {
#pragma unroll
    for (int p = 0; p < RT_P; ++p) mm(ab[p], bb[p], sc);
  };
  // steady state: multiply batch `cur` while refilling `nxt` (all full batches), the two loads of a pair issued
  // right behind the 6 MFMAs of the same slot so the matrix pipe never waits for an address burst
  auto step = [&](const frag3 (&ca)[RT_P], const frag2 (&cb)[RT_P], float csc, frag3 (&na)[RT_P], frag2 (&nb_)[RT_P],
                  float& nsc) {
    nsc = next_scale();
#pragma unroll
    for (int p = 0; p < RT_P; ++p) {
      mm(ca[p], cb[p], csc);
      ld(na[p], nb_[p], p, true, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    ta += RT_P * sa;
    tb += RT_P * sb;
    ++ib;
  };
  if (nb > 0) {
    load(a0, b0, sc0);
    while (ib + 2 <= nfull) {  // the next two batches to load are full ones
      step(a0, b0, sc0, a1, b1, sc1);
      step(a1, b1, sc1, a0, b0, sc0);
    }
    // a0 holds the last loaded batch; 0, 1 or 2 batches (one full and / or the ragged one) are left to load
    const int rem = nb - ib;
    if (rem == 0) {
      mac(a0, b0, sc0);
    } else {
      load(a1, b1, sc1);
      mac(a0, b0, sc0);
      if (rem == 2) {
        load(a0, b0, sc0);
        mac(a1, b1, sc1);
        mac(a0, b0, sc0);
      } else {
        mac(a1, b1, sc1);
      }
    }
  }

  float* slab = d.C + (int64_t)split * args.slab;
  const bool do_colsum = args.colsum_part && n0 == 0;
#pragma unroll
  for (int e1 = 0; e1 < 3; ++e1) {
    const int m = m0 + 3 * l31 + e1;
    const float csum = cs[e1] + __shfl_xor(cs[e1], 32);  // even + odd tokens
    if (m >= M) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // D rows i = 8 g + 4 lh + r  ->  n = n0 + 2 i + e2: (r, e2) walk 8 consecutive columns
      const int n = n0 + 16 * g + 8 * lh;
      float* o = slab + (int64_t)m * N + n;
      if (n < N)
        *reinterpret_cast<float4*>(o) = make_float4(acc[e1][0][4 * g], acc[e1][1][4 * g], acc[e1][0][4 * g + 1],
                                                    acc[e1][1][4 * g + 1]);
      if (n + 4 < N)
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[e1][0][4 * g + 2], acc[e1][1][4 * g + 2],
                                                        acc[e1][0][4 * g + 3], acc[e1][1][4 * g + 3]);
    }
    if (do_colsum && lh == 0) args.colsum_part[(int64_t)split * args.slab + m] = csum;
  }
}

template <bool RS>
__global__ __launch_bounds__(256, RT_OCC) void gemm_tn_reg_kernel(const GemmArgs args) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tiles = args.tiles_m * args.tiles_n;
  const int xcd = blockIdx.x & 7, task = (blockIdx.x >> 3) * 4 + wave;  // task index inside this XCD
  const int split = (task / tiles) * 8 + xcd, tile = task % tiles;
  if (split >= args.nsplit) return;
  tn_reg_task<RS>(args, split, tile);
}

// Up to four weight-gradient GEMMs in ONE launch (round 4): the four Linears of a transformer block — fc2, fc1, proj, qkv —
// whose operands all exist once the block's data-gradient chain has run (csrc/blocks.hip).  Each problem keeps the task
// list it would have on its own (tiles x token splits, the splits dealt over the XCDs); XCD x runs its tasks of problem 0,
// then of problem 1, ...: one wave per task as before, but a single ramp / drain and no stream boundaries between the
// problems (a launch of this kernel is exactly one round of waves over the SIMDs, so four launches drained the chip four
// times).  Same per-task arithmetic: bit-identical partials.
constexpr int TN_GROUP = 4;
struct GemmGroupArgs {
  GemmArgs p[TN_GROUP];
  int start[TN_GROUP + 1];   // first task (inside an XCD) of every problem
  int n;
};
__global__ __launch_bounds__(256, RT_OCC) void gemm_tn_reg_group_kernel(const GemmGroupArgs g) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, task = (blockIdx.x >> 3) * 4 + wave;
  if (task >= g.start[g.n]) return;
  // (an if-chain over the members — every branch reads the kernel arguments at constant offsets, the problem ends up in
  // scalar registers; indexing the argument array with a run-time value would move it to scratch — and ONE copy of the
  // task body behind it)
  GemmArgs a;
  int t;
  if (task < g.start[1]) {
    a = g.p[0]; t = task;
  } else if (task < g.start[2]) {
    a = g.p[1]; t = task - g.start[1];
  } else if (task < g.start[3]) {
    a = g.p[2]; t = task - g.start[2];
  } else {
    a = g.p[3]; t = task - g.start[3];
  }
  const int tiles = a.tiles_m * a.tiles_n, split = (t / tiles) * 8 + xcd;
  if (split >= a.nsplit) return;
  tn_reg_task<true>(a, split, t % tiles);
}

// ---- TN GEMM, bf16x3 form (round 5, default with neosr_set_gemm_x3): weight gradients dW[m][n] = sum_t dY[t][m] X[t][n].
// In the register-fed kernel above every loaded value feeds ONE MFMA, so splitting it into bf16 pieces there costs as much
// as it saves (measured at parity: profiles/NEGATIVE_RESULTS.md 5.8).  Here a workgroup of four waves owns a 192 x 192
// output tile over a run of tokens; per step of 16 tokens the 384 columns (192 of dY, 192 of X) x 16 tokens are gathered
// by the 256 threads — thread = (column, token octet), 8 coalesced dword loads down the column — split ONCE into three
// bf16 pieces and written to LDS as [piece 3][octet 2][column 384] x 16 bytes: exactly the 8-token operand a lane of
// v_mfma_f32_32x32x16_bf16 needs, so the MFMA loop is 18 conflict-free ds_read_b128 + 54 MFMAs per wave and step (a wave
// owns 3 x 3 tiles of 32 x 32; every staged value feeds 6 tiles x 6 cross products) with no vector arithmetic in it.
// Two stages of 36 KB (two workgroups per CU), the next step's 24 loads per thread in flight under the current step's
// MFMAs, one barrier per step.  Token runs are a function of K alone (tn_lds_ksplit), so a problem gets the same partial
// slabs from a single launch and from a grouped one; they are summed in split order by the caller as before.
constexpr int LT = 192;                          // tile side
constexpr int LT_STAGE = 6 * 2 * LT * 4;         // floats per stage: [piece * 2 + octet][column 0..383][4 floats = 8 bf16]
template <bool RS>
__device__ __forceinline__ void tn_lds_task(const GemmArgs& args, int split, int tile) {
  __shared__ __attribute__((aligned(1024))) float stage[2 * LT_STAGE];
  __shared__ float cs_lds[2 * LT];
  const neosr_gemm_desc& d = args.d;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = (tile / args.tiles_n) * LT, n0 = (tile % args.tiles_n) * LT;
  const int M = d.M, N = d.N;
  const bool fast3 = __builtin_amdgcn_readfirstlane(args.fast3) != 0;
  const int t_lo = split * args.ksplit_len;
  const int t_hi = min(d.K, t_lo + args.ksplit_len);
  const int nsteps = (t_hi - t_lo + 15) >> 4;
  // rows past the run read zeros (the resources end behind token t_hi - 1), columns past the matrix get an out-of-range offset
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, t_hi * d.lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, t_hi * d.ldb * 4, 0x00020000);
  // the three staging units of this thread: unit u = tid + 256 k -> column u % 384 (0..191: dY, 192..383: X), octet u / 384;
  // which matrix and which octet a unit has is the same for the 64 lanes of a wave
  int vo[3], col[3];
  bool isa[3];
  int oct[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int u = tid + 256 * k;
    col[k] = u % (2 * LT);
    oct[k] = __builtin_amdgcn_readfirstlane(u / (2 * LT));
    isa[k] = __builtin_amdgcn_readfirstlane(col[k] < LT ? 1 : 0) != 0;
    const int c = isa[k] ? m0 + col[k] : n0 + col[k] - LT;
    vo[k] = (c < (isa[k] ? M : N)) ? c * 4 : 0x7ffffff0;
  }
  float raw[3][8];
  float rsc[3] = {1.f, 1.f, 1.f};   // DropPath row scale of a dY unit's 8 tokens: requested with the unit (read at the split it
                                    // was a scalar-memory round trip per unit and step in front of the LDS stores)
  auto gload = [&](int step) {
    const int t = t_lo + 16 * step;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (RS && isa[k] && d.row_scale) rsc[k] = d.row_scale[min(t + 8 * oct[k], d.K - 1) / d.rows_per_scale];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = t + 8 * oct[k] + e;
        raw[k][e] = isa[k] ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo[k], row * d.lda * 4, 0))
                           : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vo[k], row * d.ldb * 4, 0));
      }
    }
  };
  float cs[3] = {0.f, 0.f, 0.f};   // column sums of dY (the bias gradient) of this thread's dY units
  auto sstore = [&](int step, int buf) {
    float* sb = stage + buf * LT_STAGE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float v[8];
      const float sc = rsc[k];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = (RS && isa[k]) ? raw[k][e] * sc : raw[k][e];
        if (isa[k]) cs[k] += v[e];
      }
      bf16x8 P[3];
      split3v(v, P);
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3)
        *reinterpret_cast<bf16x8*>(sb + ((2 * p3 + oct[k]) * (2 * LT) + col[k]) * 4) = P[p3];
    }
  };
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mac = [&](int buf) {
    const float* sb = stage + buf * LT_STAGE;
    bf16x8 Bf[3][3];
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3)
        Bf[bn][p3] = *reinterpret_cast<const bf16x8*>(sb + ((2 * p3 + lh) * (2 * LT) + LT + wn * 96 + 32 * bn + l31) * 4);
#pragma unroll
    for (int bm = 0; bm < 3; ++bm) {
      bf16x8 Af[3];
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3)
        Af[p3] = *reinterpret_cast<const bf16x8*>(sb + ((2 * p3 + lh) * (2 * LT) + wm * 96 + 32 * bm + l31) * 4);
#pragma unroll
      for (int bn = 0; bn < 3; ++bn) acc[bm][bn] = mac6(Bf[bn], Af, acc[bm][bn], fast3);   // D rows <-> n, columns <-> m
    }
  };
  if (nsteps > 0) {
    gload(0);
    sstore(0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) gload(s + 1);
      mac(s & 1);
      if (s + 1 < nsteps) sstore(s + 1, (s + 1) & 1);
      __syncthreads();
    }
  }
  // partial tile -> this split's slab (caller sums the slabs in split order)
  float* slab = d.C + (int64_t)split * args.slab;
#pragma unroll
  for (int bm = 0; bm < 3; ++bm) {
    const int m = m0 + wm * 96 + 32 * bm + l31;
    if (m >= M) continue;
#pragma unroll
    for (int bn = 0; bn < 3; ++bn)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 96 + 32 * bn + 8 * g + 4 * lh;
        if (n < N)
          *reinterpret_cast<float4*>(slab + (int64_t)m * N + n) =
              make_float4(acc[bm][bn][4 * g], acc[bm][bn][4 * g + 1], acc[bm][bn][4 * g + 2], acc[bm][bn][4 * g + 3]);
      }
  }
  if (args.colsum_part && n0 == 0) {   // (uniform) the two octet halves of every dY column, summed in a fixed order
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (isa[k]) cs_lds[oct[k] * LT + col[k]] = cs[k];
    __syncthreads();
    if (tid < LT && m0 + tid < M) args.colsum_part[(int64_t)split * args.slab + m0 + tid] = cs_lds[tid] + cs_lds[LT + tid];
  }
}
__global__ __launch_bounds__(256, 2) void gemm_tn_lds_x3_kernel(const GemmArgs args) {
  const int tiles = args.tiles_m * args.tiles_n;
  const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  if (split >= args.nsplit) return;
  tn_lds_task<true>(args, split, tile);
}
// up to four problems in one launch (the four Linears of a transformer block, csrc/blocks.hip): workgroup -> (problem,
// split, tile); `start` counts workgroups.  (The body as a macro-free if-chain over the kernel arguments, as in the
// register-fed group kernel.)
__global__ __launch_bounds__(256, 2) void gemm_tn_lds_x3_group_kernel(const GemmGroupArgs g) {
  const int wg = blockIdx.x;
  if (wg >= g.start[g.n]) return;
  GemmArgs a;
  int t;
  if (wg < g.start[1]) {
    a = g.p[0]; t = wg;
  } else if (wg < g.start[2]) {
    a = g.p[1]; t = wg - g.start[1];
  } else if (wg < g.start[3]) {
    a = g.p[2]; t = wg - g.start[2];
  } else {
    a = g.p[3]; t = wg - g.start[3];
  }
  const int tiles = a.tiles_m * a.tiles_n;
  tn_lds_task<true>(a, t / tiles, t % tiles);
}

// column sums of a row-major [rows, cols] matrix (bias gradients, LayerNorm / relative-position-bias
// partials): 64 columns x 4 row lanes per workgroup, each row lane walks its rows with 256-byte
// coalesced wave loads, the 4 lanes are combined through LDS in a fixed order.  Two launches when
// rows > 1024 (per-slab partials, then the same kernel over the partial matrix).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     int rows, int cols, int ld, int rows_per_block,
                                                     int accumulate) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float* p = x + c;
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
      s0 += p[(int64_t)r * ld];
      s1 += p[(int64_t)(r + 4) * ld];
      s2 += p[(int64_t)(r + 8) * ld];
      s3 += p[(int64_t)(r + 12) * ld];
    }
    for (; r < r1; r += 4) s0 += p[(int64_t)r * ld];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < cols) {
    const float s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    float* o = out + (int64_t)blockIdx.y * cols + c;
    *o = accumulate ? *o + s : s;
  }
}

// Batched form of the same reduction (neosr_colsum_many): block b of the launch belongs to job j with
// start[j] <= b < start[j + 1]; inside the job it is block (bx, by) of colsum_kernel's grid.
constexpr int CSM_JOBS = 32;
struct ColsumBatch {
  const float* x[CSM_JOBS];
  float* out[CSM_JOBS];
  int rows[CSM_JOBS], cols[CSM_JOBS], ld[CSM_JOBS], rpb[CSM_JOBS], acc[CSM_JOBS], gx[CSM_JOBS];
  int start[CSM_JOBS + 1];
  int n;
};
__global__ __launch_bounds__(256) void colsum_many_kernel(const ColsumBatch bt) {
  __shared__ float red[4][64];
  int j = 0;
#pragma unroll 1
  for (int i = 1; i < bt.n; ++i)
    if ((int)blockIdx.x >= bt.start[i]) j = i;
  const int local = blockIdx.x - bt.start[j];
  const int gx = bt.gx[j], bx = local % gx, by = local / gx;
  const int rows = bt.rows[j], cols = bt.cols[j], ld = bt.ld[j], rpb = bt.rpb[j];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = bx * 64 + tx;
  const int r0 = by * rpb, r1 = min(rows, r0 + rpb);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    const float* p = bt.x[j] + c;
    int r = r0 + ty;
    for (; r + 12 < r1; r += 16) {
      s0 += p[(int64_t)r * ld];
      s1 += p[(int64_t)(r + 4) * ld];
      s2 += p[(int64_t)(r + 8) * ld];
      s3 += p[(int64_t)(r + 12) * ld];
    }
    for (; r < r1; r += 4) s0 += p[(int64_t)r * ld];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < cols) {
    const float s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    float* o = bt.out[j] + (int64_t)by * cols + c;
    *o = bt.acc[j] ? *o + s : s;
  }
}

// split-K factor of the TN (weight-gradient) GEMM: the output is tiny (<= a few hundred rows and
// columns) while K = B*H*W is huge, so the K range is cut until ~1024 workgroups exist (4 per CU),
// never shorter than 4 chunks per workgroup
#ifndef TN_TARGET_BLOCKS
#define TN_TARGET_BLOCKS 768
#endif
#ifdef GEMM_NO_GLDS
constexpr bool g_no_glds = true;
#else
constexpr bool g_no_glds = false;
#endif
#ifdef GEMM_NO_TNREG
constexpr bool g_no_tnreg = true;
#else
constexpr bool g_no_tnreg = false;
#endif
#ifndef TN_REG_ROUNDS
#define TN_REG_ROUNDS RT_OCC
#endif
int g_tn_rounds = TN_REG_ROUNDS;
int g_bm64_below = 600;   // 128 x 64 tiles of a launch below which the NT GEMM switches to 64-row tiles
std::atomic<int> g_gemm_x3{-1};   // (read by the forward and the autograd thread) bf16x3 products in the NT kernels: -1 = read NEOSR_AMD_GEMM_X3 on first use (default on)
bool gemm_x3() {
  if (g_gemm_x3 < 0) {
    const char* e = getenv("NEOSR_AMD_GEMM_X3");
    g_gemm_x3 = (e && e[0] == '0') ? 0 : 1;
  }
  return g_gemm_x3 == 1;
}

// register-fed TN kernel: usable when the ragged last m tile still splits into whole 3-column lane groups
bool tn_reg_ok(int M, int N, int K) {
  return !g_no_tnreg && (M % RT_M) % 3 == 0 && N % 4 == 0 && (int64_t)K * (M > N ? M : N) * 4 < (1ll << 31);
}
// tokens per split: the (tile, split) tasks — one wave each — just fill the 128 SIMDs of every XCD (g_tn_rounds
// times over), in whole batches of 2 RT_P tokens and never less than 4 batches
int tn_reg_ksplit(int M, int N, int K) {
  const int tiles = ceil_div(M, RT_M) * ceil_div(N, RT_N);
  int per_xcd = (128 * g_tn_rounds) / tiles;
  if (per_xcd < 1) per_xcd = 1;
  int len = ceil_div(ceil_div(K, 8 * per_xcd), 2 * RT_P) * 2 * RT_P;
  const int min_len = 4 * 2 * RT_P;
  if (len < min_len) len = min_len;
  return len;
}

// bf16x3 LDS kernel: tokens per split — a function of K alone (single and grouped launches of a problem cut it the same way):
// 64 splits of a whole number of 32-token pairs of steps, at least 128 tokens each
int tn_lds_ksplit(int K) {
  int len = ceil_div(ceil_div(K, 64), 32) * 32;
  return len < 128 ? 128 : len;
}
bool tn_lds_ok(const neosr_gemm_desc& d) {
  return gemm_x3() && d.M % 4 == 0 && d.N % 4 == 0 && (int64_t)d.K * (d.lda > d.ldb ? d.lda : d.ldb) * 4 < (int64_t(1) << 31) &&
         (!d.row_scale || d.rows_per_scale % 32 == 0);
}

int tn_splits(int M, int N, int K) {
  const int tiles = ceil_div(M, BM) * ceil_div(N, BN);
  int s = ceil_div(TN_TARGET_BLOCKS, tiles);
  const int smax = K / (4 * BK) > 1 ? K / (4 * BK) : 1;
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" int64_t neosr_gemm_workspace_bytes(const neosr_gemm_desc* d) {
  if (!d || d->mode != NEOSR_GEMM_TN) return 256;
  // split-K slabs + the staging area of the column-sum reduction (<= 16384 + M*N floats)
  // (slabs carry M extra floats for the column sums of A; the stage is sized for <= 256 row slabs of 64 columns)
  const int64_t slab = (int64_t)d->M * d->N + d->M;
  int ns = tn_splits(d->M, d->N, d->K);
  if (tn_reg_ok(d->M, d->N, d->K)) {
    const int nr = ceil_div(d->K, tn_reg_ksplit(d->M, d->N, d->K));
    if (nr > ns) ns = nr;
  }
  {   // (the bf16x3 LDS kernel's split count; sized whatever neosr_set_gemm_x3 says now, so a cached size serves both forms)
    const int nl = ceil_div(d->K, tn_lds_ksplit(d->K));
    if (nl > ns) ns = nl;
  }
  return ((int64_t)(ns + 1) * slab + 16384 + 256 * 64 + 64) * 4;
}

extern "C" int neosr_gemm(const neosr_gemm_desc* dp, void* stream) {
  NEOSR_CHECK(dp, "gemm: null descriptor");
  neosr_gemm_desc d = *dp;
  NEOSR_CHECK(d.A && d.B && d.C && d.M > 0 && d.N > 0 && d.K > 0, "gemm: bad args");
  NEOSR_CHECK(d.mode >= 0 && d.mode <= 2, "gemm: bad mode");
  auto al = [](const void* p, int ld) { return !p || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
  NEOSR_CHECK(al(d.A, d.lda) && al(d.C, d.ldc) && al(d.res, d.ldres) && al(d.aux_in, d.ldaux) &&
                  al(d.aux_out, d.ldaux) && d.ldb % 4 == 0,
              "gemm: A/C/res/aux must be 16-byte aligned and all leading dimensions multiples of 4");
  NEOSR_CHECK((uintptr_t)d.B % 4 == 0 && (uintptr_t)d.bias % 4 == 0, "gemm: B / bias must be 4-byte aligned");
  NEOSR_CHECK(d.N % 4 == 0 && (d.mode == NEOSR_GEMM_TN ? d.M % 4 == 0 : d.K % 4 == 0),
              "gemm: N and the contiguous reduction/row extent must be multiples of 4");
  NEOSR_CHECK(!d.row_scale || d.rows_per_scale > 0, "gemm: row_scale needs rows_per_scale");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a;
  a.d = d;
  a.ksplit_len = d.K;
  a.colsum_part = nullptr;
  a.slab = 0;
  a.b_vec = al(d.B, d.ldb) && al(d.bias, 0);
  a.fast3 = (gemm_x3() && neosr_conv::fast_matmul()) ? 1 : 0;
  a.tiles_m = ceil_div(d.M, BM);
  a.tiles_n = ceil_div(d.N, BN);
  dim3 grid(ceil_div(a.tiles_m * a.tiles_n, 8) * 8, 1, 1);
  const bool prof = neosr_prof_on();
  if (prof)  // algorithmic work of the product itself: 2MNK FLOP, 4(MK + NK + MN) bytes
    neosr_prof_begin(NEOSR_PROF_GEMM_NT + d.mode, stream, 2.0 * d.M * d.N * d.K,
                     4.0 * ((double)d.M * d.K + (double)d.N * d.K + (double)d.M * d.N));
  // (the direct-to-LDS kernels address A and B through buffer resources with 32-bit byte offsets)
  const bool nt_small = (int64_t)d.M * d.lda * 4 < (int64_t(1) << 31) && (int64_t)d.N * d.ldb * 4 < (int64_t(1) << 31);
  if (d.mode == NEOSR_GEMM_NT && a.b_vec && !g_no_glds && nt_small) {
    // 128-row tiles when they fill the chip's resident workgroup slots at least ~twice, else 64-row tiles
    static const int env64 = [] { const char* e = getenv("NEOSR_GEMM_BM64"); return e ? atoi(e) : -1; }();
    // (768 = 3 resident 128-row workgroups per CU: a launch whose last round is at most 60 % full — M = 16 384: N = 180 is
    // half a round, N = 540 one and a half — runs faster as twice as many 64-row tiles; measured, tools/bench_linear.py)
    const int t128 = a.tiles_m * a.tiles_n, tail = t128 % 768;
    const bool bm64 = env64 >= 0 ? env64 != 0 : (t128 < g_bm64_below || (t128 < 2 * 768 && tail > 0 && tail <= 460));
    if (bm64) {
      a.tiles_m = ceil_div(d.M, BM2);
      grid.x = ceil_div(a.tiles_m * a.tiles_n, 8) * 8;
      if (gemm_x3()) hipLaunchKernelGGL(gemm_nt_glds64_x3_kernel, grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL(gemm_nt_glds64_kernel, grid, dim3(256), 0, st, a);
    } else {
      if (gemm_x3()) hipLaunchKernelGGL(gemm_nt_glds_x3_kernel, grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL(gemm_nt_glds_kernel, grid, dim3(256), 0, st, a);
    }
  } else if (d.mode == NEOSR_GEMM_NT) {
    hipLaunchKernelGGL(gemm_mfma_kernel<0>, grid, dim3(256), 0, st, a);
  } else if (d.mode == NEOSR_GEMM_NN) {
    // (same last-round rule as the NT kernel; the staged kernel keeps 5 workgroups of 128 rows per CU resident)
    static const int env64 = [] { const char* e = getenv("NEOSR_GEMM_NN64"); return e ? atoi(e) : -1; }();
    const int t128 = a.tiles_m * a.tiles_n, tail = t128 % 768;
    // bf16x3 products: the direct-to-LDS kernels with the [K][N] weight tile gathered into their register staging
    const bool nn_small = (int64_t)d.M * d.lda * 4 < (int64_t(1) << 31) && (int64_t)d.K * d.ldb * 4 < (int64_t(1) << 31);
    if (gemm_x3() && !g_no_glds && nn_small) {
      if (env64 >= 0 ? env64 != 0 : (d.aux_in || t128 < g_bm64_below || (t128 < 2 * 768 && tail > 0 && tail <= 460))) {
        a.tiles_m = ceil_div(d.M, BM2);
        grid.x = ceil_div(a.tiles_m * a.tiles_n, 8) * 8;
        hipLaunchKernelGGL(gemm_nn_glds64_x3_kernel, grid, dim3(256), 0, st, a);
      } else {
        hipLaunchKernelGGL(gemm_nn_glds_x3_kernel, grid, dim3(256), 0, st, a);
      }
      if (prof) neosr_prof_end(stream);
      NEOSR_LAUNCH_CHECK();
      return 0;
    }
    // ... and always under the GELU' epilogue (fc2's data gradient): half-size accumulators interleave that long
    // vector epilogue with other workgroups' MFMAs (M = 32 768: 71.6 -> 62.9 us)
    if (env64 >= 0 ? env64 != 0 : (d.aux_in || t128 < g_bm64_below || (t128 < 2 * 768 && tail > 0 && tail <= 460))) {
      a.tiles_m = ceil_div(d.M, 64);
      grid.x = ceil_div(a.tiles_m * a.tiles_n, 8) * 8;
      hipLaunchKernelGGL((gemm_mfma_kernel<1, 64>), grid, dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL(gemm_mfma_kernel<1>, grid, dim3(256), 0, st, a);
    }
  } else {
    NEOSR_CHECK(d.workspace, "gemm TN: workspace missing");
    // (a row scale must be constant over the register kernel's 32-token batches)
    const bool reg = tn_reg_ok(d.M, d.N, d.K) && (!d.row_scale || d.rows_per_scale % (2 * RT_P) == 0);
    const bool ldsx3 = tn_lds_ok(d);
    if (ldsx3) {
      a.ksplit_len = tn_lds_ksplit(d.K);
      a.tiles_m = ceil_div(d.M, LT);
      a.tiles_n = ceil_div(d.N, LT);
    } else if (reg) {
      a.ksplit_len = tn_reg_ksplit(d.M, d.N, d.K);
      a.tiles_m = ceil_div(d.M, RT_M);
      a.tiles_n = ceil_div(d.N, RT_N);
    } else {
      const int ns = tn_splits(d.M, d.N, d.K);
      a.ksplit_len = ceil_div(ceil_div(d.K, ns), BK) * BK;
    }
    const int nsplit = ceil_div(d.K, a.ksplit_len);
    float* out = d.C;
    const int ldc = d.ldc;
    NEOSR_CHECK(ldc == d.N, "gemm TN: C must be dense (ldc == N)");
    // slab = [M*N partial C | M partial column sums of A]: when the caller keeps colsum_a right behind C, ONE
    // fixed-order column-sum pass over the [nsplit][slab] matrix finishes both
    const int64_t mn = (int64_t)d.M * d.N;
    a.slab = mn + (d.colsum_a ? d.M : 0);
    a.d.C = d.workspace;
    a.nsplit = nsplit;
    a.colsum_part = d.colsum_a ? d.workspace + mn : nullptr;
    float* stage = d.workspace + (int64_t)nsplit * a.slab;
    grid.x = ldsx3 ? nsplit * a.tiles_m * a.tiles_n
             : reg ? 8 * ceil_div(ceil_div(nsplit, 8) * a.tiles_m * a.tiles_n, 4)
                   : ceil_div(nsplit, 8) * 8 * a.tiles_m * a.tiles_n;
    if (ldsx3)
      hipLaunchKernelGGL(gemm_tn_lds_x3_kernel, grid, dim3(256), 0, st, a);
    else if (reg && d.row_scale)
      hipLaunchKernelGGL(gemm_tn_reg_kernel<true>, grid, dim3(256), 0, st, a);
    else if (reg)
      hipLaunchKernelGGL(gemm_tn_reg_kernel<false>, grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(gemm_mfma_kernel<2>, grid, dim3(256), 0, st, a);
    if (prof) neosr_prof_end(stream);
    NEOSR_LAUNCH_CHECK();
    // accumulate == 2: leave the [nsplit][slab] partials in the workspace for a batched reduction later
    // (neosr_colsum_many: rows = nsplit = -return value, cols = ld = M N (+ M with colsum_a, which must follow C))
    if (d.accumulate == 2) {
      NEOSR_CHECK(!d.colsum_a || d.colsum_a == out + mn, "gemm TN: deferred reduction needs colsum_a right behind C");
      return -nsplit;
    }
    if (d.colsum_a == out + mn)
      return neosr_colsum(d.workspace, out, stage, nsplit, (int)a.slab, (int)a.slab, d.accumulate, stream);
    if (int rc = neosr_colsum(d.workspace, out, stage, nsplit, (int)mn, (int)a.slab, d.accumulate, stream)) return rc;
    if (d.colsum_a)
      return neosr_colsum(d.workspace + mn, d.colsum_a, stage, nsplit, d.M, (int)a.slab, d.accumulate, stream);
    return 0;
  }
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// n <= 4 TN problems (each as neosr_gemm would take it, with its own workspace) in one launch; the split reductions are
// always left to the caller (neosr_colsum_many): nsplit_out[i] = rows of problem i's partial matrix [rows][M N (+ M)].
// Returns -1 when a problem does not qualify for the register-fed kernel (the caller then launches them one by one).
extern "C" int neosr_gemm_tn_group(const neosr_gemm_desc* descs, int32_t n, int32_t* nsplit_out, void* stream) {
  NEOSR_CHECK(descs && nsplit_out && n >= 1 && n <= TN_GROUP, "gemm_tn_group: 1..%d problems", TN_GROUP);
  GemmGroupArgs g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  int start = 0;
  bool all_lds = true;   // every problem takes the bf16x3 LDS kernel -> its group form
  for (int i = 0; i < n; ++i) all_lds = all_lds && descs[i].mode == NEOSR_GEMM_TN && tn_lds_ok(descs[i]);
  for (int i = 0; i < n; ++i) {
    const neosr_gemm_desc& d = descs[i];
    NEOSR_CHECK(d.A && d.B && d.C && d.workspace && d.mode == NEOSR_GEMM_TN && d.M > 0 && d.N > 0 && d.K > 0,
                "gemm_tn_group: bad problem");
    auto al = [](const void* p, int ld) { return !p || ((uintptr_t)p % 16 == 0 && ld % 4 == 0); };
    NEOSR_CHECK(al(d.A, d.lda) && al(d.B, d.ldb) && d.N % 4 == 0 && d.M % 4 == 0 && d.ldc == d.N,
                "gemm_tn_group: operands must be 16-byte aligned, dense C");
    NEOSR_CHECK(!d.row_scale || d.rows_per_scale > 0, "gemm_tn_group: row_scale needs rows_per_scale");
    const int64_t mn = (int64_t)d.M * d.N;
    NEOSR_CHECK(!d.colsum_a || d.colsum_a == d.C + mn, "gemm_tn_group: colsum_a must sit right behind C");
    if (!all_lds && !(tn_reg_ok(d.M, d.N, d.K) && (!d.row_scale || d.rows_per_scale % (2 * RT_P) == 0))) return -1;
    GemmArgs& a = g.p[i];
    a.d = d;
    a.b_vec = 1;
    a.fast3 = (gemm_x3() && neosr_conv::fast_matmul()) ? 1 : 0;
    a.ksplit_len = all_lds ? tn_lds_ksplit(d.K) : tn_reg_ksplit(d.M, d.N, d.K);
    a.tiles_m = ceil_div(d.M, all_lds ? LT : RT_M);
    a.tiles_n = ceil_div(d.N, all_lds ? LT : RT_N);
    a.nsplit = ceil_div(d.K, a.ksplit_len);
    a.slab = mn + (d.colsum_a ? d.M : 0);
    a.d.C = d.workspace;
    a.colsum_part = d.colsum_a ? d.workspace + mn : nullptr;
    nsplit_out[i] = a.nsplit;
    g.start[i] = start;
    start += (all_lds ? a.nsplit : ceil_div(a.nsplit, 8)) * a.tiles_m * a.tiles_n;
  }
  for (int i = n; i <= TN_GROUP; ++i) g.start[i] = start;
  const bool prof = neosr_prof_on();
  if (prof) {
    double fl = 0, by = 0;
    for (int i = 0; i < n; ++i) {
      fl += 2.0 * descs[i].M * descs[i].N * descs[i].K;
      by += 4.0 * ((double)descs[i].M * descs[i].K + (double)descs[i].N * descs[i].K + (double)descs[i].M * descs[i].N);
    }
    neosr_prof_begin(NEOSR_PROF_GEMM_TN, stream, fl, by);
  }
  if (all_lds) hipLaunchKernelGGL(gemm_tn_lds_x3_group_kernel, dim3(start), dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(gemm_tn_reg_group_kernel, dim3(8 * ceil_div(start, 4)), dim3(256), 0, (hipStream_t)stream, g);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_colsum(const float* x, float* out, float* workspace, int32_t rows, int32_t cols,
                            int32_t ld, int32_t accumulate, void* stream) {
  NEOSR_CHECK(x && out && workspace && rows > 0 && cols > 0 && ld >= cols, "colsum: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int gx = ceil_div(cols, 64);
  // row slabs: enough workgroups to fill the chip when there are few column blocks, and never more
  // than 1024 rows per slab; one launch when a single slab does
  int nblk = ceil_div(256, gx);
  if (nblk < ceil_div(rows, 1024)) nblk = ceil_div(rows, 1024);
  if (nblk > ceil_div(rows, 16)) nblk = ceil_div(rows, 16);
  if (nblk > 256) nblk = 256;
  if (nblk <= 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, 1), dim3(256), 0, st, x, out, rows, cols, ld, rows, accumulate);
  } else {
    const int rpb = ceil_div(ceil_div(rows, nblk), 4) * 4;
    nblk = ceil_div(rows, rpb);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, nblk), dim3(256), 0, st, x, workspace, rows, cols, ld, rpb, 0);
    hipLaunchKernelGGL(colsum_kernel, dim3(gx, 1), dim3(256), 0, st, workspace, out, nblk, cols, cols, nblk,
                       accumulate);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

namespace {
// row slabs of one job, exactly as neosr_colsum cuts them (same summation order as the unbatched call)
inline void colsum_slabs(int rows, int cols, int& nblk, int& rpb) {
  const int gx = ceil_div(cols, 64);
  nblk = ceil_div(256, gx);
  if (nblk < ceil_div(rows, 1024)) nblk = ceil_div(rows, 1024);
  if (nblk > ceil_div(rows, 16)) nblk = ceil_div(rows, 16);
  if (nblk > 256) nblk = 256;
  if (nblk <= 1) {
    nblk = 1;
    rpb = rows;
  } else {
    rpb = ceil_div(ceil_div(rows, nblk), 4) * 4;
    nblk = ceil_div(rows, rpb);
  }
}
}  // namespace

extern "C" int64_t neosr_colsum_many_workspace_floats(const neosr_colsum_item* items, int32_t n) {
  int64_t tot = 64;
  for (int i = 0; items && i < n; ++i) {
    int nblk, rpb;
    colsum_slabs(items[i].rows, items[i].cols, nblk, rpb);
    if (nblk > 1) tot += (int64_t)nblk * items[i].cols;
  }
  return tot;
}

extern "C" int neosr_colsum_many(const neosr_colsum_item* items, int32_t n, float* workspace, void* stream) {
  NEOSR_CHECK(items && n > 0 && workspace, "colsum_many: bad args");
  hipStream_t st = (hipStream_t)stream;
  int64_t woff = 0;
  for (int base = 0; base < n; base += CSM_JOBS) {
    const int m = n - base < CSM_JOBS ? n - base : CSM_JOBS;
    ColsumBatch a, b;  // stage 1 (all jobs), stage 2 (the jobs cut into row slabs)
    a.n = m;
    b.n = 0;
    int sa = 0, sb = 0;
    for (int i = 0; i < m; ++i) {
      const neosr_colsum_item& it = items[base + i];
      NEOSR_CHECK(it.x && it.out && it.rows > 0 && it.cols > 0 && it.ld >= it.cols, "colsum_many: bad item");
      int nblk, rpb;
      colsum_slabs(it.rows, it.cols, nblk, rpb);
      const int gx = ceil_div(it.cols, 64);
      float* part = workspace + woff;
      a.x[i] = it.x; a.rows[i] = it.rows; a.cols[i] = it.cols; a.ld[i] = it.ld; a.rpb[i] = rpb; a.gx[i] = gx;
      a.out[i] = nblk > 1 ? part : it.out;
      a.acc[i] = nblk > 1 ? 0 : it.accumulate;
      a.start[i] = sa;
      sa += gx * nblk;
      if (nblk > 1) {
        const int k = b.n++;
        b.x[k] = part; b.out[k] = it.out; b.rows[k] = nblk; b.cols[k] = it.cols; b.ld[k] = it.cols; b.rpb[k] = nblk;
        b.gx[k] = gx; b.acc[k] = it.accumulate; b.start[k] = sb;
        sb += gx;
        woff += (int64_t)nblk * it.cols;
      }
    }
    a.start[m] = sa;
    b.start[b.n] = sb;
    hipLaunchKernelGGL(colsum_many_kernel, dim3(sa), dim3(256), 0, st, a);
    if (b.n) hipLaunchKernelGGL(colsum_many_kernel, dim3(sb), dim3(256), 0, st, b);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// The Linear GEMMs' products on the bf16 MFMA from bf16x3 operands (1, default; env NEOSR_AMD_GEMM_X3) or on the fp32 MFMA
// (0).  Both are fp32-faithful (see split3); the results differ in the last bits.  Returns the previous setting.
extern "C" int neosr_set_gemm_x3(int on) {
  const int prev = gemm_x3() ? 1 : 0;
  g_gemm_x3 = on ? 1 : 0;
  return prev;
}
