// conv_common.h — shared pieces of the 3x3 convolution kernels (conv_mfma.hip: register-staged kernel and
// the dispatch; conv_glds.hip: direct-to-LDS kernel and weight packing; conv_thin.hip: 3- / 1-channel layers).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"
#include "../../include/neosr_amd.h"

namespace neosr_conv {

constexpr int TH = 4;               // output rows per workgroup (one per wave)
constexpr int TW = 32;              // output cols per workgroup (one MFMA M-tile)
constexpr int CK = 16;              // reduction channels per chunk
constexpr int HALO_W = TW + 2;      // 34
constexpr int HALO_H = TH + 2;      // 6
constexpr int IN_PIX = HALO_H * HALO_W;   // 204
constexpr int INS = CK + 1;         // LDS pixel stride (odd -> conflict-free A reads)
constexpr int NT = 2;               // 32-wide N tiles per workgroup
constexpr int NB = NT * 32;         // 64 output channels per workgroup
constexpr int WROW_F = CK * 9 + 1;  // fwd   weight LDS row stride 145
constexpr int WROW_D = NB * 9 + 1;  // dgrad weight LDS row stride 577
constexpr int IN_LDS = IN_PIX * INS;                                            // 3468 floats
constexpr int W_LDS = (NB * WROW_F > CK * WROW_D) ? NB * WROW_F : CK * WROW_D;  // 9280 floats
constexpr int IN_F4 = (IN_PIX * 4 + 255) / 256;                                 // 4 float4 / thread
constexpr int W_F4 = (NB * CK * 9) / (256 * 4);                                 // 9 float4 / thread

struct ConvArgs {
  neosr_conv_desc d;
  int tiles_x, tiles_y;
  int scalar_in;                 // thin-K kernel: the input's channel stride / base is not 16-byte friendly
  int xcd;                       // 1: XCD-aware tile order (see xcd_tile)
  int tx_shift, ty_shift;        // log2 of tiles_x / tiles_y when both are powers of two, else -1 (F(4x4,3x3) kernel)
  unsigned long long* timeline;  // debug only (NEOSR_TIMELINE builds)
};

#ifdef NEOSR_TIMELINE
#define TL_MARK(slot)                                                         \
  do {                                                                        \
    if (args.timeline && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) \
      args.timeline[(threadIdx.x >> 6) * 64 + (slot)] = clock64();             \
  } while (0)
#else
#define TL_MARK(slot) do {} while (0)
#endif


// Workgroup b of a launch is observed to run on XCD b % 8, each XCD with a private 4 MB L2 (placement is not a
// contract: this is a SPEED choice only, any permutation is correct).  With tiles dealt round-robin the vertical
// neighbours of a tile (which share 2 of its 6 halo rows) and the n-blocks of a tile sit on different XCDs, so every
// L2 fetches its own copy of the halo / of the whole input tile (measured 84 MB per RDB conv launch against 43 MB
// algorithmic).  Here XCD x takes the x-th contiguous band of tiles instead; bijective for any tile count.
__device__ __forceinline__ int xcd_tile(int bid, int ntiles, int on) {
  if (!on) return bid;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int x = bid & 7, k = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

unsigned long long* debug_timeline();  // neosr_debug_set_timeline (NEOSR_TIMELINE builds)
bool xcd_enabled();  // NEOSR_AMD_XCD=0 / neosr_set_xcd_aware(0) restores dispatch order (A/B measurements)


// 4x4 / stride-2 kernels run as a 3x3 over the space-to-depth tensor (neosr_conv_desc.s2d_c): a channel
// of sub-pixel (dy, dx) only meets block taps by in {1, dy ? 0 : 2}, bx in {1, dx ? 0 : 2}.  Returns the 9-bit
// mask of live taps in LOOP order (backward-data walks the taps flipped).
__device__ __forceinline__ int s2d_tap_mask(int sub, bool dgrad) {
  const int dy = (sub >> 1) & 1, dx = sub & 1;
  int r1 = dy ? 0 : 2, c1 = dx ? 0 : 2;
  if (dgrad) { r1 = 2 - r1; c1 = 2 - c1; }
  return (1 << 4) | (1 << (3 + c1)) | (1 << (r1 * 3 + 1)) | (1 << (r1 * 3 + c1));
}


// Out-of-range lanes are redirected on the ADDRESS side (to a zero page for loads, to a per-lane
// trash slot for stores) so that no VALU ever touches a loaded value before the LDS store and the
// epilogue is straight-line code: a select on the DATA side makes hipcc wait for the load right
// where it was issued, which serialises the prefetch (measured: 1.6-5.6k cycles per chunk).
static __device__ __attribute__((aligned(256))) float g_zero_page[64];
static __device__ __attribute__((aligned(256))) float g_trash[1024];  // 16 B per thread of a workgroup


// generic guarded 4-channel load (any alignment, ragged channel count)
__device__ __forceinline__ float4 ld4_generic(const float* p, int c, int C, float fill) {
  float4 v = make_float4(fill, fill, fill, fill);
  if (c < C) v.x = p[0];
  if (c + 1 < C) v.y = p[1];
  if (c + 2 < C) v.z = p[2];
  if (c + 3 < C) v.w = p[3];
  return v;
}


// FAST-path epilogue of one 32-channel tile, in two halves so that a kernel can issue the loads
// (bias, slopes, residuals, accumulate-in, derivative mask) ahead of its last chunk of MFMAs.
// D layout (D = W * X^T): lane holds pixel lane&31 and channels nbase + 8g + 4*(lane>>5) + {0..3} in
// acc[4g..4g+3].  Straight-line code: invalid lanes are redirected on the address side.
struct EpiRegs {
  float4 bias[4], sl[4], a0[4], a1[4], a2[4], mk[4];
};

__device__ __forceinline__ void epi_load(const neosr_conv_desc& d, int nbase, int64_t pix, bool pix_ok,
                                         int lh, float s_uni, bool extra, EpiRegs& R) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int chq = nbase + 8 * g + 4 * lh;
    const bool ok = pix_ok && chq < d.N;
    const int cs0 = chq < d.N ? chq : 0;
    R.bias[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    R.sl[g] = make_float4(s_uni, s_uni, s_uni, s_uni);
    R.a0[g] = R.a1[g] = R.a2[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    R.mk[g] = make_float4(1.f, 1.f, 1.f, 1.f);
    if (d.bias) R.bias[g] = *reinterpret_cast<const float4*>(d.bias + cs0);
    if (d.act == ACT_PRELU) R.sl[g] = *reinterpret_cast<const float4*>(d.prelu + cs0);
    if (extra) {
      R.a1[g] = *reinterpret_cast<const float4*>(
          (ok && d.res1 && chq < d.res1_nch) ? d.res1 + pix * d.res1_cs + chq : g_zero_page);
      R.a2[g] = *reinterpret_cast<const float4*>(
          (ok && d.res2 && chq < d.res2_nch) ? d.res2 + pix * d.res2_cs + chq : g_zero_page);
      R.a0[g] = *reinterpret_cast<const float4*>(
          (ok && d.accumulate) ? d.out + pix * d.out_cs + chq : g_zero_page);
    }
    if (d.out_mask)  // invalid lanes read zeros -> scaled garbage goes to the trash slot
      R.mk[g] = *reinterpret_cast<const float4*>(ok ? d.out_mask + pix * d.out_mask_cs + chq : g_zero_page);
  }
}

__device__ __forceinline__ void epi_store(const neosr_conv_desc& d, const f32x16& acc, int nbase,
                                          int64_t pix, bool pix_ok, int lh, int tid, const EpiRegs& R) {
  float4 o[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float bb[4] = {R.bias[g].x, R.bias[g].y, R.bias[g].z, R.bias[g].w};
    const float ss[4] = {R.sl[g].x, R.sl[g].y, R.sl[g].z, R.sl[g].w};
    const float r1[4] = {R.a1[g].x, R.a1[g].y, R.a1[g].z, R.a1[g].w};
    const float r2[4] = {R.a2[g].x, R.a2[g].y, R.a2[g].z, R.a2[g].w};
    const float r0[4] = {R.a0[g].x, R.a0[g].y, R.a0[g].z, R.a0[g].w};
    const float mm[4] = {R.mk[g].x, R.mk[g].y, R.mk[g].z, R.mk[g].w};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = acc[4 * g + e] + bb[e];
      t = t > 0.f ? t : t * ss[e];
      t = t * d.alpha + r1[e];
      t = t * d.alpha2 + r2[e];
      t += r0[e];
      v[e] = mm[e] > 0.f ? t : t * d.out_mask_slope;
    }
    o[g] = make_float4(v[0], v[1], v[2], v[3]);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int chq = nbase + 8 * g + 4 * lh;
    const bool ok = pix_ok && chq < d.N;
    *reinterpret_cast<float4*>(ok ? d.out + pix * d.out_cs + chq : g_trash + tid * 4) = o[g];
  }
}


// launchers of the kernels that live in other translation units
void launch_glds(const ConvArgs& a, dim3 grid, hipStream_t st);
void launch_wino(const ConvArgs& a, hipStream_t st);  // sizes its own grid (8 x 16-pixel tiles x 32-cout blocks)
extern int g_wino4_concurrency;                        // see conv_wino4.hip
int wino4_n64_mode();                                  // neosr_set_wino4_n64: -1 by the fill estimate, 0 never, 1 whenever N > 32
void launch_wino4(const ConvArgs& a, hipStream_t st); // F(4x4,3x3): 16 x 16-pixel tiles x 32-cout blocks, 768 threads
bool fast_matmul();                                    // neosr_set_fast_matmul / NEOSR_AMD_FAST_MATMUL=1: the two-piece bf16 tier of the F(4x4,3x3) kernels
bool wino_enabled();                                   // NEOSR_AMD_WINOGRAD=0 / neosr_set_winograd(0): direct kernel
int wino_mode();                                       // 0 direct, 1 F(2x2,3x3) everywhere, 2 F(4x4,3x3) where an image is given
void launch_thin_k(const ConvArgs& a, dim3 grid, hipStream_t st);
void launch_thin_n(const ConvArgs& a, hipStream_t st);  // sizes its own grid (4 x 64-pixel tiles)

}  // namespace neosr_conv
