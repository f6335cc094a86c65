// conv3x3.hip — 3x3/s1/p1 convolution family for gfx950 (MI355X), channels-last activations.
//
//   conv3x3_mfma_kernel<DGRAD>  forward and backward-data as an implicit GEMM on
//                               v_mfma_f32_32x32x2_f32 (exact fp32, fp32 accumulate):
//                               M = 32 output pixels of one image row, N = 32 output channels,
//                               K = (channel chunk of 16) x 9 taps.  A 4x32-pixel output tile per
//                               256-thread workgroup (one row per wavefront), the 6x34 input halo
//                               tile and the weight chunk are staged through LDS with the global
//                               loads of chunk c+1 in flight (in registers) while chunk c is on
//                               the matrix pipe.
//   conv3x3_wgrad_kernel        backward-weight: M = 32 cout, N = 32 cin (x 9 taps = 9
//                               accumulators), K = pixels; operands go straight from HBM/L2 to
//                               VGPRs (lanes run along channels, the contiguous dim), split-K
//                               over image rows, partials reduced in a fixed order.
//
// Reference call sites replaced: neosr/archs/esrgan_arch.py:109-116,137-142,196-214;
// neosr/archs/compact_arch.py:76-79 (see include/neosr_amd.h).
#include "common.h"
#include "prof.h"
#include "../../include/neosr_amd.h"

namespace {

constexpr int TH = 4;               // output rows per workgroup (one per wave)
constexpr int TW = 32;              // output cols per workgroup (one MFMA M-tile)
constexpr int CK = 16;              // reduction channels per chunk
constexpr int HALO_W = TW + 2;      // 34
constexpr int HALO_H = TH + 2;      // 6
constexpr int IN_PIX = HALO_H * HALO_W;   // 204
constexpr int INS = CK + 1;         // LDS pixel stride (odd -> conflict-free A reads)
constexpr int NT = 2;               // 32-wide N tiles per workgroup
constexpr int NB = NT * 32;         // 64 output channels per workgroup
constexpr int WROW_F = CK * 9 + 1;  // fwd   weight LDS: [n][k*9+tap], row stride 145
constexpr int WROW_D = NB * 9 + 1;  // dgrad weight LDS: [k][n*9+tap], row stride 577
constexpr int IN_LDS = IN_PIX * INS;                                            // 3468 floats
constexpr int W_LDS = (NB * WROW_F > CK * WROW_D) ? NB * WROW_F : CK * WROW_D;  // 9280 floats
constexpr int IN_F4 = (IN_PIX * 4 + 255) / 256;                                 // 4 float4 / thread
constexpr int W_PER_T = (NB * CK * 9) / 256;                                    // 36 floats / thread

struct ConvArgs {
  neosr_conv_desc d;
  int vec_in;    // `in` rows are 16-byte aligned -> float4 loads
  int vec_mask;
  int tiles_x, tiles_y;
};

template <bool DGRAD, int NTV>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ lin,
                                              const float* __restrict__ lw, int wave, int l31,
                                              int lh, f32x16 (&acc)[NT]) {
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ty = tap / 3, tx = tap % 3;
    const float* ap = lin + ((wave + ty) * HALO_W + l31 + tx) * INS + lh;
#pragma unroll
    for (int ks = 0; ks < CK / 2; ++ks) {
      const float a = ap[ks * 2];
#pragma unroll
      for (int nt = 0; nt < NTV; ++nt) {
        float b;
        if (DGRAD)
          b = lw[(ks * 2 + lh) * WROW_D + (nt * 32 + l31) * 9 + (8 - tap)];
        else
          b = lw[(nt * 32 + l31) * WROW_F + (ks * 2 + lh) * 9 + tap];
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[nt], 0, 0, 0);
      }
    }
  }
}

template <bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ float lds[IN_LDS + W_LDS];
  float* lin = lds;
  float* lw = lds + IN_LDS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;

  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * NB;
  const int nvalid = min(NB, d.N - n0);
  const int ntv = (nvalid + 31) >> 5;

  const int H = d.H, W = d.W, K = d.K;
  const int Hin = d.ups ? (H >> 1) : H, Win = d.ups ? (W >> 1) : W;
  const float* __restrict__ inb = d.in + (int64_t)b * Hin * Win * d.in_cs;
  const float* __restrict__ maskb =
      d.in_mask ? d.in_mask + (int64_t)b * Hin * Win * d.mask_cs : nullptr;

  // per-thread input staging slots: offsets are chunk-invariant
  int in_off[IN_F4], mk_off[IN_F4];
#pragma unroll
  for (int i = 0; i < IN_F4; ++i) {
    const int idx = tid + i * 256;
    in_off[i] = -1;
    mk_off[i] = -1;
    if (idx < IN_PIX * 4) {
      const int pix = idx >> 2, q = idx & 3;
      const int py = pix / HALO_W, px = pix - py * HALO_W;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int sy = d.ups ? (gy >> 1) : gy, sx = d.ups ? (gx >> 1) : gx;
        in_off[i] = (sy * Win + sx) * d.in_cs + q * 4;
        mk_off[i] = (sy * Win + sx) * d.mask_cs + q * 4;
      }
    }
  }

  // weight staging sub-indices (see gload): fwd (n_sub, r_sub) = (tid/16, tid%16),
  // dgrad (k_sub, r_sub) = (tid/64, tid%64)
  const int w_sub_hi = DGRAD ? (tid >> 6) : (tid >> 4);
  const int w_sub_lo = DGRAD ? (tid & 63) : (tid & 15);

  float4 rin[IN_F4];
  float rw[W_PER_T];

  auto gload = [&](int c0) {
#pragma unroll
    for (int i = 0; i < IN_F4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = c0 + (((tid + i * 256) & 3) << 2);
      if (in_off[i] >= 0 && c < K) {
        const float* p = inb + in_off[i] + c0;
        if (args.vec_in && c + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (c + 1 < K) v.y = p[1];
          if (c + 2 < K) v.z = p[2];
          if (c + 3 < K) v.w = p[3];
        }
        if (d.in_prelu) {
          const float* s = d.in_prelu + c;
          v.x = v.x > 0.f ? v.x : v.x * s[0];
          if (c + 1 < K) v.y = v.y > 0.f ? v.y : v.y * s[1];
          if (c + 2 < K) v.z = v.z > 0.f ? v.z : v.z * s[2];
          if (c + 3 < K) v.w = v.w > 0.f ? v.w : v.w * s[3];
        }
        if (maskb) {
          const float* mp = maskb + mk_off[i] + c0;
          float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
          if (args.vec_mask && c + 3 < K) {
            m = *reinterpret_cast<const float4*>(mp);
          } else {
            m.x = mp[0];
            if (c + 1 < K) m.y = mp[1];
            if (c + 2 < K) m.z = mp[2];
            if (c + 3 < K) m.w = mp[3];
          }
          float s0 = d.mask_slope, s1 = s0, s2 = s0, s3 = s0;
          if (d.mask_slopes) {
            const float* s = d.mask_slopes + c;
            s0 = s[0];
            if (c + 1 < K) s1 = s[1];
            if (c + 2 < K) s2 = s[2];
            if (c + 3 < K) s3 = s[3];
          }
          v.x = m.x > 0.f ? v.x : v.x * s0;
          v.y = m.y > 0.f ? v.y : v.y * s1;
          v.z = m.z > 0.f ? v.z : v.z * s2;
          v.w = m.w > 0.f ? v.w : v.w * s3;
        }
      }
      rin[i] = v;
    }
    const int ckv = min(CK, K - c0);
    // Weight staging: every element address is (one per-thread base) + (wave-uniform offset), so
    // the 36 loads share a single address VGPR instead of 36 hoisted ones.
    if (!DGRAD) {
      // fwd: LDS [n][k*9+tap] <- w[(n0+n), c0 .. c0+ckv) (one contiguous run of ckv*9 floats per n)
      // thread -> (n_sub = tid/16, r_sub = tid%16); n = n_sub + 16*nn, r = r_sub + 16*rr
      const float* wp = d.w + ((int64_t)(n0 + w_sub_hi) * d.w_cin + c0) * 9 + w_sub_lo;
      const int rlim = ckv * 9;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          float v = 0.f;
          if (w_sub_hi + 16 * nn < nvalid && w_sub_lo + 16 * rr < rlim)
            v = wp[(int64_t)nn * 16 * d.w_cin * 9 + rr * 16];
          rw[nn * 9 + rr] = v;
        }
    } else {
      // dgrad: LDS [k][n*9+tap] <- w[(c0+k), n0 .. n0+nvalid) (one contiguous run per k)
      // thread -> (k_sub = tid/64, r_sub = tid%64); k = k_sub + 4*kk, r = r_sub + 64*rr
      const float* wp = d.w + ((int64_t)(c0 + w_sub_hi) * d.w_cin + n0) * 9 + w_sub_lo;
      const int rlim = nvalid * 9;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
          float v = 0.f;
          if (w_sub_hi + 4 * kk < ckv && w_sub_lo + 64 * rr < rlim)
            v = wp[(int64_t)kk * 4 * d.w_cin * 9 + rr * 64];
          rw[kk * 9 + rr] = v;
        }
    }
  };

  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < IN_F4; ++i) {
      const int idx = tid + i * 256;
      if (idx < IN_PIX * 4) {
        float* p = lin + (idx >> 2) * INS + ((idx & 3) << 2);
        p[0] = rin[i].x;
        p[1] = rin[i].y;
        p[2] = rin[i].z;
        p[3] = rin[i].w;
      }
    }
    if (!DGRAD) {
      float* lp = lw + w_sub_hi * WROW_F + w_sub_lo;
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) lp[nn * 16 * WROW_F + rr * 16] = rw[nn * 9 + rr];
    } else {
      float* lp = lw + w_sub_hi * WROW_D + w_sub_lo;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) lp[kk * 4 * WROW_D + rr * 64] = rw[kk * 9 + rr];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

  const int nchunks = (K + CK - 1) / CK;
  gload(0);
  for (int c = 0; c < nchunks; ++c) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (c + 1 < nchunks) gload((c + 1) * CK);
    if (ntv == 2)
      compute_chunk<DGRAD, 2>(lin, lw, wave, l31, lh, acc);
    else
      compute_chunk<DGRAD, 1>(lin, lw, wave, l31, lh, acc);
  }

  // epilogue.  D layout (32x32): col j = lane&31 -> channel, row i = (r&3)+8*(r>>2)+4*(lane>>5) -> pixel
  const int y = y0 + wave;
  if (y >= H) return;
  const int64_t rowpix = ((int64_t)b * H + y) * W;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (nt >= ntv) break;
    const int ch = n0 + nt * 32 + l31;
    if (ch >= d.N) continue;
    const float bias = d.bias ? d.bias[ch] : 0.f;
    const float pslope = (d.act == ACT_PRELU) ? d.prelu[ch] : d.slope;
    const bool r1 = d.res1 && ch < d.res1_nch;
    const bool r2 = d.res2 && ch < d.res2_nch;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (x >= W) continue;
      const int64_t pix = rowpix + x;
      float v = acc[nt][r] + bias;
      if (d.act == ACT_LRELU || d.act == ACT_PRELU)
        v = v > 0.f ? v : v * pslope;
      else if (d.act == ACT_RELU)
        v = fmaxf(v, 0.f);
      v *= d.alpha;
      if (r1) v += d.res1[pix * d.res1_cs + ch];
      v *= d.alpha2;
      if (r2) v += d.res2[pix * d.res2_cs + ch];
      float* op = d.out + pix * d.out_cs + ch;
      if (d.accumulate) v += *op;
      *op = v;
    }
  }
}

}  // namespace

extern "C" int neosr_conv3x3(const neosr_conv_desc* dp, void* stream) {
  const neosr_conv_desc& d = *dp;
  NEOSR_CHECK(d.in && d.w && d.out, "conv3x3: null tensor");
  NEOSR_CHECK(d.B > 0 && d.H > 0 && d.W > 0 && d.K > 0 && d.N > 0, "conv3x3: bad geometry");
  NEOSR_CHECK(d.mode == NEOSR_CONV_FWD || d.mode == NEOSR_CONV_DGRAD, "conv3x3: bad mode");
  if (d.mode == NEOSR_CONV_FWD)
    NEOSR_CHECK(d.K == d.w_cin && d.N <= d.w_cout, "conv3x3 fwd: K=%d N=%d vs w (%d,%d)", d.K, d.N,
                d.w_cout, d.w_cin);
  else
    NEOSR_CHECK(d.K == d.w_cout && d.N <= d.w_cin, "conv3x3 dgrad: K=%d N=%d vs w (%d,%d)", d.K,
                d.N, d.w_cout, d.w_cin);
  NEOSR_CHECK(!d.ups || ((d.H % 2 == 0) && (d.W % 2 == 0) && !d.in_mask),
              "conv3x3: ups needs even H,W and no mask");
  NEOSR_CHECK(d.act != NEOSR_ACT_PRELU || d.prelu, "conv3x3: PReLU needs slopes");
  ConvArgs a;
  a.d = d;
  a.vec_in = (d.in_cs % 4 == 0) && ((uintptr_t)d.in % 16 == 0);
  a.vec_mask = d.in_mask && (d.mask_cs % 4 == 0) && ((uintptr_t)d.in_mask % 16 == 0);
  a.tiles_x = ceil_div(d.W, TW);
  a.tiles_y = ceil_div(d.H, TH);
  dim3 grid(a.tiles_x * a.tiles_y * d.B, ceil_div(d.N, NB));
  hipStream_t st = (hipStream_t)stream;
  const bool prof = neosr_prof_on();
  if (prof) {
    const double px = (double)d.B * d.H * d.W;
    // algorithmic traffic: read |x| + |W|, write |y| (fp32)
    neosr_prof_begin(d.mode == NEOSR_CONV_FWD ? NEOSR_PROF_CONV_FWD : NEOSR_PROF_CONV_DGRAD, stream,
                     2.0 * px * d.K * d.N * 9.0,
                     4.0 * (px / (d.ups ? 4.0 : 1.0) * d.K + px * d.N + 9.0 * d.K * d.N));
  }
  if (d.mode == NEOSR_CONV_FWD)
    hipLaunchKernelGGL(conv3x3_mfma_kernel<false>, grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv3x3_mfma_kernel<true>, grid, dim3(256), 0, st, a);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

