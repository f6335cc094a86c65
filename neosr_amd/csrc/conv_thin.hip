// conv_thin.hip — forward / backward-data kernels for layers with 3 (or 1) channels on one side.
#include "conv_common.h"

using namespace neosr_conv;

namespace {

// Thin convolutions: the first / last layers of every network here have 3 (or 1) channels on one side
// (esrgan / swinir / hat conv_first and conv_last, U-Net conv0 / conv9, VGG conv1_1).  On the 32-wide
// tiles above they waste 5-10x of the matrix pipe; these two kernels make them memory-bound instead.
//
// (1) K <= 4 reduction channels (3 -> 64 forward, 64 -> 3 backward-data): the reduction index is packed
//     as k' = tap*4 + ch (36 values, 18 MFMA steps of 32x32x2) instead of 9 taps x a 16-channel chunk.
//     One pass: stage the 6x34x4 halo and the 64 x 36 weight slab, 18 (x2) MFMAs per wave, epilogue.
constexpr int TK_INS = 5;               // LDS pixel stride (odd -> conflict-free b32 reads)
constexpr int TK_WROW = 37;             // weight row stride (36 + 1)

__global__ __launch_bounds__(256, 3) void conv3x3_thin_k_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ float lin[IN_PIX * TK_INS];
  __shared__ float lw[NB * TK_WROW];
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
  const bool dgrad = d.mode == NEOSR_CONV_DGRAD;

  if (tid < IN_PIX) {
    const int py = tid / HALO_W, px = tid - py * HALO_W;
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const float* src = ok ? d.in + (((int64_t)b * H + gy) * W + gx) * d.in_cs : g_zero_page;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (args.scalar_in) {  // channel stride not a multiple of 4 / unaligned base: K scalar loads
      v.x = src[0];
      if (K > 1) v.y = src[1];
      if (K > 2) v.z = src[2];
      if (K > 3) v.w = src[3];
    } else {
      v = *reinterpret_cast<const float4*>(src);
    }
    float* q = lin + tid * TK_INS;  // channels >= K of the quad are padding of the buffer: never used
    q[0] = v.x;
    q[1] = K > 1 ? v.y : 0.f;
    q[2] = K > 2 ? v.z : 0.f;
    q[3] = K > 3 ? v.w : 0.f;
  }
  {  // 64 x 36 slab = 9 elements per thread, all loads in flight before the first LDS store
    float wv[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int e = j * 256 + tid;
      const int n = e / 36, kp = e - n * 36, tap = kp >> 2, ch = kp & 3;
      const bool ok = n < nvalid && ch < K;
      const float* src = dgrad ? d.w + ((int64_t)ch * d.w_cin + n0 + n) * 9 + (8 - tap)
                               : d.w + ((int64_t)(n0 + n) * d.w_cin + ch) * 9 + tap;
      wv[j] = *(ok ? src : g_zero_page);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int e = j * 256 + tid;
      const int n = e / 36;
      lw[n * TK_WROW + (e - n * 36)] = wv[j];
    }
  }
  const int y = y0 + wave, x = x0 + l31;
  const bool pix_ok = y < H && x < W;
  const int64_t pix = pix_ok ? ((int64_t)b * H + y) * W + x : 0;
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const bool extra = d.res1 || d.res2 || d.accumulate;
  __syncthreads();

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 18; ++s) {
    const int tap = s >> 1;                    // k' = 2s + lh -> tap = k' / 4, ch = k' % 4
    const int chb = (s & 1) * 2;               // ch = chb + lh
    const float a = lin[((wave + tap / 3) * HALO_W + l31 + tap % 3) * TK_INS + chb + lh];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nt >= ntv) break;
      const float bv = lw[(nt * 32 + l31) * TK_WROW + 2 * s + lh];
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a, acc[nt], 0, 0, 0);
    }
  }
  // memory-bound kernel: registers are spent on resident waves (latency hiding), not on hoisted loads
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    if (nt < ntv) {
      EpiRegs R;
      epi_load(d, n0 + nt * 32, pix, pix_ok, lh, s_uni, extra, R);
      epi_store(d, acc[nt], n0 + nt * 32, pix, pix_ok, lh, tid, R);
    }
}

// (2) N <= 4 output channels (64 -> 3 forward, 3 <- 64 backward-data): v_mfma_f32_4x4x1_16b_f32, whose 16
//     independent 4x4 blocks are used as 4 output channels (rows) x 64 pixels (4 per block): lane = pixel,
//     the 4 accumulator registers = the 4 output channels, so the matrix pipe runs at 3/4 (N = 3) instead of
//     3/32 utilisation and every lane ends up holding exactly its own pixel.  Workgroup = 4 rows x 64 pixels,
//     32-channel chunks: input halo [6 x 66 px][32 + 4] in LDS (16-byte fragment reads, stride 36 floats:
//     conflict-free), weights [tap][k quad][n][4 k] read as 4-address broadcasts.
constexpr int TN_W = 64, TN_HW = TN_W + 2, TN_PIX = HALO_H * TN_HW;  // 396
constexpr int TN_CK = 32, TN_INS = TN_CK + 4;
constexpr int TN_F4 = (TN_PIX * (TN_CK / 4) + 255) / 256;            // 13 float4 per thread

template <bool MASK>
__global__ __launch_bounds__(256, 2) void conv3x3_thin_n_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ __attribute__((aligned(16))) float lin[TN_PIX * TN_INS];
  __shared__ __attribute__((aligned(16))) float lw[9 * (TN_CK / 4) * 4 * 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  int bid = blockIdx.x;
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TN_W, y0 = ty * TH;
  const int H = d.H, W = d.W, K = d.K, N = d.N;
  const bool dgrad = d.mode == NEOSR_CONV_DGRAD;
  const float* __restrict__ inb = d.in + (int64_t)b * H * W * d.in_cs;
  const float* __restrict__ mkb = MASK ? d.in_mask + (int64_t)b * H * W * d.mask_cs : nullptr;

  // staging slots: granule g = i*256 + tid -> pixel g / 8, channel quad g % 8
  int in_off[TN_F4], mk_off[TN_F4];
#pragma unroll
  for (int i = 0; i < TN_F4; ++i) {
    const int g = i * 256 + tid, p = g >> 3;
    in_off[i] = -1;
    mk_off[i] = 0;
    if (p < TN_PIX) {
      const int py = p / TN_HW, px = p - py * TN_HW;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        in_off[i] = (gy * W + gx) * d.in_cs + (g & 7) * 4;
        mk_off[i] = (gy * W + gx) * d.mask_cs + (g & 7) * 4;
      }
    }
  }

  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < K; c0 += TN_CK) {
    float4 rin[TN_F4], rmk[MASK ? TN_F4 : 1];
#pragma unroll
    for (int i = 0; i < TN_F4; ++i) {
      const bool ok = in_off[i] >= 0 && c0 + ((i * 256 + tid) & 7) * 4 < K;
      rin[i] = *reinterpret_cast<const float4*>(ok ? inb + in_off[i] + c0 : g_zero_page);
      if (MASK) rmk[i] = *reinterpret_cast<const float4*>(ok ? mkb + mk_off[i] + c0 : g_zero_page);
    }
    // weight slab of this chunk: lw[tap][kq][n][e] = W(n, c0 + 4 kq + e, tap)
    float wv[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = j * 256 + tid;
      wv[j] = 0.f;
      if (e < 9 * (TN_CK / 4) * 16) {
        const int ke = e & 3, n = (e >> 2) & 3, kq = (e >> 4) & 7, tap = e >> 7;
        const int k = c0 + kq * 4 + ke;
        if (n < N && k < K)
          wv[j] = dgrad ? d.w[((int64_t)k * d.w_cin + n) * 9 + (8 - tap)] : d.w[((int64_t)n * d.w_cin + k) * 9 + tap];
      }
    }
    __syncthreads();  // the previous chunk has been consumed
#pragma unroll
    for (int i = 0; i < TN_F4; ++i) {
      const int g = i * 256 + tid, p = g >> 3;
      if (p < TN_PIX) {
        float4 v = rin[i];
        if (MASK) {
          const float4 m = rmk[i];
          v.x = m.x > 0.f ? v.x : v.x * d.mask_slope;
          v.y = m.y > 0.f ? v.y : v.y * d.mask_slope;
          v.z = m.z > 0.f ? v.z : v.z * d.mask_slope;
          v.w = m.w > 0.f ? v.w : v.w * d.mask_slope;
        }
        *reinterpret_cast<float4*>(lin + p * TN_INS + (g & 7) * 4) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = j * 256 + tid;
      if (e < 9 * (TN_CK / 4) * 16) lw[e] = wv[j];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* xp = lin + ((wave + tap / 3) * TN_HW + lane + tap % 3) * TN_INS;
      const float* wq = lw + (tap * (TN_CK / 4) * 4 + (lane & 3)) * 4;
#pragma unroll
      for (int kq = 0; kq < TN_CK / 4; ++kq) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + kq * 4);
        const f32x4 wf = *reinterpret_cast<const f32x4*>(wq + kq * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[e], xv[e], acc[e], 0, 0, 0);
      }
    }
  }
  const f32x4 r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  const int y = y0 + wave, x = x0 + lane;
  if (y < H && x < W) {
    float s_uni = 1.f;
    if (d.act == ACT_LRELU) s_uni = d.slope;
    else if (d.act == ACT_RELU) s_uni = 0.f;
    float* op = d.out + (((int64_t)b * H + y) * W + x) * d.out_cs;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      if (n < N) {
        float t = r[n] + (d.bias ? d.bias[n] : 0.f);
        op[n] = t > 0.f ? t : t * s_uni;
      }
    }
  }
}


}  // namespace

void neosr_conv::launch_thin_k(const ConvArgs& a, dim3 grid, hipStream_t st) {
  hipLaunchKernelGGL(conv3x3_thin_k_kernel, grid, dim3(256), 0, st, a);
}

void neosr_conv::launch_thin_n(const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_x = ceil_div(a.d.W, TN_W);
  dim3 grid(a.tiles_x * a.tiles_y * a.d.B, 1);
  if (a.d.in_mask) hipLaunchKernelGGL(conv3x3_thin_n_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv3x3_thin_n_kernel<false>, grid, dim3(256), 0, st, a);
}
