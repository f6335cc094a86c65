// conv_glds.hip — direct-to-LDS 3x3 convolution for gfx950 (the RDB trunk's forward and gather-form
// backward-data launches, and every layer-level conv whose weights have a packed image) + the weight
// packing kernels (conv_pack.h).
#include <cstring>
#include "conv_common.h"
#include "conv_pack.h"

using namespace neosr_conv;

namespace {

// Direct-to-LDS variant (needs d.w_pack).  One workgroup = 4 rows x 32 pixels x 32 output channels;
// chunks of 16 reduction channels; two LDS buffers of 31 KB (input halo 13 KB + weight slab 18 KB) so
// two workgroups share a CU.  Per chunk a wave issues 7-8 global_load_lds_dwordx4 (no staging VGPRs,
// no ds_write pass) and the workgroup meets at ONE barrier.
//   input image : granule (16 B) index = p*4 + (kq ^ ((p >> 2) & 3)), p = halo pixel (6 x 34), kq =
//                 channel quad; the permutation is applied on the (per-lane) global address, so 4
//                 lanes still fetch one pixel's 64 contiguous bytes, and the 16 lanes of a
//                 ds_read_b128 phase (consecutive pixels, same kq) land in 16 different bank groups
//   weight image: [tap][kq][n 32] granules — already the order of neosr_conv3x3_pack_weights()
// Fragments: lanes lh = 0 read quad 2s, lanes lh = 1 quad 2s+1; MFMA e of step s multiplies channel
// 8s + 4lh + e on both operands.
constexpr int GL_IN_GRAN = 13 * 64;                    // 816 used
constexpr int GL_W_GRAN = 9 * 4 * 32;                  // 1152 = 18 wave loads
constexpr int GL_BUF = (GL_IN_GRAN + GL_W_GRAN) * 4;   // floats per buffer (31 744 B)

__device__ __forceinline__ void glds16(const float* src, float* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <bool S2D>
__global__ __launch_bounds__(256, 2) void conv3x3_glds_kernel(const ConvArgs args) {
  const neosr_conv_desc& d = args.d;
  __shared__ __attribute__((aligned(1024))) float lds[2 * GL_BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  TL_MARK(0);

  int bid = xcd_tile(blockIdx.x, gridDim.x, args.xcd);
  const int tx = bid % args.tiles_x;
  bid /= args.tiles_x;
  const int ty = bid % args.tiles_y;
  const int b = bid / args.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int n0 = blockIdx.y * 32;
  const int H = d.H, W = d.W, K = d.K;
  const int Hin = d.ups ? (H >> 1) : H, Win = d.ups ? (W >> 1) : W;
  const float* __restrict__ inb = d.in + (int64_t)b * Hin * Win * d.in_cs;
  const int nchunks = (K + CK - 1) / CK;
  const float* __restrict__ wp = d.w_pack + (int64_t)blockIdx.y * nchunks * (GL_W_GRAN * 4) + lane * 4;

  // input granule of this thread in wave-load i: g = i*256 + tid -> pixel i*64 + tid/4, slot tid&3;
  // the slot holds channel quad (tid & 3) ^ ((p >> 2) & 3), and (p >> 2) & 3 == (tid >> 4) & 3 for all i
  const int q4 = ((tid & 3) ^ ((tid >> 4) & 3)) << 2;
  int in_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pix = (tid >> 2) + i * 64;
    in_off[i] = -1;
    if (pix < IN_PIX) {
      const int py = pix / HALO_W, px = pix - py * HALO_W;
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        const int sy = d.ups ? (gy >> 1) : gy, sx = d.ups ? (gx >> 1) : gx;
        in_off[i] = (sy * Win + sx) * d.in_cs + q4;
      }
    }
  }

  auto issue = [&](int c, int buf) {
    float* ibuf = lds + buf * GL_BUF;
    float* wbuf = ibuf + GL_IN_GRAN * 4;
    const int c0 = c * CK;
    const bool kq_ok = c0 + q4 < K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 3 && wave != 0) break;
      const float* src = (in_off[i] >= 0 && kq_ok) ? inb + in_off[i] + c0 : g_zero_page;
      glds16(src, ibuf + (i * 4 + wave) * 256);
    }
    const float* ws = wp + (int64_t)c * (GL_W_GRAN * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int j = (3 - wave) + 4 * i;  // 18 slab loads dealt so that every wave issues 7-8 in total
      if (j < 18) glds16(ws + j * 256, wbuf + j * 256);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const bool dg = d.mode == NEOSR_CONV_DGRAD;
  const int s2d_dgrad_mask = (d.s2d_c > 0 && dg && d.s2d_c % 32 == 0) ? s2d_tap_mask(n0 / d.s2d_c, true) : 0x1ff;
  // S2D = false keeps the tap loop branch-free (the compiler software-pipelines the LDS reads across
  // taps); s2d_c launches pay a wave-uniform branch per tap
  auto compute = [&](int buf, int tapmask) {
    const float* ibuf = lds + buf * GL_BUF;
    const float* wbuf = ibuf + GL_IN_GRAN * 4 + (lh * 32 + l31) * 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (S2D && !((tapmask >> tap) & 1)) continue;  // wave-uniform
      const int p = (wave + tap / 3) * HALO_W + l31 + tap % 3;
      const int sw = (p >> 2) & 3;
      const float* ap = ibuf + p * 16;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap + ((lh ^ sw) << 2));
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(ap + (((2 + lh) ^ sw) << 2));
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(wbuf + (tap * 4) * 128);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(wbuf + (tap * 4 + 2) * 128);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[e], a0[e], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[e], a1[e], acc, 0, 0, 0);
    }
  };

  auto chunk_mask = [&](int c) {
    if (d.s2d_c > 0 && !dg && d.s2d_c % CK == 0) return s2d_tap_mask((c * CK) / d.s2d_c, false);
    return s2d_dgrad_mask;
  };
  issue(0, 0);
  const int y = y0 + wave, x = x0 + l31;
  const bool pix_ok = y < H && x < W;
  const int64_t pix = pix_ok ? ((int64_t)b * H + y) * W + x : 0;
  float s_uni = 1.f;
  if (d.act == ACT_LRELU) s_uni = d.slope;
  else if (d.act == ACT_RELU) s_uni = 0.f;
  const bool extra = d.res1 || d.res2 || d.accumulate;
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  TL_MARK(1);
  for (int c = 0; c + 1 < nchunks; ++c) {
    issue(c + 1, (c + 1) & 1);
    TL_MARK(2 + c * 4);
    compute(c & 1, chunk_mask(c));
    TL_MARK(3 + c * 4);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // chunk c+1 has landed ...
    __syncthreads();                      // ... for every wave, and buffer c&1 is free again
    TL_MARK(4 + c * 4);
  }
  EpiRegs R;
  epi_load(d, n0, pix, pix_ok, lh, s_uni, extra, R);  // in flight under the last 72 MFMAs
  compute((nchunks - 1) & 1, chunk_mask(nchunks - 1));
  TL_MARK(62);
  epi_store(d, acc, n0, pix, pix_ok, lh, tid, R);
  TL_MARK(63);
}


// weight repack (see conv_pack.h): one thread per 16-byte granule of the destination image
__global__ __launch_bounds__(256) void conv_pack_kernel(const neosr_pack::Batch batch) {
  const neosr_pack::Image& im = batch.im[blockIdx.y];
  const int nch = (im.K + 15) >> 4, nblk = (im.N + 31) >> 5;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nblk * nch * GL_W_GRAN) return;
  const int n32 = g & 31, kq = (g >> 5) & 3;
  int rest = g >> 7;
  const int tap = rest % 9;
  rest /= 9;
  const int chunk = rest % nch, nb = rest / nch;
  const int n = nb * 32 + n32, k0 = chunk * 16 + kq * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < im.N && k0 < im.K) {
    for (int s = 0; s < im.nseg; ++s) {
      const neosr_pack::Seg& sg = im.seg[s];
      if (k0 < sg.k_lo || k0 >= sg.k_lo + sg.k_cnt) continue;
      const int kk = k0 - sg.k_lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kk + e >= sg.k_cnt) break;
        v[e] = im.mode == NEOSR_CONV_FWD
                   ? sg.w[((int64_t)(sg.n_lo + n) * sg.w_cin + kk + e) * 9 + tap]
                   : sg.w[((int64_t)(kk + e) * sg.w_cin + sg.n_lo + n) * 9 + (8 - tap)];
      }
    }
  }
  *reinterpret_cast<float4*>(im.dst + (int64_t)g * 4) = make_float4(v[0], v[1], v[2], v[3]);
}


}  // namespace

void neosr_conv::launch_glds(const ConvArgs& a, dim3 grid, hipStream_t st) {
  if (a.d.s2d_c > 0) hipLaunchKernelGGL(conv3x3_glds_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(conv3x3_glds_kernel<false>, grid, dim3(256), 0, st, a);
}

int neosr_pack::launch(const Image* images, int n, void* stream) {
  NEOSR_CHECK(images && n > 0, "conv pack: bad arguments");
  for (int i0 = 0; i0 < n; i0 += BATCH) {
    const int cnt = n - i0 < BATCH ? n - i0 : BATCH;
    Batch bt;
    memset(&bt, 0, sizeof(bt));
    int64_t gran = 0;
    for (int i = 0; i < cnt; ++i) {
      bt.im[i] = images[i0 + i];
      const int64_t g = image_floats(bt.im[i].N, bt.im[i].K) / 4;
      gran = g > gran ? g : gran;
    }
    dim3 grid((unsigned)((gran + 255) / 256), cnt);
    hipLaunchKernelGGL(conv_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, bt);
  }
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t neosr_conv3x3_pack_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return -1;
  return neosr_pack::image_floats(N, K) * 4;
}

extern "C" int neosr_conv3x3_pack_weights(const float* w, int32_t w_cout, int32_t w_cin, int32_t mode,
                                          float* dst, void* stream) {
  NEOSR_CHECK(w && dst && w_cout > 0 && w_cin > 0, "conv3x3_pack_weights: bad arguments");
  NEOSR_CHECK(mode == NEOSR_CONV_FWD || mode == NEOSR_CONV_DGRAD, "conv3x3_pack_weights: bad mode");
  NEOSR_CHECK((uintptr_t)dst % 16 == 0, "conv3x3_pack_weights: dst must be 16-byte aligned");
  neosr_pack::Image im;
  memset(&im, 0, sizeof(im));
  im.dst = dst;
  im.mode = mode;
  im.N = mode == NEOSR_CONV_FWD ? w_cout : w_cin;
  im.K = mode == NEOSR_CONV_FWD ? w_cin : w_cout;
  im.nseg = 1;
  im.seg[0].w = w;
  im.seg[0].w_cin = w_cin;
  im.seg[0].k_lo = 0;
  im.seg[0].k_cnt = im.K;
  im.seg[0].n_lo = 0;
  return neosr_pack::launch(&im, 1, stream);
}

