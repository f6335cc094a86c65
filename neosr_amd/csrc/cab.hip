// cab.hip — the non-conv parts of HAT's Channel Attention Block for gfx950
// (neosr/archs/hat_arch.py:15-52: conv3x3 - GELU - conv3x3 - ChannelAttention): erf-form GELU (A&S 7.1.26 erf, |err| <= 1.5e-7),
// global average pool / squeeze-excite MLP / sigmoid gate, and the gated residual combine of
// HAB.forward (hat_arch.py:347: x = shortcut + drop_path(attn_x) + conv_x * conv_scale).
// All HBM-bound; activations are channels-last (B, H*W, C).
#include "common.h"
#include "gelu.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t n, int cap = 4096) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// (GELU / GELU': gelu.h, shared with the F(4x4,3x3) convolution epilogue that runs them fused)

// VEC: 16-byte accesses (n a multiple of 4, pointers 16-byte aligned)
template <bool VEC>
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                   float* __restrict__ out, int64_t n) {
  if (VEC) {
    const int64_t nq = n >> 2;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(x)[q];
      float4 o;
      if (g) {
        const float4 u = reinterpret_cast<const float4*>(g)[q];
        o = make_float4(u.x * gelu_d(v.x), u.y * gelu_d(v.y), u.z * gelu_d(v.z), u.w * gelu_d(v.w));
      } else {
        o = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
      }
      reinterpret_cast<float4*>(out)[q] = o;
    }
    return;
  }
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = g ? g[e] * gelu_d(x[e]) : gelu_f(x[e]);
}

// part[b][slab][c] = sum over the slab's rows of x[b][r][c] (* y[b][r][c]); 64 columns x 4 row lanes
constexpr int SLABS = 128;
__global__ __launch_bounds__(256) void bcolsum_stage1(const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ part, int rows, int cols) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx, slab = blockIdx.y, b = blockIdx.z;
  const int rpb = (rows + SLABS - 1) / SLABS;
  const int r0 = slab * rpb, r1 = min(rows, r0 + rpb);
  float s = 0.f;
  if (c < cols) {
    const int64_t base = (int64_t)b * rows * cols + c;
    // (unrolled: eight rows' loads in flight per thread instead of one dependent load per trip — the pass was latency-bound
    // at 15 us for 3 MB; same additions in the same order)
#pragma unroll 8
    for (int r = r0 + ty; r < r1; r += 4) {
      const float v = x[base + (int64_t)r * cols];
      s += y ? v * y[base + (int64_t)r * cols] : v;
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < cols)
    part[((int64_t)b * SLABS + slab) * cols + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}
// the same partials for cols % 4 == 0, cols <= 1024 (the channels-last activations: cols = 180): one workgroup covers ALL
// columns of its slab — lane (quad qx = tid % (cols / 4), row lane ry = tid / (cols / 4)) reads 16 bytes, the 256 / (cols / 4)
// row lanes together read whole consecutive rows (contiguous memory); the row lanes are combined through LDS in order.
// (The 64-column form above read 256 bytes per wave and row and left the pass at 0.7 TB/s: 16.7 us for 11.8 MB.)
__global__ __launch_bounds__(256) void bcolsum_stage1_vec(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ part, int rows, int cols) {
  __shared__ float4 red[256];
  const int c4 = cols >> 2, rp = 256 / c4;
  const int tid = threadIdx.x, qx = tid % c4, ry = tid / c4;
  const int slab = blockIdx.y, b = blockIdx.z;
  const int rpb = (rows + SLABS - 1) / SLABS;
  const int r0 = slab * rpb, r1 = min(rows, r0 + rpb);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ry < rp) {
    const int64_t base = (int64_t)b * rows * cols + 4 * qx;
#pragma unroll 4
    for (int r = r0 + ry; r < r1; r += rp) {
      const float4 v = *reinterpret_cast<const float4*>(x + base + (int64_t)r * cols);
      if (y) {
        const float4 w = *reinterpret_cast<const float4*>(y + base + (int64_t)r * cols);
        s.x += v.x * w.x; s.y += v.y * w.y; s.z += v.z * w.z; s.w += v.w * w.w;
      } else {
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
  }
  red[tid] = s;
  __syncthreads();
  if (tid < c4) {
    float4 t = red[tid];
    for (int k = 1; k < rp; ++k) {
      const float4 u = red[k * c4 + tid];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + ((int64_t)b * SLABS + slab) * cols + 4 * tid) = t;
  }
}
__global__ __launch_bounds__(256) void bcolsum_stage2(const float* __restrict__ part, float* __restrict__ out,
                                                      int cols, int n, float scale) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // b * cols + c
  if (e >= n) return;
  const int b = e / cols, c = e - b * cols;
  // (sixteen partials requested at a time, added in slab order: one dependent load per trip left this pass latency-bound)
  float s = 0.f;
  for (int k0 = 0; k0 < SLABS; k0 += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = part[((int64_t)b * SLABS + k0 + k) * cols + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
  out[e] = s * scale;
}

constexpr int CS_MAX = 16;

// one workgroup per sample: hidden = relu(W1 pooled + b1) (Cs dot products by wave reduction),
// attn = sigmoid(W2 hidden + b2)
__global__ __launch_bounds__(256) void chan_attn_fwd_kernel(const float* __restrict__ pooled,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ hidden,
                                                            float* __restrict__ attn, int C, int Cs) {
  __shared__ float h[CS_MAX];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* p = pooled + (int64_t)b * C;
  for (int j = wave; j < Cs; j += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w1[j * C + c] * p[c];
    s = wave_reduce_sum(s);
    if (lane == 0) {
      const float v = fmaxf(s + b1[j], 0.f);
      h[j] = v;
      hidden[b * Cs + j] = v;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = b2[c];
    for (int j = 0; j < Cs; ++j) s += w2[c * Cs + j] * h[j];
    attn[(int64_t)b * C + c] = 1.f / (1.f + expf(-s));
  }
}

// single workgroup, samples in order (fixed summation order): parameter gradients + d pooled
__global__ __launch_bounds__(256) void chan_attn_bwd_kernel(const float* __restrict__ dattn,
                                                            const float* __restrict__ attn,
                                                            const float* __restrict__ hidden,
                                                            const float* __restrict__ pooled,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float* __restrict__ dpooled,
                                                            float* __restrict__ dw1, float* __restrict__ db1,
                                                            float* __restrict__ dw2, float* __restrict__ db2, int B,
                                                            int C, int Cs) {
  // Everything the sample loop reads is fetched ONCE in front of it — the gate inputs of all samples into LDS (B C <= 8 192
  // values: HAT trains with B = 4, C = 180), the weights a thread multiplies with into registers: the loop used to pay
  // three dependent global round trips per sample in a single workgroup (22 us for B = 4).  Same sums in the same order.
  constexpr int STAGE = 8192;
  __shared__ float dz2_all[STAGE], pooled_all[STAGE], hidden_all[64 * CS_MAX];
  __shared__ float dh[CS_MAX], db1_acc[CS_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool staged = B * C <= STAGE && B <= 64;
  // thread owns channels c = tid and tid + 256 (C <= 512)
  float gw2[2][CS_MAX], gb2[2] = {0.f, 0.f}, gw1[2][CS_MAX];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < CS_MAX; ++j) gw2[k][j] = gw1[k][j] = 0.f;
  if (tid < CS_MAX) db1_acc[tid] = 0.f;
  if (staged) {
    for (int i = tid; i < B * C; i += 256) {
      const float a = attn[i];
      dz2_all[i] = dattn[i] * a * (1.f - a);
      pooled_all[i] = pooled[i];
    }
    for (int i = tid; i < B * Cs; i += 256) hidden_all[i] = hidden[i];
  }
  // w2[c][j] for the (j = wave + 4 jj, c = lane + 64 kk) products of this lane, w1[j][c] for its two channels
  float w2r[CS_MAX / 4][8], w1r[2][CS_MAX];
#pragma unroll
  for (int jj = 0; jj < CS_MAX / 4; ++jj)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int j = wave + 4 * jj, c = lane + 64 * kk;
      w2r[jj][kk] = (j < Cs && c < C) ? w2[c * Cs + j] : 0.f;
    }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < CS_MAX; ++j) {
      const int c = tid + 256 * k;
      w1r[k][j] = (j < Cs && c < C) ? w1[j * C + c] : 0.f;
    }
  __shared__ float dz2_one[512];
  for (int b = 0; b < B; ++b) {
    __syncthreads();
    const float* dz2 = staged ? dz2_all + b * C : dz2_one;
    if (!staged) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
          const float a = attn[(int64_t)b * C + c];
          dz2_one[c] = dattn[(int64_t)b * C + c] * a * (1.f - a);
        }
      }
      __syncthreads();
    }
    // dh[j] = relu'(h) * sum_c dz2[c] w2[c][j]
#pragma unroll
    for (int jj = 0; jj < CS_MAX / 4; ++jj) {
      const int j = wave + 4 * jj;
      if (j < Cs) {
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int c = lane + 64 * kk;
          if (c < C) s += dz2[c] * w2r[jj][kk];
        }
        s = wave_reduce_sum(s);
        if (lane == 0) {
          const float h = staged ? hidden_all[b * Cs + j] : hidden[b * Cs + j];
          const float v = h > 0.f ? s : 0.f;
          dh[j] = v;
          db1_acc[j] += v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k;
      if (c < C) {
        const float z = dz2[c], pc = staged ? pooled_all[b * C + c] : pooled[(int64_t)b * C + c];
        gb2[k] += z;
        float dp = 0.f;
#pragma unroll
        for (int j = 0; j < CS_MAX; ++j)
          if (j < Cs) {
            gw2[k][j] += z * (staged ? hidden_all[b * Cs + j] : hidden[b * Cs + j]);
            gw1[k][j] += dh[j] * pc;
            dp += dh[j] * w1r[k][j];
          }
        dpooled[(int64_t)b * C + c] = dp;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = tid + 256 * k;
    if (c < C) {
      db2[c] = gb2[k];
#pragma unroll
      for (int j = 0; j < CS_MAX; ++j)
        if (j < Cs) {
          dw2[c * Cs + j] = gw2[k][j];
          dw1[j * C + c] = gw1[k][j];
        }
    }
  }
  if (tid < Cs) db1[tid] = db1_acc[tid];
}

// VEC: C a multiple of 4 and 16-byte aligned pointers: one quad of channels per thread, 32-bit index arithmetic (the scalar
// form spends two 64-bit divisions per element: 21 us for 2.9 million values)
template <bool VEC>
__global__ __launch_bounds__(256) void scale_channels_add_kernel(const float* __restrict__ y,
                                                                 const float* __restrict__ attn,
                                                                 const float* __restrict__ res,
                                                                 float* __restrict__ out, int64_t n, int rows, int C,
                                                                 float alpha) {
  if (VEC) {
    const int c4 = C >> 2;
    const unsigned per4 = (unsigned)rows * c4, nq = (unsigned)(n >> 2);
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < nq; q += gridDim.x * 256u) {
      const unsigned b = q / per4, c = (q % c4) * 4;
      const float4 v = reinterpret_cast<const float4*>(y)[q];
      const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)b * C + c);
      float4 o = make_float4(alpha * v.x * a.x, alpha * v.y * a.y, alpha * v.z * a.z, alpha * v.w * a.w);
      if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[q];
        o = make_float4(r.x + o.x, r.y + o.y, r.z + o.z, r.w + o.w);
      }
      reinterpret_cast<float4*>(out)[q] = o;
    }
    return;
  }
  const int64_t per = (int64_t)rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per), c = (int)(e % C);
    const float v = alpha * y[e] * attn[(int64_t)b * C + c];
    out[e] = res ? res[e] + v : v;
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void scale_channels_bwd_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ attn,
                                                                 const float* __restrict__ dpooled,
                                                                 float* __restrict__ dy, int64_t n, int rows, int C,
                                                                 float alpha, float inv_rows) {
  if (VEC) {
    const int c4 = C >> 2;
    const unsigned per4 = (unsigned)rows * c4, nq = (unsigned)(n >> 2);
    for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < nq; q += gridDim.x * 256u) {
      const unsigned b = q / per4, c = (q % c4) * 4;
      const float4 v = reinterpret_cast<const float4*>(g)[q];
      const float4 a = *reinterpret_cast<const float4*>(attn + (int64_t)b * C + c);
      const float4 p = *reinterpret_cast<const float4*>(dpooled + (int64_t)b * C + c);
      reinterpret_cast<float4*>(dy)[q] = make_float4(alpha * v.x * a.x + p.x * inv_rows, alpha * v.y * a.y + p.y * inv_rows,
                                                      alpha * v.z * a.z + p.z * inv_rows, alpha * v.w * a.w + p.w * inv_rows);
    }
    return;
  }
  const int64_t per = (int64_t)rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per), c = (int)(e % C);
    dy[e] = alpha * g[e] * attn[(int64_t)b * C + c] + dpooled[(int64_t)b * C + c] * inv_rows;
  }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int neosr_gelu(const float* x, const float* g, float* out, int64_t n, void* stream) {
  NEOSR_CHECK(x && out && n > 0, "gelu: bad args");
  if (n % 4 == 0 && al16(x) && al16(g) && al16(out))
    hipLaunchKernelGGL(gelu_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, g, out, n);
  else
    hipLaunchKernelGGL(gelu_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_batched_colsum(const float* x, const float* y, float* out, float* workspace, int32_t B,
                                    int32_t rows, int32_t cols, float scale, void* stream) {
  NEOSR_CHECK(x && out && workspace && B > 0 && rows > 0 && cols > 0, "batched_colsum: bad args");
  if (cols % 4 == 0 && cols <= 1024 && al16(x) && al16(y) && al16(workspace))
    hipLaunchKernelGGL(bcolsum_stage1_vec, dim3(1, SLABS, B), dim3(256), 0, (hipStream_t)stream, x, y, workspace, rows, cols);
  else
    hipLaunchKernelGGL(bcolsum_stage1, dim3(ceil_div(cols, 64), SLABS, B), dim3(256), 0, (hipStream_t)stream, x, y,
                       workspace, rows, cols);
  hipLaunchKernelGGL(bcolsum_stage2, dim3(ceil_div(B * cols, 256)), dim3(256), 0, (hipStream_t)stream, workspace, out,
                     cols, B * cols, scale);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_channel_attention_fwd(const float* pooled, const float* w1, const float* b1, const float* w2,
                                           const float* b2, float* hidden, float* attn, int32_t B, int32_t C,
                                           int32_t Cs, void* stream) {
  NEOSR_CHECK(pooled && w1 && b1 && w2 && b2 && hidden && attn && B > 0 && C > 0 && Cs > 0 && Cs <= CS_MAX,
              "channel_attention_fwd: bad args (squeezed channels <= 16)");
  hipLaunchKernelGGL(chan_attn_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, pooled, w1, b1, w2, b2, hidden,
                     attn, C, Cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_channel_attention_bwd(const float* dattn, const float* attn, const float* hidden,
                                           const float* pooled, const float* w1, const float* w2, float* dpooled,
                                           float* dw1, float* db1, float* dw2, float* db2, int32_t B, int32_t C,
                                           int32_t Cs, void* stream) {
  NEOSR_CHECK(dattn && attn && hidden && pooled && w1 && w2 && dpooled && dw1 && db1 && dw2 && db2 && B > 0 &&
                  C > 0 && C <= 512 && Cs > 0 && Cs <= CS_MAX,
              "channel_attention_bwd: bad args (C <= 512, squeezed channels <= 16)");
  hipLaunchKernelGGL(chan_attn_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dattn, attn, hidden, pooled, w1,
                     w2, dpooled, dw1, db1, dw2, db2, B, C, Cs);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_scale_channels_add(const float* y, const float* attn, const float* res, float* out, int32_t B,
                                        int32_t rows, int32_t C, float alpha, void* stream) {
  NEOSR_CHECK(y && attn && out && B > 0 && rows > 0 && C > 0, "scale_channels_add: bad args");
  const int64_t n = (int64_t)B * rows * C;
  if (C % 4 == 0 && n < (int64_t(1) << 32) && al16(y) && al16(attn) && al16(res) && al16(out))
    hipLaunchKernelGGL(scale_channels_add_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, y, attn, res,
                       out, n, rows, C, alpha);
  else
    hipLaunchKernelGGL(scale_channels_add_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, attn, res,
                       out, n, rows, C, alpha);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_scale_channels_bwd(const float* g, const float* attn, const float* dpooled, float* dy, int32_t B,
                                        int32_t rows, int32_t C, float alpha, void* stream) {
  NEOSR_CHECK(g && attn && dpooled && dy && B > 0 && rows > 0 && C > 0, "scale_channels_bwd: bad args");
  const int64_t n = (int64_t)B * rows * C;
  if (C % 4 == 0 && n < (int64_t(1) << 32) && al16(g) && al16(attn) && al16(dpooled) && al16(dy))
    hipLaunchKernelGGL(scale_channels_bwd_kernel<true>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, g, attn,
                       dpooled, dy, n, rows, C, alpha, 1.f / rows);
  else
    hipLaunchKernelGGL(scale_channels_bwd_kernel<false>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, attn,
                       dpooled, dy, n, rows, C, alpha, 1.f / rows);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
