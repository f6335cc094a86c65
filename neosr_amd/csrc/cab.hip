// cab.hip — the non-conv parts of HAT's Channel Attention Block for gfx950
// (neosr/archs/hat_arch.py:15-52: conv3x3 - GELU - conv3x3 - ChannelAttention): exact-erf GELU,
// global average pool / squeeze-excite MLP / sigmoid gate, and the gated residual combine of
// HAB.forward (hat_arch.py:347: x = shortcut + drop_path(attn_x) + conv_x * conv_scale).
// All HBM-bound; activations are channels-last (B, H*W, C).
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

inline int grid_for(int64_t n, int cap = 4096) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z);
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                   float* __restrict__ out, int64_t n) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = g ? g[e] * gelu_d(x[e]) : gelu_f(x[e]);
}

// part[b][slab][c] = sum over the slab's rows of x[b][r][c] (* y[b][r][c]); 64 columns x 4 row lanes
constexpr int SLABS = 32;
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
__global__ __launch_bounds__(256) void bcolsum_stage2(const float* __restrict__ part, float* __restrict__ out,
                                                      int cols, int n, float scale) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // b * cols + c
  if (e >= n) return;
  const int b = e / cols, c = e - b * cols;
  float s = 0.f;
  for (int k = 0; k < SLABS; ++k) s += part[((int64_t)b * SLABS + k) * cols + c];
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
  __shared__ float dz2[512], dh[CS_MAX], db1_acc[CS_MAX];
  __shared__ float red[4][CS_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // thread owns channels c = tid and tid + 256 (C <= 512)
  float gw2[2][CS_MAX], gb2[2] = {0.f, 0.f}, gw1[2][CS_MAX];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < CS_MAX; ++j) gw2[k][j] = gw1[k][j] = 0.f;
  if (tid < CS_MAX) db1_acc[tid] = 0.f;
  for (int b = 0; b < B; ++b) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k;
      if (c < C) {
        const float a = attn[(int64_t)b * C + c];
        dz2[c] = dattn[(int64_t)b * C + c] * a * (1.f - a);
      }
    }
    __syncthreads();
    // dh[j] = relu'(h) * sum_c dz2[c] w2[c][j]
    for (int j = wave; j < Cs; j += 4) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += dz2[c] * w2[c * Cs + j];
      s = wave_reduce_sum(s);
      if (lane == 0) {
        const float v = hidden[b * Cs + j] > 0.f ? s : 0.f;
        dh[j] = v;
        db1_acc[j] += v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k;
      if (c < C) {
        const float z = dz2[c], pc = pooled[(int64_t)b * C + c];
        gb2[k] += z;
        float dp = 0.f;
#pragma unroll
        for (int j = 0; j < CS_MAX; ++j)
          if (j < Cs) {
            gw2[k][j] += z * hidden[b * Cs + j];
            gw1[k][j] += dh[j] * pc;
            dp += dh[j] * w1[j * C + c];
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
  (void)red;
}

__global__ __launch_bounds__(256) void scale_channels_add_kernel(const float* __restrict__ y,
                                                                 const float* __restrict__ attn,
                                                                 const float* __restrict__ res,
                                                                 float* __restrict__ out, int64_t n, int rows, int C,
                                                                 float alpha) {
  const int64_t per = (int64_t)rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per), c = (int)(e % C);
    const float v = alpha * y[e] * attn[(int64_t)b * C + c];
    out[e] = res ? res[e] + v : v;
  }
}

__global__ __launch_bounds__(256) void scale_channels_bwd_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ attn,
                                                                 const float* __restrict__ dpooled,
                                                                 float* __restrict__ dy, int64_t n, int rows, int C,
                                                                 float alpha, float inv_rows) {
  const int64_t per = (int64_t)rows * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / per), c = (int)(e % C);
    dy[e] = alpha * g[e] * attn[(int64_t)b * C + c] + dpooled[(int64_t)b * C + c] * inv_rows;
  }
}

}  // namespace

extern "C" int neosr_gelu(const float* x, const float* g, float* out, int64_t n, void* stream) {
  NEOSR_CHECK(x && out && n > 0, "gelu: bad args");
  hipLaunchKernelGGL(gelu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g, out, n);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_batched_colsum(const float* x, const float* y, float* out, float* workspace, int32_t B,
                                    int32_t rows, int32_t cols, float scale, void* stream) {
  NEOSR_CHECK(x && out && workspace && B > 0 && rows > 0 && cols > 0, "batched_colsum: bad args");
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
  hipLaunchKernelGGL(scale_channels_add_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, attn, res, out,
                     n, rows, C, alpha);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_scale_channels_bwd(const float* g, const float* attn, const float* dpooled, float* dy, int32_t B,
                                        int32_t rows, int32_t C, float alpha, void* stream) {
  NEOSR_CHECK(g && attn && dpooled && dy && B > 0 && rows > 0 && C > 0, "scale_channels_bwd: bad args");
  const int64_t n = (int64_t)B * rows * C;
  hipLaunchKernelGGL(scale_channels_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, attn, dpooled,
                     dy, n, rows, C, alpha, 1.f / rows);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
