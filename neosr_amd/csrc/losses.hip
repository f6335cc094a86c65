// losses.hip — the remaining pixel losses of neosr/losses/basic_loss.py for gfx950: MSELoss (:57-86,
// F.mse_loss) and HuberLoss (:89-127, F.huber_loss with `delta`), reduction = "mean".
// Same shape as the L1 kernels: fixed-order two-stage sum forward, one elementwise pass backward.
#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

constexpr int RED_BLOCKS = 1024;

__device__ __forceinline__ float term(float d, int kind, float delta) {
  if (kind == NEOSR_LOSS_MSE) return d * d;
  const float a = fabsf(d);
  return a < delta ? 0.5f * d * d : delta * (a - 0.5f * delta);
}
__device__ __forceinline__ float dterm(float d, int kind, float delta) {
  if (kind == NEOSR_LOSS_MSE) return 2.f * d;
  return fabsf(d) < delta ? d : (d > 0.f ? delta : -delta);
}

__global__ __launch_bounds__(256) void ploss_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int64_t n, int kind, float delta,
                                                            float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s += term(a[i] - b[i], kind, delta);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void ploss_finalize_kernel(const float* __restrict__ part, int np, float scale,
                                                             float* __restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ __launch_bounds__(256) void ploss_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ gout, int64_t n, int kind, float delta,
                                                        float scale, float* __restrict__ ga) {
  const float g = gout[0] * scale;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    ga[i] = g * dterm(a[i] - b[i], kind, delta);
}

inline int grid_for(int64_t n, int cap) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int neosr_pointwise_loss_fwd(const float* pred, const float* target, int64_t n, int32_t kind, float delta,
                                        float loss_weight, float* loss_out, float* workspace, void* stream) {
  NEOSR_CHECK(pred && target && loss_out && workspace && n > 0, "pointwise_loss_fwd: bad args");
  NEOSR_CHECK(kind == NEOSR_LOSS_MSE || (kind == NEOSR_LOSS_HUBER && delta > 0.f), "pointwise_loss_fwd: bad kind / delta");
  const int nb = grid_for(n, RED_BLOCKS);
  hipLaunchKernelGGL(ploss_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, target, n, kind, delta,
                     workspace);
  hipLaunchKernelGGL(ploss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, nb,
                     loss_weight / (float)n, loss_out);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_pointwise_loss_bwd(const float* pred, const float* target, const float* grad_out, int64_t n,
                                        int32_t kind, float delta, float loss_weight, float* grad_pred, void* stream) {
  NEOSR_CHECK(pred && target && grad_out && grad_pred && n > 0, "pointwise_loss_bwd: bad args");
  hipLaunchKernelGGL(ploss_bwd_kernel, dim3(grid_for(n, 8192)), dim3(256), 0, (hipStream_t)stream, pred, target, grad_out,
                     n, kind, delta, loss_weight / (float)n, grad_pred);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
