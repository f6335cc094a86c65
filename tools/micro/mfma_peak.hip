// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip with no memory traffic: what the conv / GEMM
// kernels can at best approach.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256, 2) void peak(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same, but the operands change every iteration (data toggling costs power -> clock)
template <int CHAINS>
__global__ __launch_bounds__(256, 2) void peak_toggle(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 0.37f, b = b0 - threadIdx.x * 0.11f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
      a = a * -1.0003f + 0.001f;
      b = b * -0.9997f - 0.002f;
    }
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS>
void run_toggle(int wgs, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak_toggle<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 2 * (double)CHAINS * iters * 4 * wgs;
    printf("toggling operands chains=%d wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", CHAINS, wgs, iters, ms, fl / ms / 1e9);
  }
}

template <int CHAINS>
void run(int wgs, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(peak<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak<CHAINS>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 2 * (double)CHAINS * iters * 4 * wgs;
    printf("chains=%d wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", CHAINS, wgs, iters, ms, fl / ms / 1e9);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  run<1>(512, 20000, out);    // one dependent chain per wave, 2 waves / SIMD
  run<4>(512, 5000, out);     // four independent chains
  run<4>(256, 5000, out);     // 1 wave / SIMD
  run<4>(512, 100000, out);   // ~0.5 s: thermally settled clock
  run_toggle<4>(512, 100000, out);
  run_toggle<1>(512, 400000, out);
  return 0;
}
