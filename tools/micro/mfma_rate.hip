// Issue rate of the bf16 MFMA shapes on gfx950: NW waves per SIMD, each a loop of NCHAIN independent accumulation chains.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NCHAIN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(threadIdx.x * 3 + e); }
  f32x16 c32[NCHAIN];
  f32x4 c16[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i) {
    for (int r = 0; r < 16; ++r) c32[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) c16[i][r] = 0.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) {
      if (SHAPE == 32) c32[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c32[i], 0, 0, 0);
      else c16[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c16[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NCHAIN; ++i) s += SHAPE == 32 ? c32[i][0] + c32[i][7] : c16[i][0] + c16[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NCHAIN>
void run(int wg_per_cu, const char* name) {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, NCHAIN><<<grid, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<SHAPE, NCHAIN><<<grid, 256>>>(out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * NCHAIN * wg_per_cu;   // 4 waves per WG = 1 per SIMD
  const double flops = (SHAPE == 32 ? 32768.0 : 16384.0) * iters * NCHAIN * 4.0 * grid;
  printf("%-28s chains %d, %d waves/SIMD: %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), %.0f TFLOP/s\n", name, NCHAIN, wg_per_cu,
         ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, flops / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  run<32, 1>(1, "32x32x16 bf16");
  run<32, 2>(1, "32x32x16 bf16");
  run<32, 4>(1, "32x32x16 bf16");
  run<32, 4>(2, "32x32x16 bf16");
  run<16, 1>(1, "16x16x32 bf16");
  run<16, 2>(1, "16x16x32 bf16");
  run<16, 4>(1, "16x16x32 bf16");
  run<16, 8>(1, "16x16x32 bf16");
  run<16, 8>(2, "16x16x32 bf16");
  return 0;
}
