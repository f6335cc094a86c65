// ub.hip — what a wave's vector instructions cost beside fp32 MFMAs on gfx950 ("the matrix passes and the
// vector instructions of all resident waves share one issue stream, time = sum").  Per loop iteration: 8 independent
// v_mfma_f32_32x32x2_f32 (8 accumulators, no dependency stalls) + NV scalar v_fma_f32 (or NV/2 v_pk_fma_f32), one or two
// waves per SIMD, registers only.  Prints ns per iteration and wave pair.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ub.hip -o /tmp/ub && /tmp/ub
// (Re-created in round 4: the round-3 copy lived in the untracked experiments/ directory.)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, bool PK, bool MFMA>
__global__ __launch_bounds__(256) void ub_kernel(float* out, int iters, float seed) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (i + r);
  float a = seed + threadIdx.x, b = seed * 2.f;
  float v[16];
  f32x2 p[8];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  for (int i = 0; i < 8; ++i) p[i] = (f32x2){seed + i, seed - i};
  for (int it = 0; it < iters; ++it) {
    if (MFMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    if (!PK) {
#pragma unroll
      for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(a), "v"(b));
    } else {
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
    }
    a += 1e-9f;   // operands change every iteration (the power-limited clock of live data)
  }
  float s = a;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool PK, bool MFMA>
void run(const char* name, int waves_per_simd, float* out) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;   // 4 waves per workgroup = one per SIMD; 1 or 2 workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((ub_kernel<NV, PK, MFMA>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((ub_kernel<NV, PK, MFMA>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD %d: %8.1f ns per iteration\n", name, waves_per_simd, ms * 1e6 / iters);
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<0, false, true>("8 MFMA", w, out);
    run<32, false, true>("8 MFMA + 32 v_fma_f32", w, out);
    run<64, false, true>("8 MFMA + 64 v_fma_f32", w, out);
    run<64, true, true>("8 MFMA + 32 v_pk_fma_f32", w, out);
    run<64, false, false>("64 v_fma_f32 alone", w, out);
    run<64, true, false>("32 v_pk_fma_f32 alone", w, out);
  }
  return 0;
}
