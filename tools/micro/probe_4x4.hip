// Layout probe for v_mfma_f32_4x4x1_16b_f32: hypothesis  A: lane -> (block = lane/4, row i = lane%4),
// B: lane -> (block = lane/4, col j = lane%4), D[v] of lane -> (block = lane/4, row i = v, col j = lane%4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f + l, 1000.f + 7.f * l, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); float h[256];
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
    const int b = l / 4, j = l % 4;
    const float a = 1.f + (4 * b + v), bb = 1000.f + 7.f * (4 * b + j);
    if (h[l * 4 + v] != a * bb) { if (bad < 8) printf("lane %d v %d: got %g want %g\n", l, v, h[l * 4 + v], a * bb); ++bad; }
  }
  printf("4x4x1_16b layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return 0;
}
