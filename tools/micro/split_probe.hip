// split_probe.hip — the three rates that bound a bf16x3 ("split") Winograd trunk on gfx950:
//   (1) do a wave's vector instructions hide under bf16 MFMA passes (they do NOT under f32 MFMA passes: ub.hip)?
//   (2) how fast can every CU stream a SHARED weight image out of L2 (16-byte loads per lane, distinct rows per wave)?
//   (3) both together, in the proportions of one 32-channel chunk of the F(4x4,3x3) loop.
// One workgroup of 768 threads per CU (3 waves per SIMD, as conv3x3_wino4_kernel), registers + an L2-resident image only.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// per iteration and wave: NL 16-byte loads per lane (1 KB per wave each), NM bf16 MFMAs (16x16x32), NF f32 MFMAs (16x16x4),
// NV scalar v_fma_f32
template <int NL, int NM, int NF, int NV, int PH = 0>
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void probe_kernel(const float* __restrict__ img, int img_bytes, float* out, int iters, float seed) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, img_bytes, 0x00020000);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
  float fa = seed + lane, fb = seed * 0.5f;
  i32x4 u[NL > 0 ? NL : 1];
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) u[i] = (i32x4){0, 0, 0, 0};
  // every wave walks the image in its own order (as twelve waves each streaming their own rows of U): row = 1 KB
  const int rows = img_bytes >> 10;
  int row = (wave * 977 + (PH ? (int)blockIdx.x * PH : 0)) % rows;   // PH: per-workgroup phase (de-synchronises the CUs' walks)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      u[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, lane * 16, row * 1024, 0));
      row += 12;
      if (row >= rows) row -= rows;
    }
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i & 7], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 15]) : "v"(fa), "v"(fb));
    // the loaded registers feed the next iteration's operands (keeps the loads live, as MFMA operands would)
#pragma unroll
    for (int i = 0; i < NL; ++i) a[i & 7] ^= (short)u[i][i & 3];
    fa += 1e-9f;
  }
  float s = fa;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  s += a[0] + a[3];
  out[blockIdx.x * 768 + tid] = s;
}

// NM x (one MFMA + VPM v_fma_f32) in program order (sched_barrier pins it): does the vector work hide INSIDE the MFMA passes
// of its own wave when it is interleaved (MI355X_MICROARCH.md: <= 5 single-issue instructions per 32-cycle MFMA)?
template <int NM, int VPM, bool BF>
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void inter_kernel(float* out, int iters, float seed) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
  float fa = seed + lane, fb = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (BF) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
      else acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < VPM; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(i * VPM + j) & 15]) : "v"(fa), "v"(fb));
      __builtin_amdgcn_sched_barrier(0);
    }
    fa += 1e-9f;
  }
  float s = fa;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 768 + tid] = s;
}
template <int NM, int VPM, bool BF>
void run_inter(const char* name, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((inter_kernel<NM, VPM, BF>), dim3(256), dim3(768), 0, 0, out, 50, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((inter_kernel<NM, VPM, BF>), dim3(256), dim3(768), 0, 0, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %8.1f ns / iteration\n", name, ms * 1e6 / iters);
}

// the same stream as LDS-DMA (buffer_load ... lds, 1 KB per wave-instruction into a per-wave 8 KB ring)
template <int NL>
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void dma_kernel(const float* __restrict__ img, int img_bytes, float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) float ring[12 * 8 * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, img_bytes, 0x00020000);
  const int rows = img_bytes >> 10;
  int row = (wave * 977) % rows;
  typedef __attribute__((address_space(3))) void* lds_void;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_void)(ring + (wave * 8 + (i & 7)) * 256), 16, lane * 16, row * 1024, 0, 0);
      row += 12;
      if (row >= rows) row -= rows;
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
  }
  out[blockIdx.x * 768 + tid] = ring[tid];
}
template <int NL>
void run_dma(const char* name, const float* img, int img_bytes, float* out, int grid = 256) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((dma_kernel<NL>), dim3(grid), dim3(768), 0, 0, img, img_bytes, out, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL((dma_kernel<NL>), dim3(grid), dim3(768), 0, 0, img, img_bytes, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / iters;
  const double gbs_cu = NL * 12.0 * 1024.0 / ns;
  printf("%-46s %8.1f ns / iteration   %6.1f GB/s per CU  %6.2f TB/s over %d CUs\n", name, ns, gbs_cu, gbs_cu * grid / 1000, grid);
}

template <int NL, int NM, int NF, int NV, int PH = 0>
void run(const char* name, const float* img, int img_bytes, float* out, int grid = 256) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((probe_kernel<NL, NM, NF, NV, PH>), dim3(grid), dim3(768), 0, 0, img, img_bytes, out, 50, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe_kernel<NL, NM, NF, NV, PH>), dim3(grid), dim3(768), 0, 0, img, img_bytes, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / iters;
  const double gbs_cu = NL * 12.0 * 1024.0 / ns;   // bytes per ns per CU = GB/s per CU
  printf("%-46s %8.1f ns / iteration   %6.1f GB/s per CU  %6.2f TB/s over %d CUs\n", name, ns, gbs_cu, gbs_cu * grid / 1000, grid);
}

int main() {
  const int img_bytes = 2654208;   // 36 x 192 x 64 x 6 B: the bf16x3 U image of an RDB conv5
  float *img, *out;
  hipMalloc(&img, img_bytes);
  hipMemset(img, 0, img_bytes);
  hipMalloc(&out, 256 * 768 * 4);
  // one 32-channel chunk, 32 output channels, per wave: f32 today = 12 loads + 48 f32 MFMAs + ~150 vector instructions;
  // split = 18 loads + 36 bf16 MFMAs (6 products x 6 positions x 2 cout blocks / 2 chunks per K = 32 ... per chunk) + ~300
  run<0, 36, 0, 0>("36 bf16 MFMA", img, img_bytes, out);
  run<0, 36, 0, 150>("36 bf16 MFMA + 150 v_fma", img, img_bytes, out);
  run<0, 36, 0, 300>("36 bf16 MFMA + 300 v_fma", img, img_bytes, out);
  run<0, 0, 0, 300>("300 v_fma alone", img, img_bytes, out);
  run<0, 0, 48, 0>("48 f32 MFMA", img, img_bytes, out);
  run<0, 0, 48, 150>("48 f32 MFMA + 150 v_fma", img, img_bytes, out);
  run<12, 0, 0, 0>("12 loads alone", img, img_bytes, out);
  run<18, 0, 0, 0>("18 loads alone", img, img_bytes, out);
  run<12, 0, 48, 150>("f32 chunk: 12 loads + 48 f32 MFMA + 150 v_fma", img, img_bytes, out);
  run<18, 36, 0, 300>("split chunk: 18 loads + 36 bf16 MFMA + 300 v_fma", img, img_bytes, out);
  run<18, 36, 0, 200>("split chunk: 18 loads + 36 bf16 MFMA + 200 v_fma", img, img_bytes, out);
  run<12, 18, 0, 200>("2-piece chunk: 12 loads + 18 bf16 MFMA + 200 v_fma", img, img_bytes, out);
  run<18, 72, 0, 300>("split chunk N=64: 36 loads.. (18) + 72 MFMA + 300", img, img_bytes, out);
  run<18, 0, 0, 0, 131>("18 loads alone, per-CU phase 131 rows", img, img_bytes, out);
  run<18, 0, 0, 0, 7>("18 loads alone, per-CU phase 7 rows", img, img_bytes, out);
  run<18, 0, 0, 0, 1>("18 loads alone, per-CU phase 1 row", img, img_bytes, out);
  run<18, 36, 0, 300, 131>("split chunk, per-CU phase 131", img, img_bytes, out);
  run<12, 0, 48, 150, 131>("f32 chunk, per-CU phase 131", img, img_bytes, out);
  run_inter<36, 0, true>("interleaved: 36 x (bf16 MFMA)", out);
  run_inter<36, 2, true>("interleaved: 36 x (bf16 MFMA + 2 v_fma)", out);
  run_inter<36, 4, true>("interleaved: 36 x (bf16 MFMA + 4 v_fma)", out);
  run_inter<36, 8, true>("interleaved: 36 x (bf16 MFMA + 8 v_fma)", out);
  run_inter<48, 0, false>("interleaved: 48 x (f32 MFMA)", out);
  run_inter<48, 3, false>("interleaved: 48 x (f32 MFMA + 3 v_fma)", out);
  run_inter<48, 6, false>("interleaved: 48 x (f32 MFMA + 6 v_fma)", out);
  run<18, 0, 0, 0>("18 loads alone, 64 workgroups", img, img_bytes, out, 64);
  run<18, 0, 0, 0>("18 loads alone, 8 workgroups (1 per XCD)", img, img_bytes, out, 8);
  run_dma<18>("18 LDS-DMA loads alone", img, img_bytes, out);
  run_dma<18>("18 LDS-DMA loads alone, 64 workgroups", img, img_bytes, out, 64);
  return 0;
}
