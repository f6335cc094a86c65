// cu_hog.hip — stand-in for an RCCL collective on a 1-GPU lease: `blocks` workgroups that each hold a CU's resources
// (256 threads, 32 KB of LDS, so a conv3x3_wino4_chain_kernel workgroup — 153 KB of LDS, the whole register file — cannot
// become resident beside one) for `usec` microseconds of wall clock, then leave.  tools/ddp_contention.py launches it on
// the exchange stream wherever GradSync would issue an all-reduce.  hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(256) void cu_hog_kernel(unsigned usec, unsigned* sink) {
  __shared__ unsigned pad[8192];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint64_t t0 = wall_clock64();             // 100 MHz constant clock
  const uint64_t ticks = (uint64_t)usec * 100;
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) {
    acc += pad[(threadIdx.x * 7 + acc) & 8191];
    __builtin_amdgcn_s_sleep(32);
  }
  if (acc == 0xffffffffu) sink[0] = acc;           // keeps the loop alive
}

extern "C" int cu_hog(int blocks, unsigned usec, void* sink, void* stream) {
  hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, usec, (unsigned*)sink);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
