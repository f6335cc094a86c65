// Shared device/host helpers for the neosr_amd gfx950 kernel library.
// Everything here targets CDNA4 (MI355X): 64-lane wavefronts, fp32-input MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NEOSR_WAVE 64

// activation ids shared by host and device (see include/neosr_amd.h)
enum : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_PRELU = 3, ACT_GELU = 4 };

// error plumbing (api.hip owns the storage)
void neosr_set_error(const char* fmt, ...);

#define NEOSR_CHECK(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      neosr_set_error(__VA_ARGS__);            \
      return 1;                                \
    }                                          \
  } while (0)

#define NEOSR_HIP(expr)                                                        \
  do {                                                                         \
    hipError_t e__ = (expr);                                                   \
    if (e__ != hipSuccess) {                                                   \
      neosr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                      __FILE__, __LINE__);                                     \
      return 2;                                                                \
    }                                                                          \
  } while (0)

#define NEOSR_LAUNCH_CHECK() NEOSR_HIP(hipGetLastError())

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
