// pair_probe.hip — VERDICT r5 #1, step A: would a CU-PAIR split of the reduction (two 16x16 tiles per workgroup, half of
// every layer's input channels, U bytes per CU halved) make the bf16x3 form of the F(4x4,3x3) chain loop pay at B = 16?
//
// Two questions, each with the kill criterion VERDICT r5 states:
//   (1) the "chunk-equivalent" (the matrix work of one 32-channel chunk of today's loop = 2 tiles x 16 channels) at the
//       chain kernel's geometry — 768 threads per CU, 3 waves per SIMD, 153 KB of LDS so that one workgroup owns a CU —
//       with EVERYTHING a wave of that loop issues per chunk: U loads from an L2-resident image, the raw-tile LDS-DMA
//       (45 KB per chunk and CU), the ds_read_b128 of the column pass, the vector work of transform + split, the MFMAs,
//       one barrier.  Kill: not <= 2.3 us.
//       Wave mapping priced here ("P"): wave = (transform row ti, half row) -> 3 positions x 32 Winograd tiles x 32
//       output channels = 48 accumulator registers (the only 12-wave mapping with M = 32 per U register that fits 168
//       registers; (ti, channel parity) with M = 32 needs 96 accumulators).  Its price: the column pass runs over 5 of
//       the 6 patch columns in BOTH waves of a row and over both tiles: 2 x 5 x 3.67 = 37 ds_read_b128 per wave and
//       chunk-equivalent against 22 today, ~320 vector instructions against ~150.
//       The same kernel with today's proportions (f32: 12 loads, 48 f32 MFMA, 150 vector, 22 reads; fast tier: 12
//       loads, 24 bf16 MFMA, 182 vector, 42 LDS) calibrates the probe against the REAL loop: 3.2 us and 2.75 us.
//   (2) the partial-sum exchange of the pair: 32 KB (N = 32) / 64 KB (N = 64) written through (sc1) to the partner's
//       inbox, drain, flag, partner polls, loads.  Serialised round trip per layer, and the same with ~2 us of matrix
//       work between the write and the read.  Kill: not <= 2.5 us exposed.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pair_probe.hip -o tools/micro/bin/pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int AUX_SC1 = 16;

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// per iteration (= chunk-equivalent) and wave:
//   NL4 / NL2: 16-byte / 8-byte U loads per lane; NM bf16 16x16x32 MFMAs; NF f32 16x16x4 MFMAs; NV vector instructions;
//   NR ds_read_b128 of the raw tile; DMA: the 45 KB raw chunk as 45 wave-level LDS-DMA pieces (waves 0-8 issue 4, 9-11 3);
//   the work is issued in TWO halves (reads -> vector -> MFMA, twice) as the real loops do (lo / hi triple, or tile a / b)
template <int NL4, int NL2, int NM, int NF, int NV, int NR, bool DMA>
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void chunk_kernel(const float* __restrict__ img, int img_bytes, const float* __restrict__ act, int act_bytes, float* out,
                  int iters, float seed) {
  __shared__ __attribute__((aligned(1024))) float raw[2][11520];   // two 45 KB raw buffers
  __shared__ __attribute__((aligned(1024))) float pad[16128];      // + 63 KB: one workgroup per CU, as the chain kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const auto ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, img_bytes, 0x00020000);
  const auto ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(act), 0, act_bytes, 0x00020000);
  if (tid < 256) pad[tid * 63] = seed;
  for (int i = tid; i < 2 * 11520; i += 768) (&raw[0][0])[i] = seed * (float)(i & 7);
  __syncthreads();
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  float v[24];
  for (int i = 0; i < 24; ++i) v[i] = seed + i + lane;
  float fa = seed + lane, fb = seed * 0.5f;
  constexpr int NLA = NL4 > 0 ? NL4 : 1, NLB = NL2 > 0 ? NL2 : 1;
  i32x4 u4[NLA];
  f32x2 u2[NLB];
  for (int i = 0; i < NLA; ++i) u4[i] = (i32x4){0, 0, 0, 0};
  for (int i = 0; i < NLB; ++i) u2[i] = (f32x2){0.f, 0.f};
  const int rows = img_bytes >> 10;
  int row = (wave * 977) % rows;
  // this workgroup's tile of the activation image: 45 KB pieces, a different one per iteration (L2 / HBM resident mix as
  // in the chain: the raw tile of a chunk is read by one CU, once per layer)
  const int act_tiles = act_bytes / 46080;
  int at = (int)blockIdx.x % act_tiles;
  f32x4 ld[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    if (DMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r == 3 && wave >= 9) break;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void)(&raw[cur ^ 1][0] + (r * 12 + wave) * 256), 16, lane * 16,
                                                 at * 46080 + (r * 12 + wave) * 1024, 0, AUX_SC1);
      }
      at += 256;
      if (at >= act_tiles) at -= act_tiles;
    }
    // U loads of this chunk (the real loops issue them one half ahead; the probe only needs them in flight beside the work)
#pragma unroll
    for (int i = 0; i < NL4; ++i) {
      u4[i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, lane * 16, row * 1024, 0));
      row += 12;
      if (row >= rows) row -= rows;
    }
#pragma unroll
    for (int i = 0; i < NL2; ++i) {
      u2[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ru, lane * 8, row * 1024, 0));
      row += 12;
      if (row >= rows) row -= rows;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // column-pass reads: conflict-free 16-byte reads (lane-contiguous), distinct 1 KB rows of the current raw buffer
      constexpr int R0 = NR / 2, R1 = NR - NR / 2;
      const int nr = h ? R1 : R0;
#pragma unroll
      for (int i = 0; i < (h ? R1 : R0); ++i) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&raw[cur][((wave * 3 + i + h * 19) % 45) * 256 + lane * 4]);
        ld[i & 3] = i < 4 ? d : ld[i & 3] + d;
      }
      (void)nr;
      __builtin_amdgcn_sched_barrier(0);
      fa += ld[0][0] * 1e-30f + ld[1][1] * 1e-30f + ld[2][2] * 1e-30f + ld[3][3] * 1e-30f;
#pragma unroll
      for (int i = 0; i < NV / 2 - 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i % 24]) : "v"(fa), "v"(fb));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM / 2; ++i) acc[i % 12] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i % 12], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NF / 2; ++i) acc[i % 12] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[i % 12], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NL4; ++i) a[i & 7] ^= (short)u4[i][i & 3];
#pragma unroll
    for (int i = 0; i < NL2; ++i) fb += u2[i][i & 1] * 1e-30f;
    fa += 1e-9f;
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float s = fa + fb;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 24; ++i) s += v[i];
  s += a[0] + a[3] + pad[(tid * 21) & 16127] + raw[0][tid];
  out[blockIdx.x * 768 + tid] = s;
}

template <int NL4, int NL2, int NM, int NF, int NV, int NR, bool DMA>
double run_chunk(const char* name, const float* img, int img_bytes, const float* act, int act_bytes, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  hipLaunchKernelGGL((chunk_kernel<NL4, NL2, NM, NF, NV, NR, DMA>), dim3(256), dim3(768), 0, 0, img, img_bytes, act, act_bytes,
                     out, 50, 1.f);
  HIP_OK(hipEventRecord(e0));
  hipLaunchKernelGGL((chunk_kernel<NL4, NL2, NM, NF, NV, NR, DMA>), dim3(256), dim3(768), 0, 0, img, img_bytes, act, act_bytes,
                     out, iters, 1.f);
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double ns = ms * 1e6 / iters;
  printf("%-78s %8.1f ns / chunk-equivalent\n", name, ns);
  fflush(stdout);
  return ns;
}

// ---- (2) the pair exchange.  Workgroup w and w ^ 1 are partners (one workgroup per CU: 153 KB of LDS).  Per round: every
// thread writes KB16 16-byte pieces of its "partial" to the PARTNER's inbox (sc1), drains, the workgroup publishes the round
// in its flag word, wave 0 polls the partner's flag, then everybody loads its own inbox (sc1) — and MM MFMAs per wave sit
// between the publish and the poll (work that does not depend on the exchange: the next layer's old chunks).
template <int PIECES, int MM>
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void xchg_kernel(float* inbox, unsigned* flags, float* out, int rounds, int epoch0, float seed) {
  __shared__ __attribute__((aligned(1024))) float pad[39168];   // 153 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int me = blockIdx.x, partner = me ^ 1;
  pad[tid] = seed;
  const int box_bytes = PIECES * 768 * 16;
  const auto rmine = __builtin_amdgcn_make_buffer_rsrc(inbox + (size_t)me * (box_bytes / 4), 0, box_bytes, 0x00020000);
  float* pbox = inbox + (size_t)partner * (box_bytes / 4);
  const i32x4 rpart = {__builtin_amdgcn_readfirstlane((int)(uintptr_t)pbox),
                       __builtin_amdgcn_readfirstlane((int)(((uintptr_t)pbox >> 32) & 0xffff)), box_bytes > 0 ? box_bytes : 16, 0x00020000};
  const auto rflag = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 4096, 0x00020000);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){seed, seed, seed, seed};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  f32x4 part = {seed, seed + 1, seed + 2, seed + 3};
  f32x4 sum = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r) {
    const unsigned epoch = (unsigned)(epoch0 + r + 1);
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int voff = (p * 768 + tid) * 16;
      asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" : : "v"(part), "v"(voff), "s"(rpart) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store((gu32*)flags + me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < MM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 7], 0, 0, 0);
    if (wave == 0) {
      unsigned spins = 0;
      while (true) {
        const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(rflag, partner * 4, 0, AUX_SC1);
        if (f >= epoch || ++spins > (1u << 22)) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const f32x4 g = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rmine, (p * 768 + tid) * 16, 0, AUX_SC1));
      sum += g;
    }
    part += sum * 1e-30f;
    // (the inbox is re-written next round only after the partner has consumed it: the partner's next write follows its own
    // read of ITS inbox, which follows my flag of this round — a two-round protocol would double-buffer; the probe's
    // payload values are not checked, only the timing)
  }
  float s = sum[0] + sum[1] + sum[2] + sum[3] + pad[(tid * 51) % 39168];
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  out[blockIdx.x * 768 + tid] = s;
}

static int g_epoch = 0;
template <int PIECES, int MM>
void run_xchg(const char* name, float* inbox, unsigned* flags, float* out) {
  const int rounds = 1000;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  hipLaunchKernelGGL((xchg_kernel<PIECES, MM>), dim3(256), dim3(768), 0, 0, inbox, flags, out, 20, g_epoch, 1.f);
  g_epoch += 20;
  HIP_OK(hipEventRecord(e0));
  hipLaunchKernelGGL((xchg_kernel<PIECES, MM>), dim3(256), dim3(768), 0, 0, inbox, flags, out, rounds, g_epoch, 1.f);
  g_epoch += rounds;
  HIP_OK(hipEventRecord(e1));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-78s %8.1f ns / round\n", name, ms * 1e6 / rounds);
  fflush(stdout);
}
int main() {
  const int img_bytes = 2654208;            // 36 x 192 x 64 x 6 B: the bf16x3 U image of an RDB conv5 (L2-resident, shared)
  const int act_bytes = 46080 * 256 * 5;    // 59 MB of raw-tile pieces: five chunks per CU in rotation
  float *img, *act, *out, *inbox;
  unsigned* flags;
  HIP_OK(hipMalloc(&img, img_bytes));
  HIP_OK(hipMemset(img, 0, img_bytes));
  HIP_OK(hipMalloc(&act, act_bytes));
  HIP_OK(hipMemset(act, 0, act_bytes));
  HIP_OK(hipMalloc(&out, 256 * 768 * 4));
  HIP_OK(hipMalloc(&inbox, (size_t)256 * 8 * 768 * 16));
  HIP_OK(hipMemset(inbox, 0, (size_t)256 * 8 * 768 * 16));
  HIP_OK(hipMalloc(&flags, 4096));
  HIP_OK(hipMemset(flags, 0, 4096));

  printf("== (1) chunk-equivalent, everything a wave issues per chunk (256 workgroups x 768 threads, one per CU)\n");
  //        NL4 NL2  NM  NF   NV  NR  DMA
  run_chunk<12, 0, 0, 48, 150, 22, true>("calib f32 today: 12 ld, 48 f32 MFMA, 150 vec, 22 rd, DMA  [real loop 3.2 us]", img, img_bytes, act, act_bytes, out);
  run_chunk<12, 0, 24, 0, 182, 22, true>("calib fast tier: 12 ld, 24 bf16 MFMA, 182 vec, 22 rd, DMA [real loop 2.75 us]", img, img_bytes, act, act_bytes, out);
  run_chunk<18, 0, 36, 0, 300, 22, true>("bf16x3, no pair split: 18 ld, 36 MFMA, 300 vec, 22 rd, DMA  [5.1: 2.96 us]", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 36, 0, 320, 37, true>("PAIR bf16x3 (P): 6x(16+8 B) ld, 36 MFMA, 320 vec, 37 rd, DMA", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 36, 0, 320, 37, false>("PAIR bf16x3 (P) without the raw DMA", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 36, 0, 320, 0, true>("PAIR bf16x3 (P) without the column-pass reads", img, img_bytes, act, act_bytes, out);
  run_chunk<0, 0, 36, 0, 320, 37, true>("PAIR bf16x3 (P) without the U loads", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 36, 0, 16, 37, true>("PAIR bf16x3 (P) without the vector work", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 0, 0, 320, 37, true>("PAIR bf16x3 (P) without the MFMAs", img, img_bytes, act, act_bytes, out);
  run_chunk<12, 0, 36, 0, 300, 37, true>("PAIR bf16x3, 8 B per weight ([wh,wm] + [wh,wl]): 12 ld, 36 MFMA, 300 vec, 37 rd", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 6, 36, 0, 260, 30, true>("PAIR bf16x3, optimistic: 260 vec, 30 rd", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 0, 24, 0, 220, 37, true>("PAIR fast tier (2-piece): 6 ld, 24 MFMA, 220 vec, 37 rd, DMA", img, img_bytes, act, act_bytes, out);
  run_chunk<6, 0, 0, 48, 170, 37, true>("PAIR f32 MFMA: 6 ld, 48 f32 MFMA, 170 vec, 37 rd, DMA", img, img_bytes, act, act_bytes, out);

  printf("== (2) pair exchange through L2 (sc1 stores -> drain -> flag -> poll -> sc1 loads), serialised rounds\n");
  run_xchg<0, 0>("flag handshake only (no payload)", inbox, flags, out);
  run_xchg<3, 0>("32 KB + 4 KB payload (N = 32: 3 x 16 B per thread = 36 KB)", inbox, flags, out);
  run_xchg<6, 0>("64 KB payload (N = 64: 6 x 16 B per thread = 72 KB)", inbox, flags, out);
  run_xchg<0, 54>("no payload, 54 bf16 MFMA per wave (~1.2 us) between publish and poll", inbox, flags, out);
  run_xchg<3, 54>("36 KB payload, 54 bf16 MFMA per wave between publish and poll", inbox, flags, out);
  run_xchg<3, 108>("36 KB payload, 108 bf16 MFMA per wave (~2.4 us) between publish and poll", inbox, flags, out);
  run_xchg<6, 108>("72 KB payload, 108 bf16 MFMA per wave between publish and poll", inbox, flags, out);
  return 0;
}
