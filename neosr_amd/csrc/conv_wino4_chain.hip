// conv_wino4_chain.hip — a CHAIN of Winograd F(4x4,3x3) convolution layers in ONE launch (gfx950, fp32 MFMA).
//
// The RDB trunk of RRDBNet (neosr/archs/esrgan_arch.py:82-142: five 3x3 convolutions per residual dense block, each
// reading the concatenation of the block input and every earlier convolution's output) and its gather-form backward-data
// pass are sequences of DEPENDENT launches of conv3x3_wino4_kernel with one 16 x 16-pixel tile per CU at B = 16.  Per
// launch that kernel pays ~3 us of stream boundary, ~1.6 us of prologue (first DMA + weight round trip) and ~1 us of store
// drain against 8-17 us of matrix work.  This kernel runs a whole table of such layers (nets.hip: the fifteen
// convolutions of one RRDB) with the SAME arithmetic per layer (same transforms, same MFMA order: bit-identical
// results), one persistent workgroup per pixel tile:
//   * a layer's outputs are written through to memory (sc1 stores); every storing wave drains them, the workgroup
//     publishes "layer l done" in its tile's flag word (relaxed agent-scope store), and a consumer reads activations only
//     by sc1 `buffer_load ... lds` (L1 bypassed, MI355X_MICROARCH.md / cdna_hip_programming.md Guideline 16, form R1);
//   * inside a dense block only the NEWEST 32-channel slice of a layer's input was written by the previous layer; it is
//     the LAST chunk of the reduction.  The chunks in front of it are older than a whole layer, so the wait for the 3 x 3
//     neighbour tiles' flags sits two chunks ahead of the first dependent DMA (the poll load is issued at the start of
//     that iteration by wave 0 and looked at behind its MFMAs) and costs nothing when the neighbours keep pace; chunk 0
//     of the next layer is requested during the last chunk of the current one into a THIRD raw buffer, so the next
//     layer's prologue is hidden behind the current layer's tail;
//   * only a layer whose whole input is new (conv1 of the next dense block; `dep` = 0) waits in the open: drain ->
//     publish -> poll -> first DMA, about what the stream boundary it replaces costs;
//   * every spin is bounded (status word, see W4ChainArgs); the host launches at most one workgroup per CU and only
//     when the device has that many CUs, and zeroes the flag words in front of every pass.
// LDS: two exchange / raw objects of 54 KB as in conv3x3_wino4_kernel plus one 45 KB raw buffer for chunk 0 = 153 KB.
#include <cstddef>
#include <cstring>
#include <type_traits>
#include <vector>
#include <map>
#include <mutex>
#include <stdlib.h>
#include "conv_wino4.h"
#include "conv_wino4_chain.h"
#include "prof.h"

using namespace neosr_conv;

namespace {

typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int AUX_SC1 = 16;  // cache-policy bit of buffer loads / stores: sc1 (write-through store, L1-bypassing load)
// Bound of a flag wait, in polls (~1-2.5 us each once the back-off has grown: well over a minute).  A wait this long means a
// workgroup of the launch never became resident (the device has fewer free CUs than tiles for good); legitimate waits are
// microseconds, or — when another kernel holds CUs for a while, e.g. an RCCL collective that itself waits for a peer
// rank during warm-up — as long as that kernel runs.
constexpr unsigned SPIN_LIMIT = 1u << 25;
// A wait that has lasted this many polls (~64 short ones, then ~2.5-3 us each: about a millisecond) leaves a mark in
// status[1] — nothing is aborted, the results stay valid.  The models read the word through neosr_conv_chain_health with
// their loss scalars and, on data-parallel runs, agree through that all-reduce to leave the chain launches on EVERY rank
// at the same iteration (a collective that holds CUs for milliseconds costs a chain launch ~40 % of that time; one launch
// per convolution almost nothing: DESIGN §5) instead of one rank raising while the others sit in an all-reduce.
constexpr unsigned SLOW_SPINS = 400;

#ifdef NEOSR_TIMELINE
#define CTL_MARK(l, m)                                                                            \
  do {                                                                                            \
    if (args.timeline && !args.indep && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (l) < 16)   \
      args.timeline[((threadIdx.x >> 6) * 16 + (l)) * 16 + (m)] = clock64();                       \
  } while (0)
#else
#define CTL_MARK(l, m) do {} while (0)
#endif

typedef int w4c_i32x4 __attribute__((ext_vector_type(4)));
// 16-byte write-through (sc1) buffer store as inline assembly: hipcc's wait-count pass does not see it.  vmcnt counts loads
// and stores together and they complete out of order with each other, so behind a store it knows about the pass can only
// wait for "everything" at the next use of ANY loaded register — in a 64-channel layer the second unit's epilogue
// operands would wait for the first unit's write-through stores to drain (~2.7k cycles).  Hidden, its counted waits stay
// exact for loads (pending stores only make the hardware count larger: a wait can last longer than needed, never
// shorter); the one place that needs the stores themselves complete — the barrier in front of the flag — says
// s_waitcnt vmcnt(0) explicitly.  (s_nop: the store reads four data registers; the next instruction may overwrite them.)
__device__ __forceinline__ void store16_sc1(f32x4 v, w4c_i32x4 rsrc, int voff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(rsrc) : "memory");
}
// (Round 4 tried plain write-back stores for tiles whose 3 x 3 neighbourhood shares their XCD — all of them at B = 16,
// where an XCD's band of tiles is two whole samples: 1.2 % SLOWER (632 vs 640 LR-patches/s) and WRONG — the chain tests
// fail: the block -> XCD map is not a contract and a consumer on another L2 reads stale lines.  The stores stay sc1.)

// SPLIT: the fast_matmul tier (conv_wino4.h), as conv3x3_wino4_kernel<., true>.  (The body is a device function template
// behind two plain kernels: as a kernel TEMPLATE hipcc 7.2 failed the host-side substitution without a diagnostic.)
template <bool SPLIT>
__device__ __forceinline__ void chain_body(const W4ChainArgs& args) {
  __shared__ __attribute__((aligned(1024))) float ldsA[QBUF];
  __shared__ __attribute__((aligned(1024))) float ldsB[QBUF];
  __shared__ __attribute__((aligned(1024))) float ldsC[QSLOTS * 32];  // chunk 0 of every layer
  // The layer table, copied once from the kernel-argument segment: a record read from there costs a scalar-load round
  // trip of ~1.3 us (the segment is not served from a warm cache), two of them in a row per layer (the wave-mapping flag,
  // then the record) on the epilogue waves' path to the layer barrier.  From LDS a record is a few broadcast reads of one
  // uniform address + v_readfirstlane per field.
  __shared__ __attribute__((aligned(16))) int tabL[W4_MAX_LAYERS * 32];
  const int tid_k = threadIdx.x;
  const int tid = tid_k;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ti = wave >> 1, sel = wave & 1;   // transform row; channel parity (N64 layers: cout half)
  const int prio = wave >= 8 ? 2 : (wave >= 4 ? 1 : 0);  // waves w, w + 4, w + 8 share a SIMD: three static priorities

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 gtab = *reinterpret_cast<const i32x4*>(g_qt.gran[tid]);
  {
    static_assert(offsetof(W4ChainArgs, layers) == 0, "the table is read as the first bytes of the kernel arguments");
    static_assert(offsetof(W4Layer, out) == 48 && offsetof(W4Layer, K) == 56 && offsetof(W4Layer, act) == 88 &&
                      offsetof(W4Layer, n64) == 92 && offsetof(W4Layer, dep) == 96 && offsetof(W4Layer, slope) == 104 &&
                      offsetof(W4Layer, out_mask_slope) == 116,
                  "dword indices of the record reader below");
    typedef const __attribute__((address_space(4))) int* kptr_t;
    const kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    if (tid < W4_MAX_LAYERS * 32) tabL[tid] = ka[tid];
  }

  int bid = xcd_tile(blockIdx.x, gridDim.x, args.xcd);
  int tx, ty, b;
  if (args.tx_shift >= 0 && args.ty_shift >= 0) {
    tx = bid & (args.tiles_x - 1);
    ty = (bid >> args.tx_shift) & (args.tiles_y - 1);
    b = bid >> (args.tx_shift + args.ty_shift);
  } else {
    tx = bid % args.tiles_x;
    const int r = bid / args.tiles_x;
    ty = r % args.tiles_y;
    b = r / args.tiles_y;
  }
  const int x0 = tx * QT, y0 = ty * QT;
  const int H = args.H, W = args.W, in_cs = args.in_cs;

  // ---- DMA granules of this thread (the same for every layer: one geometry, one channel stride per chain)
  int in_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = gtab[r];
    const int y = e & 0xff, x = (e >> 8) & 0xff;
    const int gy = y0 + y - 1, gx = x0 + x - 1;
    const int q4 = (e >> 14) & 0x3fc;
    const bool ok = (e >> 24) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    in_off[r] = ok ? ((gy * W + gx) * in_cs + q4) * 4 : 0x7ffffff0;
  }
  auto make_rin = [&](const float* in, int K) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (int64_t)b * H * W * in_cs, 0,
                                             ((H * W - 1) * in_cs + K) * 4, 0x00020000);
  };
  typedef decltype(make_rin(nullptr, 0)) rsrc_t;
  auto issue = [&](rsrc_t rin, int c, float* buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r == 3 && wave >= 9) break;  // granules 2304 .. 2879: waves 0-8
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void)(buf + (r * 12 + wave) * 256), 16, in_off[r], c * 128, 0, AUX_SC1);
    }
  };

  // ---- flag words: own tile, and (wave 0, lanes 0-8) the 3 x 3 neighbourhood; lane 9 watches the status word
  const int my_flag = (b * args.tiles_y + ty) * args.tiles_x + tx;
  int nb_flag = -1;
  if (lane < 9) {
    const int ny = ty + lane / 3 - 1, nx = tx + lane % 3 - 1;
    if ((unsigned)ny < (unsigned)args.tiles_y && (unsigned)nx < (unsigned)args.tiles_x)
      nb_flag = (b * args.tiles_y + ny) * args.tiles_x + nx;
  }
  const auto rflag = __builtin_amdgcn_make_buffer_rsrc(args.flags, 0, 0x7ffffff0, 0x00020000);
  const auto rstat = __builtin_amdgcn_make_buffer_rsrc(args.status, 0, 4, 0x00020000);
  // (both words stay in their own registers until they are looked at: a select right here makes hipcc wait vmcnt(0) for the
  // two loads it has just issued — the ISA of rounds 3-5 had that wait at the START of the polling chunk, in front of wave
  // 0's MFMAs: one exposed L1-bypassing round trip per layer, which the other eleven waves then sat out at the barrier.
  // Round 6: 639.7 -> 645.5 LR-patches/s on the headline, same box, three runs each)
  struct PollWords { unsigned v, s; };
  auto poll_load = [&]() -> PollWords {  // (out-of-range offsets read 0)
    PollWords w;
    w.v = __builtin_amdgcn_raw_buffer_load_b32(rflag, nb_flag >= 0 ? nb_flag * 4 : 0x7ffffff0, 0, AUX_SC1);
    w.s = __builtin_amdgcn_raw_buffer_load_b32(rstat, lane == 9 ? 0 : 0x7ffffff0, 0, AUX_SC1);
    return w;
  };
  // every neighbour has finished `need` layers (or the chain was aborted by a timeout somewhere)
  auto poll_ok = [&](PollWords w, unsigned need) -> bool {
    const unsigned v = lane == 9 ? w.s : w.v;
    const bool ok = lane >= 9 || nb_flag < 0 || v >= need;
    const bool ab = lane == 9 && v != 0;
    return __builtin_amdgcn_ballot_w64(!ok) == 0 || __builtin_amdgcn_ballot_w64(ab) != 0;
  };
  auto poll_wait = [&](PollWords v, unsigned need) {
    if (!args.sync) return;
    unsigned spins = 0;
    while (!poll_ok(v, need)) {
      if (spins < 64) __builtin_amdgcn_s_sleep(4);   // back off: a long wait should not load the memory system
      else __builtin_amdgcn_s_sleep(64);
      v = poll_load();
      if (++spins == SLOW_SPINS && lane == 0)
        __hip_atomic_store((gu32*)args.status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (spins > SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store((gu32*)args.status, 1u + need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };

  // (the layer table is part of the kernel arguments: uniform reads of args.layers[l] are scalar loads from the
  // kernel-argument segment whatever the kernel has stored in between; only direct member reads — taking the array's
  // address makes hipcc copy all of it to scratch)
  const int nl = args.nlayers;
  uintptr_t nx_u = 0;      // first fields of the NEXT layer's record (see the layer start)
  int nx_K = 0, nx_N = 0, nx_n64 = 0;
  auto read_next = [&](int ln) {
    const int* rec = tabL + ln * 32;
    nx_u = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane(rec[2]) | ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane(rec[3]) << 32);
    nx_K = __builtin_amdgcn_readfirstlane(rec[14]);
    nx_N = __builtin_amdgcn_readfirstlane(rec[15]);
    nx_n64 = __builtin_amdgcn_readfirstlane(rec[23]);
  };
  bool c0_issued = false;   // chunk 0 of the layer about to start was requested by the previous layer

  auto layer = [&](auto n64tag, const int l) {
    constexpr bool N64 = decltype(n64tag)::value;
    CTL_MARK(l, 8);
    // weights pointer, K and N of this layer travel in scalar registers from the previous layer's tail (nx_*): the first
    // weight / table loads below leave before anything waits.  hipcc puts s_waitcnt vmcnt(0) in front of ANY LDS read
    // while an LDS-DMA is in flight (it cannot tell the raw buffers from tabL), which for the epilogue waves means
    // "until my output stores have drained" — with the loads already under way the two latencies overlap instead of adding.
    W4Layer L;
    L.u = (const float*)nx_u; L.K = nx_K; L.N = nx_N;
    const int K = L.K, nchunks = K >> 5;
    CTL_MARK(l, 9);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);

    // (opaque copy of the thread id: everything derived from it below is recomputed per layer — hoisted out of the layer
    // loop for both wave mappings it would occupy ~60 registers across the whole chain)
    int tid = tid_k;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int t16 = lane & 15, kq = lane >> 4;
    const int prow = N64 ? (wave & ~1) : wave;   // N64: the k-parity 0 row; parity 1 = the same offsets with bit 4 flipped
    const i32x4 pa_lo = *reinterpret_cast<const i32x4*>(g_qt.pa[prow][lane]);
    const i32x4 pa_hi = *reinterpret_cast<const i32x4*>(g_qt.pa[prow][lane] + 4);

    // ---- U image of this wave's 32-cout block: [chunk][pos 36][kp 2][cout block 2][k quad 4][cout 16][4] floats
    const int nblk = N64 ? sel : 0;
    const bool blk_ok = nblk * 32 < L.N;
    const auto ru = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(L.u) + (int64_t)(blk_ok ? nblk : 0) * nchunks * QU_CHUNK, 0, blk_ok ? nchunks * QU_CHUNK * 4 : 0,
        0x00020000);
    const int u_lane = lane * 16;
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(ru, 0, 0, 0)) u32x4_t;
    auto load_u3 = [&](int c, int kp, int j0, f32x4 (&u)[3][2]) {
      const int u_wave = ((ti * 6) * 4 + kp * 2) * 1024;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + nb * 1024, c * (QU_CHUNK * 4) + u_wave + (j0 + j) * 4096, 0);
          u[j][nb] = __builtin_bit_cast(f32x4, v);
        }
    };
    f32x4 ulo[3][2], uhi[3][2], vlo[3], vhi[3];
    CTL_MARK(l, 0);
    load_u3(0, N64 ? 0 : sel, 0, ulo);
    __builtin_amdgcn_sched_barrier(0);
    {
      const int* rec = tabL + l * 32;
      auto w = [&](int k) { return __builtin_amdgcn_readfirstlane(rec[k]); };
      auto ptr = [&](int k) { return (uintptr_t)(unsigned)w(k) | ((uintptr_t)(unsigned)w(k + 1) << 32); };
      L.in = (const float*)ptr(0); L.bias = (const float*)ptr(4); L.res1 = (const float*)ptr(6);
      L.res2 = (const float*)ptr(8); L.out_mask = (const float*)ptr(10); L.out = (float*)ptr(12);
      L.out_cs = w(16); L.res1_cs = w(17); L.res1_nch = w(18); L.res2_cs = w(19); L.res2_nch = w(20);
      L.out_mask_cs = w(21); L.act = w(22); L.dep = w(24);
      L.slope = __int_as_float(w(26)); L.alpha = __int_as_float(w(27)); L.alpha2 = __int_as_float(w(28));
      L.out_mask_slope = __int_as_float(w(29));
    }
    __builtin_amdgcn_sched_barrier(0);

    const rsrc_t rin = make_rin(L.in, K);
    // the layer's input holds channels written by the previous layer of THIS launch from chunk `dep` on (-1: none)
    const int dep = l > 0 ? (L.dep == 1 ? 0 : L.dep) : -1;
    if (l == 0 || (dep != 0 && !c0_issued)) issue(rin, 0, ldsC);   // (else: requested by the previous layer, or after the poll)
    // previous layer's stores drained (every storing wave), chunk 0 landed, then publish.  A table of INDEPENDENT layers
    // (args.indep: sample strips of one convolution) has nobody to tell: no drain, no flag — chunk 0 was requested during
    // the previous layer's last iteration, every wave's pieces had landed before it left that layer's last MFMAs (their
    // weights were loaded behind the request; the wait for them is vmcnt(0)) and its exchange barrier made them visible.
#ifdef CHAIN_NO_DRAIN   // timing probe only (round 6): the layer-start drain skipped behind layer 0 — RACY, wrong results
    if (l == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    if (!args.indep || l == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CTL_MARK(l, 1);
    __syncthreads();
    CTL_MARK(l, 2);
    if (!args.indep && l > 0 && tid == 0)
      __hip_atomic_store((gu32*)args.flags + my_flag, (unsigned)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dep == 0) {  // the whole input is new: wait in the open
      if (wave == 0) poll_wait(poll_load(), (unsigned)l);
      __syncthreads();
      issue(rin, 0, ldsC);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int pollc = dep >= 2 ? dep - 2 : -1;
    CTL_MARK(l, 3);

    const bool three = ti == 0 || ti == 5;
    const float ap = three ? -5.f : (ti <= 2 ? -4.f : -1.f);
    const float aq = ap;
    const float gm = three ? 4.f : (ti == 1 ? 1.f : ti == 2 ? -1.f : ti == 3 ? 2.f : -2.f);
    const int pa[4][2] = {{pa_lo[0], pa_lo[1]}, {pa_lo[2], pa_lo[3]}, {pa_hi[0], pa_hi[1]}, {pa_hi[2], pa_hi[3]}};

    f32x4 acc[6][2];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[j][nb] = splat(0.f);

    auto mac3 = [&](int j0, const f32x4 (&v)[3], const f32x4 (&u)[3][2]) {
      if constexpr (SPLIT) {
        bf16x8 bh[3], bl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) split_hi_lo(v[j], bh[j], bl[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[j0 + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u[j][nb]), bh[j], acc[j0 + j][nb], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[j0 + j][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u[j][nb]), bl[j], acc[j0 + j][nb], 0, 0, 0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[j0 + j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][nb][e], v[j][e], acc[j0 + j][nb], 0, 0, 0);
      }
    };
    auto transform = [&](const float* rb, int xr, f32x4 (&vlo)[3], f32x4 (&vhi)[3]) {
      f32x4 t[6];
      const f32x4 ap4 = splat(ap), aq4 = splat(aq), gm4 = splat(gm);
      if (three) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const int o = (c & 3) * 64;
          const f32x4 da = ld4f(rb + (pa[0][c >> 2] ^ xr) + o), db = ld4f(rb + (pa[1][c >> 2] ^ xr) + o);
          const f32x4 dc = ld4f(rb + (pa[3][c >> 2] ^ xr) + o);
          t[c] = fma4(gm4, da, fma4(ap4, db, dc));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          const int o = (c & 3) * 64;
          const f32x4 d1 = ld4f(rb + (pa[0][c >> 2] ^ xr) + o), d2 = ld4f(rb + (pa[1][c >> 2] ^ xr) + o);
          const f32x4 d3 = ld4f(rb + (pa[2][c >> 2] ^ xr) + o), d4 = ld4f(rb + (pa[3][c >> 2] ^ xr) + o);
          t[c] = fma4(gm4, fma4(aq4, d1, d3), fma4(ap4, d2, d4));
        }
      }
      const f32x4 m4 = splat(-4.f), m5 = splat(-5.f), p4 = splat(4.f), p2 = splat(2.f), m2 = splat(-2.f);
      const f32x4 a = fma4(m4, t[2], t[4]), bq = fma4(m4, t[1], t[3]);
      const f32x4 cc = t[4] - t[2], dd = t[3] - t[1];
      vlo[0] = fma4(p4, t[0], fma4(m5, t[2], t[4]));
      vlo[1] = a + bq;
      vlo[2] = a - bq;
      vhi[0] = fma4(p2, dd, cc);
      vhi[1] = fma4(m2, dd, cc);
      vhi[2] = fma4(p4, t[1], fma4(m5, t[3], t[5]));
    };

    // the next layer's chunk 0 may be requested during this layer's last chunk when it is older than this layer's output
    bool next_c0 = false;
    rsrc_t rin_next = rin;
    if (l + 1 < nl) {
      const int* nx = tabL + (l + 1) * 32;
      const int dn = __builtin_amdgcn_readfirstlane(nx[24]);
      next_c0 = dn != 0 && dn != 1;
      const uintptr_t nin = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane(nx[0]) |
                            ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane(nx[1]) << 32);
      rin_next = make_rin((const float*)nin, __builtin_amdgcn_readfirstlane(nx[14]));
    }

    for (int c = 0; c < nchunks; ++c) {
      const float* rb = c == 0 ? ldsC : ((c & 1) ? ldsA : ldsB);
      if (c < 6) CTL_MARK(l, 10 + c);
      PollWords fv = {0u, 0u};
      if (wave == 0 && c == pollc) fv = poll_load();
      if (c > 0) mac3(3, vhi, uhi);  // positions (ti, 3..5) of the previous (sub-)chunk
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < nchunks) issue(rin, c + 1, (c & 1) ? ldsB : ldsA);
      else if (next_c0) issue(rin_next, 0, ldsC);
      __builtin_amdgcn_sched_barrier(0);
      transform(rb, 0, vlo, vhi);
      __builtin_amdgcn_sched_barrier(0);
      load_u3(c, N64 ? 0 : sel, 3, uhi);
      __builtin_amdgcn_sched_barrier(0);
      mac3(0, vlo, ulo);
      __builtin_amdgcn_sched_barrier(0);
      if (N64) {  // second k-parity of the same raw chunk, no barrier in between
        load_u3(c, 1, 0, ulo);
        __builtin_amdgcn_sched_barrier(0);
        mac3(3, vhi, uhi);
        __builtin_amdgcn_sched_barrier(0);
        transform(rb, 16, vlo, vhi);
        __builtin_amdgcn_sched_barrier(0);
        load_u3(c, 1, 3, uhi);
        __builtin_amdgcn_sched_barrier(0);
        mac3(0, vlo, ulo);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (wave == 0 && c == pollc) poll_wait(fv, (unsigned)l);
      if (c + 1 < nchunks) {
        load_u3(c + 1, N64 ? 0 : sel, 0, ulo);
        if (N64) __builtin_amdgcn_s_waitcnt(0x4f78);  // vmcnt(24): chunk c + 1 has landed
        else __builtin_amdgcn_s_waitcnt(0x0f7c);      // vmcnt(12)
      }
      __syncthreads();
    }
    c0_issued = next_c0;
    CTL_MARK(l, 4);
    mac3(3, vhi, uhi);
    read_next(l + 1 < nl ? l + 1 : l);
    __builtin_amdgcn_s_setprio(0);

    // ---- epilogue (waves 4-11), as in conv3x3_wino4_kernel; stores are written through (sc1)
    const bool fin = wave >= 4;
    const int cq = N64 ? (tid & 15) << 2 : (tid & 7) << 2;
    const int eb = N64 ? (tid >> 4) & 3 : (tid >> 3) & 3;
    const int et0 = N64 ? ((tid - 256) >> 6) & 7 : ((tid - 256) >> 5) & 15;
    const int chq = cq;
    const bool ch_ok = fin && chq < L.N;
    f32x4 bias = splat(0.f);
    if (fin && L.bias) bias = ld4f(L.bias + (ch_ok ? chq : 0));
    float s_uni = 1.f;
    if (L.act == ACT_LRELU) s_uni = L.slope;
    else if (L.act == ACT_RELU) s_uni = 0.f;
    typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(ru, 0, 0, 0)) raw4_t;
    // (readfirstlane: the asm's "s" operand must be a scalar register tuple even where hipcc keeps the pointer in a VGPR)
    const w4c_i32x4 r_out = {__builtin_amdgcn_readfirstlane((int)(uintptr_t)L.out),
                             __builtin_amdgcn_readfirstlane((int)(((uintptr_t)L.out >> 32) & 0xffff)), 0x7ffffff0, 0x00020000};
    struct Epi {
      int o_out[4];
      f32x4 e1[4], e2[4], mk[4];
    };
    auto epi_load = [&](int et, Epi& E) {
      const int ey = y0 + 4 * (et >> 2), ex = x0 + 4 * (et & 3) + eb;
      const int pix0 = (b * H + ey) * W + ex;
      const bool col_ok = ch_ok && ex < W;
      bool okr[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) okr[a] = col_ok && ey + a < H;
      auto offs = [&](int cs, bool ok_ch, int (&o)[4]) {
        const int base = (pix0 * cs + chq) * 4, step = W * cs * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) o[a] = (okr[a] && ok_ch) ? base + a * step : 0x7ffffff8;
      };
      auto load4 = [&](const float* p, const int (&o)[4], f32x4 (&v)[4]) {
        const auto rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int a = 0; a < 4; ++a) v[a] = __builtin_bit_cast(f32x4, (raw4_t)__builtin_amdgcn_raw_buffer_load_b128(rr, o[a], 0, 0));
      };
      offs(L.out_cs, true, E.o_out);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        E.e1[a] = E.e2[a] = splat(0.f);
        E.mk[a] = splat(1.f);
      }
      int o[4];
      if (L.res1) {
        offs(L.res1_cs, chq < L.res1_nch, o);
        load4(L.res1, o, E.e1);
      }
      if (L.res2) {
        offs(L.res2_cs, chq < L.res2_nch, o);
        load4(L.res2, o, E.e2);
      }
      if (L.out_mask) {
        offs(L.out_mask_cs, true, o);
        load4(L.out_mask, o, E.mk);
      }
    };
    Epi E0;
    if (fin) epi_load(et0, E0);

    constexpr int ES = N64 ? 68 : QES;
    {
      float* ex_img = N64 ? (ti >= 3 ? ldsB : ldsA) : (sel ? ldsB : ldsA);
      const int er = N64 ? (ti >= 3 ? ti - 3 : ti) : ti;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const f32x4 s1 = acc[1][nb] + acc[2][nb], d1 = acc[1][nb] - acc[2][nb];
        const f32x4 s2 = acc[3][nb] + acc[4][nb], d2 = acc[3][nb] - acc[4][nb];
        const f32x4 x0v = (acc[0][nb] + s1) + s2;
        const f32x4 x1v = fma4(splat(2.f), d2, d1);
        const f32x4 x2v = fma4(splat(4.f), s2, s1);
        const f32x4 x3v = fma4(splat(8.f), d2, d1) + acc[5][nb];
        float* p = ex_img + ((er * 4) * 16 + t16) * ES + (N64 ? 32 * sel : 0) + 16 * nb + 4 * kq;
        *reinterpret_cast<f32x4*>(p) = x0v;
        *reinterpret_cast<f32x4*>(p + 16 * ES) = x1v;
        *reinterpret_cast<f32x4*>(p + 32 * ES) = x2v;
        *reinterpret_cast<f32x4*>(p + 48 * ES) = x3v;
      }
    }
    CTL_MARK(l, 5);
    __syncthreads();
    CTL_MARK(l, 6);
    if (!fin) return;

    auto epi_finish = [&](int et, Epi& E) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        asm volatile("" : "+v"(E.mk[a]));
        asm volatile("" : "+v"(E.e1[a]), "+v"(E.e2[a]));
      }
      f32x4 xi[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (N64) {
          xi[i] = ld4f((i >= 3 ? ldsB : ldsA) + (((i >= 3 ? i - 3 : i) * 4 + eb) * 16 + et) * ES + cq);
        } else {
          const int o = ((i * 4 + eb) * 16 + et) * ES + cq;
          xi[i] = ld4f(ldsA + o) + ld4f(ldsB + o);
        }
      }
      f32x4 y[4];
      {
        const f32x4 s1 = xi[1] + xi[2], d1 = xi[1] - xi[2], s2 = xi[3] + xi[4], d2 = xi[3] - xi[4];
        y[0] = (xi[0] + s1) + s2;
        y[1] = fma4(splat(2.f), d2, d1);
        y[2] = fma4(splat(4.f), s2, s1);
        y[3] = fma4(splat(8.f), d2, d1) + xi[5];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = y[a][e] + bias[e];
          t = t > 0.f ? t : t * s_uni;
          t = t * L.alpha + E.e1[a][e];
          t = t * L.alpha2 + E.e2[a][e];
          t += 0.f;   // (the `accumulate` term of the one-layer kernel: x + 0 keeps -0 -> +0 identical)
          o[e] = E.mk[a][e] > 0.f ? t : t * L.out_mask_slope;
        }
        store16_sc1(o, r_out, E.o_out[a]);
      }
    };
    if (N64) {  // second unit: its operands are requested before the first unit's arithmetic and stores
      Epi E1;
      epi_load(et0 + 8, E1);
      epi_finish(et0, E0);
      epi_finish(et0 + 8, E1);
    } else {
      epi_finish(et0, E0);
    }
    CTL_MARK(l, 7);
  };

  __syncthreads();   // tabL
  read_next(0);
  for (int l = 0; l < nl; ++l) {
    if (nx_n64) layer(std::true_type{}, l);
    else layer(std::false_type{}, l);
  }
}
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void conv3x3_wino4_chain_kernel(const W4ChainArgs args) { chain_body<false>(args); }
__global__ __attribute__((amdgpu_flat_work_group_size(768, 768), amdgpu_waves_per_eu(3, 3)))
void conv3x3_wino4_chain_split_kernel(const W4ChainArgs args) { chain_body<true>(args); }

// ---- host side -------------------------------------------------------------------------------------------------
std::mutex g_mu;
int g_chain_on = -1;                            // -1: read NEOSR_AMD_CHAIN on first use (default on)
int g_chain_sync = 1;                           // debug: 0 = flag waits skipped (timing only, racy)
bool g_chain_tripped = false;                   // neosr_conv_chain_status() has seen an aborted launch: no more chain launches

}  // namespace

extern "C" int neosr_set_conv_chain(int on) {
  const int prev = g_chain_on < 0 ? 1 : g_chain_on;
  g_chain_on = on ? 1 : 0;
  return prev;
}
extern "C" int neosr_set_conv_chain_sync(int mode) {
  const int prev = g_chain_sync;
  g_chain_sync = mode;
  return prev;
}

bool neosr_conv::chain_enabled() {
  if (g_chain_on < 0) {
    const char* e = getenv("NEOSR_AMD_CHAIN");
    g_chain_on = (e && e[0] == '0') ? 0 : 1;
  }
  return g_chain_on == 1 && !g_chain_tripped && wino_mode() == 2;
}

namespace {
// one sticky status word per device (zeroed once): a flag wait that ran into its spin bound leaves 1 + its epoch here
unsigned* status_word() {
  static std::map<int, unsigned*> words;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = words.find(dev);
  if (it != words.end()) return it->second;
  unsigned* p = nullptr;
  if (hipMalloc((void**)&p, 4096) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess) return nullptr;
  words[dev] = p;
  return p;
}
}  // namespace

// 0: no chain launch on this device ever gave up a flag wait; else 1 + the epoch the first one waited for.  Synchronises.
// Once a wait has given up, the sticky word ends every later wait of every launch at once (poll_ok), so the damaged
// launches finish quickly — on unfinished neighbour data: their results are garbage.  Nothing may consume them: the
// models read this word whenever they read their loss scalars (models/base.py: get_current_log raises), and once this
// function has seen a non-zero word the library stops using chain launches in this process (one launch per convolution
// from then on).  (Leaving the kernel early instead — an abort word every wave looks at once per layer — was measured at
// -4 % on the headline step, 645 -> 620 LR-patches/s, for a path that only runs after a failure; not kept.)
extern "C" int neosr_conv_chain_status(void) {
  unsigned* p = status_word();
  unsigned v = 0;
  if (!p || hipMemcpy(&v, p, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (v) g_chain_tripped = true;
  return (int)v;
}

namespace {
__global__ void chain_health_kernel(unsigned* status, float* dst) {
  // dst[0]: 1 when a flag wait lasted ~1 ms or more since the last neosr_conv_chain_ack; dst[1]: the sticky abort word
  const unsigned slow = __hip_atomic_load(status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  dst[0] = slow ? 1.f : 0.f;
  dst[1] = (float)__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

// The two health words of this device's chain launches as floats in dst[0..1] (device memory), enqueued on `stream` — no
// host synchronisation: the models append them to the loss scalars they reduce over the ranks.
extern "C" int neosr_conv_chain_health(float* dst, void* stream) {
  NEOSR_CHECK(dst, "conv_chain_health: null destination");
  unsigned* p = status_word();
  NEOSR_CHECK(p, "conv chain: no status word");
  hipLaunchKernelGGL(chain_health_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, p, dst);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// forget the slow-wait mark (enqueued on `stream`)
extern "C" int neosr_conv_chain_ack(void* stream) {
  unsigned* p = status_word();
  NEOSR_CHECK(p, "conv chain: no status word");
  NEOSR_HIP(hipMemsetAsync(p + 1, 0, 4, (hipStream_t)stream));
  return 0;
}

// tests: leave the mark a slow flag wait would leave
extern "C" int neosr_debug_chain_mark_slow(void* stream) {
  unsigned* p = status_word();
  NEOSR_CHECK(p, "conv chain: no status word");
  const unsigned one = 1;
  NEOSR_HIP(hipMemcpyAsync(p + 1, &one, 4, hipMemcpyHostToDevice, (hipStream_t)stream));
  NEOSR_HIP(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// Workgroups of a chain launch that are certainly co-resident: one per CU of the current device (MI355X: 256 when no
// device is visible, e.g. while sizing a workspace on a build host).
int neosr_conv::chain_max_tiles() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  return n_cu;
}

// Returns 0 on success, 1 on error (message set), -1 when the layers do not qualify (caller launches them one by one).
int neosr_conv::launch_wino4_chain(const neosr_conv_desc* d, const int* dep, int n, unsigned* flags, void* stream,
                                   bool profile) {
  if (n < 1 || n > W4_MAX_LAYERS || !chain_enabled()) return -1;
  const neosr_conv_desc& f = d[0];
  const int tiles_x = ceil_div(f.W, QT), tiles_y = ceil_div(f.H, QT);
  const int64_t tiles = (int64_t)tiles_x * tiles_y * f.B;
  if (tiles > chain_max_tiles()) return -1;   // one resident workgroup per tile, one workgroup per CU (153 KB of LDS)
  for (int i = 0; i < n; ++i)
    if (d[i].N > 32 && wino4_n64_mode() == 0) return -1;   // 64 output channels need the 64-channel wave mapping here
  auto small = [&](const void* p, int cs) { return !p || (int64_t)f.B * f.H * f.W * cs * 4 < (int64_t(1) << 31); };
  W4ChainArgs a;
  memset(&a, 0, sizeof(a));
  W4Layer* tab = a.layers;
  for (int i = 0; i < n; ++i) {
    const neosr_conv_desc& c = d[i];
    if (c.B != f.B || c.H != f.H || c.W != f.W || c.in_cs != f.in_cs) return -1;
    if (!c.w_wino4 || (uintptr_t)c.w_wino4 % 16 || c.ups || c.s2d_c || c.in_mask || c.in_prelu || c.accumulate ||
        c.out2 || c.out_mask_slopes || c.out_mask_gelu ||
        c.act == NEOSR_ACT_PRELU || c.act == NEOSR_ACT_GELU || c.K % 32 || c.K < 64 || c.N > 64 || c.N % 4 || c.in_cs % 4 || c.out_cs % 4 ||
        (uintptr_t)c.in % 16 || (uintptr_t)c.out % 16 || (uintptr_t)c.bias % 16 || (uintptr_t)c.res1 % 16 ||
        (uintptr_t)c.res2 % 16 || (uintptr_t)c.out_mask % 16 || c.res1_cs % 4 || c.res2_cs % 4 || c.out_mask_cs % 4 ||
        c.res1_nch % 4 || c.res2_nch % 4)
      return -1;
    if (!small(c.in, c.in_cs) || !small(c.out, c.out_cs) || !small(c.res1, c.res1_cs) || !small(c.res2, c.res2_cs) ||
        !small(c.out_mask, c.out_mask_cs))
      return -1;
    W4Layer& L = tab[i];
    memset(&L, 0, sizeof(L));
    L.in = c.in; L.u = c.w_wino4; L.bias = c.bias; L.res1 = c.res1; L.res2 = c.res2; L.out_mask = c.out_mask; L.out = c.out;
    L.K = c.K; L.N = c.N; L.out_cs = c.out_cs; L.res1_cs = c.res1_cs; L.res1_nch = c.res1_nch; L.res2_cs = c.res2_cs;
    L.res2_nch = c.res2_nch; L.out_mask_cs = c.out_mask_cs; L.act = c.act; L.n64 = c.N > 32 ? 1 : 0;
    L.dep = (i == 0 || !dep) ? -1 : dep[i];
    L.slope = c.slope; L.alpha = c.alpha; L.alpha2 = c.alpha2; L.out_mask_slope = c.out_mask_slope;
  }
  hipStream_t st = (hipStream_t)stream;
  a.nlayers = n;
  a.status = status_word();
  NEOSR_CHECK(a.status, "conv chain: no status word");
  a.indep = dep ? 0 : 1;   // no dependency array: independent layers (no flags are read or written)
  a.flags = flags ? flags : a.status + 64;
  a.B = f.B; a.H = f.H; a.W = f.W; a.in_cs = f.in_cs;
  a.tiles_x = tiles_x; a.tiles_y = tiles_y;
  auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  a.tx_shift = lg2(tiles_x);
  a.ty_shift = lg2(tiles_y);
  a.xcd = xcd_enabled() ? 1 : 0;
  a.sync = g_chain_sync;
  a.timeline = debug_timeline();
  if (profile && neosr_prof_on()) {
    double fl = 0, by = 0;
    const double px = (double)f.B * f.H * f.W;
    for (int i = 0; i < n; ++i) {
      fl += 2.0 * px * d[i].K * d[i].N * 9.0;
      by += 4.0 * (px * d[i].K + px * d[i].N + 9.0 * d[i].K * d[i].N);
    }
    neosr_prof_begin(f.mode == NEOSR_CONV_FWD ? NEOSR_PROF_CONV_FWD : NEOSR_PROF_CONV_DGRAD, stream, fl, by);
    neosr_prof_algo(2);
    neosr_prof_layers(n);
  }
  if (fast_matmul()) hipLaunchKernelGGL(conv3x3_wino4_chain_split_kernel, dim3((unsigned)tiles), dim3(768), 0, st, a);
  else hipLaunchKernelGGL(conv3x3_wino4_chain_kernel, dim3((unsigned)tiles), dim3(768), 0, st, a);
  if (profile && neosr_prof_on()) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

// One big convolution (more pixel tiles than CUs) as chain launches over STRIPS of samples: layer i of a launch is the same
// convolution on the next group of samples, so a persistent workgroup walks through up to 15 tiles and the request for
// the next tile's first chunk runs under the current tile's last one (a fresh workgroup per tile pays its prologue and
// its dispatch in the open: ~5k of ~40k cycles at K = 64).  Same kernel arithmetic: bit-identical to launch_wino4 with the
// same workgroup shape.  Returns 0 = done, 1 = error, -1 = not applicable.
int neosr_conv::launch_wino4_strips(const neosr_conv_desc& d, void* stream) {
  static const bool on = [] { const char* e = getenv("NEOSR_AMD_CONV_STRIPS"); return !(e && e[0] == '0'); }();
  if (!on || !chain_enabled()) return -1;
  const int tps = ceil_div(d.W, QT) * ceil_div(d.H, QT);
  const int cap = chain_max_tiles();
  if (d.ups || d.accumulate || d.N > 64 || d.K % 32 || d.K < 64 || tps > cap || (int64_t)tps * d.B < 2 * cap) return -1;
  if (d.out2 || d.out_mask_slopes || d.out_mask_gelu || d.act == NEOSR_ACT_PRELU || d.act == NEOSR_ACT_GELU) return -1;   // (epilogue features of the one-layer kernel only)
  if (d.N > 32 && wino4_n64_mode() == 0) return -1;
  const int gs = cap / tps;                 // samples per layer
  if (gs < 1 || d.B % gs) return -1;
  const int nlay = d.B / gs;
  auto off = [&](const float* p, int cs, int s) { return p ? p + (int64_t)s * gs * d.H * d.W * cs : nullptr; };
  for (int l0 = 0; l0 < nlay; l0 += W4_MAX_LAYERS) {
    const int n = nlay - l0 < W4_MAX_LAYERS ? nlay - l0 : W4_MAX_LAYERS;
    neosr_conv_desc dd[W4_MAX_LAYERS];
    for (int i = 0; i < n; ++i) {
      neosr_conv_desc& c = dd[i];
      c = d;
      c.B = gs;
      const int s = l0 + i;
      c.in = off(d.in, d.in_cs, s);
      c.out = const_cast<float*>(off(d.out, d.out_cs, s));
      c.res1 = off(d.res1, d.res1_cs, s);
      c.res2 = off(d.res2, d.res2_cs, s);
      c.out_mask = off(d.out_mask, d.out_mask_cs, s);
    }
    const int rc = launch_wino4_chain(dd, nullptr, n, nullptr, stream, false);
    if (rc != 0) {
      if (rc < 0 && l0 == 0) return -1;
      if (rc < 0) neosr_set_error("conv strips: the chain kernel refused a later launch");
      return 1;
    }
  }
  return 0;
}
