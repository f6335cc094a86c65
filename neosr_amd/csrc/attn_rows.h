// attn_rows.h — token rows of the window-attention kernels (attn.hip, attn_flash.hip) as raw-buffer loads.  Internal to
// libneosr_amd.
#pragma once
#include <cstdint>
#include "common.h"

namespace {

// Token rows of one (B*H*W, ld) matrix as a raw buffer over the workgroup's SAMPLE: a row is a 32-bit byte offset, a
// padding token (tok < 0) an offset past the end — the hardware range check returns 0 for it, and for the two floats the
// last head's 8-column group reads past the last row (hd = 30).  No branch, no 64-bit address arithmetic, no select for the
// row; only the columns past hd of an in-range group (the next head's first values) are zeroed by a select.
// (Round 6, found in the ISA: written as `cond ? *p : 0` on global pointers hipcc emits, per element, "v = 0; if (exec)
// v = load" behind a branch, and its wait-count pass — which cannot tell at the join whether the 4-byte or the 8-byte
// variant of the PREVIOUS row left loads in flight — puts s_waitcnt vmcnt(0) in front of the next row's zero moves: the
// V row of a key block waited for its K row to come back, and q / dO / O of a query block for one another — one exposed
// memory round trip per 64 x 64 tile, two more per query block.)
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((float*)nullptr, (short)0, 0, 0)) rows_rsrc_t;
struct Rows {
  rows_rsrc_t r;
  int tok0, ld;   // first token of the sample; row stride (floats)
  bool al8;       // 8-byte loads allowed (even stride, 8-byte aligned base)
};
constexpr unsigned ROW_DEAD = 0xC0000000u;   // > any sample's bytes (host check: H * W * ld * 4 <= ROW_DEAD)
__device__ __forceinline__ Rows make_rows(const float* src, int b, int hw, int ld) {
  Rows R;
  R.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + (int64_t)b * hw * ld, (short)0,
                                            (int)((unsigned)hw * (unsigned)ld * 4u), 0x00020000);
  R.tok0 = b * hw;
  R.ld = ld;
  R.al8 = (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0;
  return R;
}
// 8 consecutive floats (cols col0 + part*8..) of one token row, zero when tok < 0.  Pure loads: the columns past hd (the
// next head's first values) are zeroed where the row is CONSUMED (zero_tail8, store_row8) — a select next to the load makes
// the wave wait for the load right there, and the rows are prefetched a whole tile ahead.
__device__ __forceinline__ unsigned row_off(const Rows& R, int tok, int col0, int part) {
  return tok >= 0 ? (unsigned)((tok - R.tok0) * R.ld + col0 + part * 8) * 4u : ROW_DEAD;
}
__device__ __forceinline__ bool row_al8(const Rows& R, int col0, int hd) { return R.al8 && ((col0 | hd) & 1) == 0; }
__device__ __forceinline__ void load_row8_at(const Rows& R, unsigned off, bool al8, float (&v)[8]) {
  if (al8) {
    // even row stride, head offset and head size (30-float head slices of a 180-channel row): 8-byte aligned -> four
    // 8-byte loads instead of eight 4-byte ones (the 8x8-window kernel gained 10 % from the same change)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(R.r, off + 8 * e, 0, 0));
      v[2 * e] = t.x;
      v[2 * e + 1] = t.y;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(R.r, off + 4 * e, 0, 0));
}
__device__ __forceinline__ void load_row8(const Rows& R, int tok, int col0, int hd, int part, float (&v)[8]) {
  load_row8_at(R, row_off(R, tok, col0, part), row_al8(R, col0, hd), v);
}

}  // namespace
