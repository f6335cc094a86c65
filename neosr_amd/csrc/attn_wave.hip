// attn_wave.hip — (shifted-)window attention of the SwinIR blocks, 8x8 windows, one WAVE per (window, head).
//
// A (window, head) problem is 64 tokens x head_dim <= 30: small enough that one wave holds every matrix of it in
// registers, so the kernels have no workgroup barrier and (forward) no LDS traffic besides the 225-entry
// relative-position table.  The layout trick that makes this work: the 32x32x2 fp32 MFMA returns D[i][j] with the
// column j in the lane and the rows i = 8g + 4lh + r in the registers — which is exactly what the NEXT product
// needs as its first operand when it contracts over i (lane = row of the operand, k-slot lh <-> register (g, r)),
// provided the other operand is fetched with the same k order.  So
//   forward   S^T[j][i] = K Q^T (lane = query i, registers = keys j): the softmax over j is an in-register
//             reduction + one cross-half shuffle, and O = P V contracts over the registers with V read column-wise
//             (lane = feature d) straight from global memory.
// (A backward in the same style — 4 independent units per (window, head), S / dP recomputed on the query and on the key
// side — was built and measured: 184 us against 137 us for the workgroup-per-unit kernel of attn.hip, which stays.)
// Row operands (lane = token, 15 consecutive features per half-wave) are loaded directly from the fused qkv matrix
// in image order: torch.roll, window_partition / window_reverse and the head split are address arithmetic.
// A wave loops over its (window, head) units and requests the next unit's rows as soon as the current scores
// are formed.  Reference: neosr/archs/swinir_arch.py:150-212 (WindowAttention), :313-341 (mask), :343-392.
#include <cstdlib>
#include "common.h"
#include "attn_wave.h"
#include "prof.h"

namespace {

constexpr int WS = 8, NTOK = 64, NBIN = (2 * WS - 1) * (2 * WS - 1), NS = 16;

struct __attribute__((packed, aligned(4))) F3 { float v[3]; };

// see attn_flash.hip: contiguous band of logical workgroup ids per XCD, so the heads of one window share an L2
__device__ __forceinline__ int xcd_bid() {
  const int n = gridDim.x, q = n >> 3, r = n & 7;
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct Unit {
  int b, Wy, Wx, head;
  int nWy, nWx;
};

__device__ __forceinline__ Unit decode(const neosr_wattn_desc& d, int u) {
  Unit w;
  w.nWx = d.W / WS;
  w.nWy = d.H / WS;
  const int nW = w.nWy * w.nWx;
  w.head = u % d.heads;
  const int t = u / d.heads;
  const int wi = t % nW;
  w.b = t / nW;
  w.Wy = wi / w.nWx;
  w.Wx = wi - w.Wy * w.nWx;
  return w;
}

// image row / column of window coordinate (y, x) after the cyclic shift (shift < WS <= H, W)
__device__ __forceinline__ int wrap(int v, int n) { return v >= n ? v - n : v; }

// pixel index of window token n
__device__ __forceinline__ int pixel(const neosr_wattn_desc& d, const Unit& w, int n) {
  const int Y = wrap(w.Wy * WS + (n >> 3) + d.shift, d.H), X = wrap(w.Wx * WS + (n & 7) + d.shift, d.W);
  return (w.b * d.H + Y) * d.W + X;
}

// shifted-window mask region (swinir_arch.py:313-341) of window coordinate c along one axis
__device__ __forceinline__ int region1(int c, int Wc, int nWc, int shift) {
  return Wc == nWc - 1 ? (c < WS - shift ? 1 : 2) : 0;
}

// Row fragment of one 32-token tile: lane (l31, lh) <- X[pixel(32 t + l31)][col0 + half * lh + s], s < NS, zero past
// the lane's half of the head row.  HALF = 15 (head_dim 30): five 12-byte loads; HALF = 0: any head_dim <= 30.
template <int HALF>
__device__ __forceinline__ void load_rows(const float* X, int ld, int pix, int col0, int hd, int lh, float mul,
                                          float (&f)[NS]) {
  if (HALF == 15) {
    const float* p = X + (int64_t)pix * ld + col0 + 15 * lh;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const F3 v = *reinterpret_cast<const F3*>(p + 3 * q);
      f[3 * q] = v.v[0] * mul;
      f[3 * q + 1] = v.v[1] * mul;
      f[3 * q + 2] = v.v[2] * mul;
    }
    f[15] = 0.f;
  } else {
    const int half = (hd + 1) >> 1;
    const int cnt = lh ? hd - half : half;
    const float* p = X + (int64_t)pix * ld + col0 + half * lh;
#pragma unroll
    for (int s = 0; s < NS; ++s) f[s] = s < cnt ? p[s] * mul : 0.f;
  }
}

// per-lane column offsets (pixels) of the window columns 4 lh + r, r = 0..3 — the x part of the register order
__device__ __forceinline__ void col_offsets(const neosr_wattn_desc& d, const Unit& w, int lh, int64_t (&xoff)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) xoff[r] = (int64_t)wrap(w.Wx * WS + 4 * lh + r + d.shift, d.W);
}

// first pixel of the image row of window row y (wave-uniform)
__device__ __forceinline__ int64_t row_base(const neosr_wattn_desc& d, const Unit& w, int y) {
  return ((int64_t)w.b * d.H + wrap(w.Wy * WS + y + d.shift, d.H)) * d.W;
}

// Column operand of one 32-token tile t: lane (d, lh), k-slot (g, r) <-> token 32 t + 8 g + 4 lh + r (the register
// order of an MFMA result): 16 dword loads, each wave-load two 4*head_dim-byte row pieces
__device__ __forceinline__ void load_cols(const neosr_wattn_desc& d, const Unit& w, const float* base, int ld, int t,
                                          const int64_t (&xoff)[4], float (&c)[4][4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int64_t row = row_base(d, w, 4 * t + g);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[g][r] = base[(row + xoff[r]) * ld];
  }
}

__device__ __forceinline__ void zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// Row stride of the bias table in LDS.  A 32-lane group of the score epilogue reads (row yi + a, column xi), a = 0..3, xi =
// 0..7: with the table's own stride of 15 the four 8-float runs start 15 apart and rows a, a + 2 share banks (SQ counters,
// profiles/r05_bench_swinir_medium_sq_summary.json: bank conflicts 44 % of the kernel's LDS cycles); 40 apart they are the
// four disjoint quarters of the 32 banks (the HAT kernel below got the same re-stride in round 5).
constexpr int TS8 = 40;
__device__ __forceinline__ void load_table(const neosr_wattn_desc& d, int head, int lane, float* tab) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int n = lane + 64 * k;
    if (n < NBIN) tab[(n / (2 * WS - 1)) * TS8 + n % (2 * WS - 1)] = d.rpb_table[n * d.heads + head];
  }
}

// ---------------------------------------------------------------------------------------------------- forward
// unit = (window, head, query tile ti): S^T tiles [tj] with rows (registers) = keys j, column (lane) = query i
template <int HALF>
__global__ __launch_bounds__(256, 2) void wattn_wave_fwd_kernel(const neosr_wattn_desc d, int units) {
  __shared__ float tabs[4][(2 * WS - 1) * TS8];
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tab = tabs[wave];
  const int hd = d.C / d.heads, ld = 3 * d.C;
  constexpr int KS = HALF ? HALF : NS;
  const int stride = gridDim.x * 4;
  const int dl = l31 < hd ? l31 : hd - 1;  // feature of this lane in the column operands / outputs
  for (int u2 = xcd_bid() * 4 + wave; u2 < 2 * units; u2 += stride) {
    const int u = u2 >> 1, ti = u2 & 1;
    const Unit w = decode(d, u);
    float qf[NS], kf[2][NS];
    load_rows<HALF>(d.qkv, ld, pixel(d, w, 32 * ti + l31), w.head * hd, hd, lh, d.scale, qf);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
      load_rows<HALF>(d.qkv, ld, pixel(d, w, 32 * tj + l31), d.C + w.head * hd, hd, lh, 1.f, kf[tj]);
    load_table(d, w.head, lane, tab);
    int64_t xoff[4];
    col_offsets(d, w, lh, xoff);
    float vc[2][4][4];  // V as a column operand, keys in the register order of P
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) load_cols(d, w, d.qkv + 2 * d.C + w.head * hd + dl, ld, tj, xoff, vc[tj]);

    f32x16 st[2];
    zero(st[0]);
    zero(st[1]);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
        st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj][s], qf[s], st[tj], 0, 0, 0);

    // bias + mask, softmax over the keys (registers + the other half-wave)
    const bool masked = d.shift > 0 && (w.Wy == w.nWy - 1 || w.Wx == w.nWx - 1);
    const int yi = 4 * ti + (l31 >> 3), xi = l31 & 7;
    const float* tb = tab + (yi + WS - 1) * TS8 + xi + WS - 1 - 4 * lh;
    const int ri = region1(yi, w.Wy, w.nWy, d.shift) * 3 + region1(xi, w.Wx, w.nWx, d.shift);
    float m = -3.0e38f;
    // (the table values of a tile are read in one batch, and the mask is a select on a penalty that is 0 in unmasked windows:
    // `if (masked && ..) sc -= 100` compiled to one LDS read + wait + branch per score — 32 exposed LDS round trips per tile)
    const float mpen = masked ? 100.f : 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      float bv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[4 * g + r] = tb[-((4 * tj + g) * TS8 + r)];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ryj = region1(4 * tj + g, w.Wy, w.nWy, d.shift) * 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc = st[tj][4 * g + r] + bv[4 * g + r];
          sc -= (ryj + region1(4 * lh + r, w.Wx, w.nWx, d.shift) != ri) ? mpen : 0.f;
          st[tj][4 * g + r] = sc;
          m = fmaxf(m, sc);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(st[tj][r] - m);
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    if (d.lse && lh == 0) d.lse[(int64_t)u * NTOK + 32 * ti + l31] = m + __logf(sum);

    // O[i][d] = sum_j P[i][j] V[j][d]: P is its own first operand (lane = row i, k-slot = register)
    f32x16 o;
    zero(o);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[tj][4 * g + r] * inv, vc[tj][g][r], o, 0, 0, 0);
    if (l31 < hd) {
      float* ob = d.out + w.head * hd + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t row = row_base(d, w, 4 * ti + g);
#pragma unroll
        for (int r = 0; r < 4; ++r) ob[(row + xoff[r]) * d.C] = o[4 * g + r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ HAT: 16 x 16 windows (self)
// Same construction for the (shifted-)window self-attention of HAT's HAB (hat_arch.py:168-216): unit = (window, head,
// 32-query tile), 8 key tiles -> the whole 32 x 256 score block S^T lives in 128 accumulator registers, so there is no
// online softmax and no LDS tile; K tiles are streamed as row operands, V as column operands in the register order
// of P.  lse is stored in the layout of the streaming kernels ([(window, head)][256]) whose backward passes consume it.
constexpr int W16 = 16, NB16 = 2 * W16 - 1, NBIN16 = NB16 * NB16, NKT = 8;

struct Unit16 {
  int b, Wy, Wx, head, nWy, nWx, wh;
};

__device__ __forceinline__ Unit16 decode16(const neosr_fattn_desc& d, int u) {
  Unit16 w;
  w.nWx = d.W / W16;
  w.nWy = d.H / W16;
  const int nW = w.nWy * w.nWx;
  w.wh = u;
  w.head = u % d.heads;
  const int t = u / d.heads;
  const int wi = t % nW;
  w.b = t / nW;
  w.Wy = wi / w.nWx;
  w.Wx = wi - w.Wy * w.nWx;
  return w;
}

__device__ __forceinline__ int region16(int c, int Wc, int nWc, int shift) {
  return Wc == nWc - 1 ? (c < W16 - shift ? 1 : 2) : 0;
}

template <int HALF>
__global__ __launch_bounds__(256, 2) void wattn16_wave_fwd_kernel(const neosr_fattn_desc d, int units) {
  // the 31 x 31 relative-position table with a row stride of 48 floats: a ds_read_b32 lane group is two query rows (lanes
  // 0-15 / 16-31 = rows yi, yi + 1) of 16 consecutive columns — 31 apart they shared 15 banks (SQ: bank conflicts 46 % of the
  // LDS cycles, profiles/r04_bench_hat_l_otf_gan_sq_summary.json), 48 apart they take the two halves of the 32 banks
  constexpr int TS16 = 48;
  __shared__ float tabs[4][NB16 * TS16];
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tab = tabs[wave];
  const int hd = d.C / d.heads, ld = 3 * d.C;
  constexpr int KS = HALF ? HALF : NS - 1;
  const int stride = gridDim.x * 4;
  const int dl = l31 < hd ? l31 : hd - 1;
  for (int u8 = xcd_bid() * 4 + wave; u8 < 8 * units; u8 += stride) {
    const int ti = u8 & 7;
    const Unit16 w = decode16(d, u8 >> 3);
    // token n of the window = (y, x) = (n >> 4, n & 15); a 32-token tile t is window rows 2 t, 2 t + 1
    auto pix = [&](int n) {
      const int Y = wrap(w.Wy * W16 + (n >> 4) + d.shift, d.H), X = wrap(w.Wx * W16 + (n & 15) + d.shift, d.W);
      return (w.b * d.H + Y) * d.W + X;
    };
    // Every row / column fragment is requested TWO tiles ahead of the products that consume it (rings of three register
    // buffers, the order pinned by sched_barrier).  Left to itself hipcc — with 128 accumulator registers live — places
    // each 16-byte load right in front of the four MFMAs that read it and waits vmcnt(0) for it: ~32 exposed memory round
    // trips per unit on the key side alone (round 6, found in the ISA; the products were 17 % of the unit's time).
    // (the any-head-size variant — 16 conditional loads per fragment — has no registers for a ring: it loads in place, as before)
    constexpr int RING = HALF ? 3 : 1, PF = RING - 1;
    float qf[NS], kf[RING][NS];
    load_rows<HALF>(d.qkv, ld, pix(32 * ti + l31), w.head * hd, hd, lh, d.scale, qf);
    load_rows<HALF>(d.qkv, ld, pix(l31), d.C + w.head * hd, hd, lh, 1.f, kf[0]);
    if (RING > 2) load_rows<HALF>(d.qkv, ld, pix(32 + l31), d.C + w.head * hd, hd, lh, 1.f, kf[1]);
    if (!PF) {
      for (int n = lane; n < NBIN16; n += 64) tab[(n / NB16) * TS16 + n % NB16] = d.rpb_table[n * d.heads + w.head];
    } else {  // the head's bias table: all 16 loads of a lane in one batch beside the row loads above, then the LDS writes
      float tv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = lane + 64 * i;
        tv[i] = d.rpb_table[(n < NBIN16 ? n : NBIN16 - 1) * d.heads + w.head];
      }
      if (PF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = lane + 64 * i;
        if (n < NBIN16) tab[(n / NB16) * TS16 + n % NB16] = tv[i];
      }
    }
    if (PF) __builtin_amdgcn_sched_barrier(0);
    // S^T tiles: rows (registers) = keys 32 tj + 8 g + 4 lh + r, column (lane) = query 32 ti + l31
    f32x16 st[NKT];
#pragma unroll
    for (int tj = 0; tj < NKT; ++tj) {
      if (tj + PF < NKT)
        load_rows<HALF>(d.qkv, ld, pix(32 * (tj + PF) + l31), d.C + w.head * hd, hd, lh, 1.f, kf[(tj + PF) % RING]);
      if (PF) __builtin_amdgcn_sched_barrier(0);
      zero(st[tj]);
#pragma unroll
      for (int s = 0; s < KS; ++s) st[tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tj % RING][s], qf[s], st[tj], 0, 0, 0);
      if (PF) __builtin_amdgcn_sched_barrier(0);
    }
    // the first two V column tiles travel under the softmax.  (Raw-buffer loads: the image row is wave-uniform and rides in
    // the scalar offset, the eight column offsets of a lane are loop-invariant — no 64-bit address pair per load, which
    // is what the ring of three tiles has registers for.)
    const auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.qkv) + (int64_t)w.b * d.H * d.W * ld, (short)0,
                                                      (int)((unsigned)(d.H * d.W) * (unsigned)ld * 4u), 0x00020000);
    int xoff[2][4];   // byte offsets inside an image row
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xoff[h][r] = (wrap(w.Wx * W16 + 8 * h + 4 * lh + r + d.shift, d.W) * ld + 2 * d.C + w.head * hd + dl) * 4;
    auto rowb = [&](int y) { return wrap(w.Wy * W16 + y + d.shift, d.H) * d.W; };   // first pixel of window row y in the sample
    float vc[RING][4][4];
    auto load_v = [&](int tj, float (&c)[4][4]) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rb = __builtin_amdgcn_readfirstlane(rowb(2 * tj + (g >> 1)) * ld * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          c[g][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, xoff[g & 1][r], rb, 0));
      }
    };
    if (PF) {
      load_v(0, vc[0]);   // (one tile only: the softmax holds all 128 score registers + 16 bias values)
      __builtin_amdgcn_sched_barrier(0);
    }
    // bias + mask, softmax over the 256 keys (registers + the other half-wave)
    const bool masked = d.shift > 0 && (w.Wy == w.nWy - 1 || w.Wx == w.nWx - 1);
    const int yi = 2 * ti + (l31 >> 4), xi = l31 & 15;
    const float* tb = tab + (yi + W16 - 1) * TS16 + xi + W16 - 1 - 4 * lh;
    const int ri = region16(yi, w.Wy, w.nWy, d.shift) * 3 + region16(xi, w.Wx, w.nWx, d.shift);
    float m = -3.0e38f;
    // (as in the 8 x 8 kernel: the 16 table values of a key tile in one batch of LDS reads, the mask as a select on a
    // penalty that is 0 in unmasked windows — the `if (masked && ..)` form was one LDS read + wait + branch per score)
    const float mpen = masked ? 100.f : 0.f;
#pragma unroll
    for (int tj = 0; tj < NKT; ++tj) {
      float bv[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[4 * g + r] = tb[-((2 * tj + (g >> 1)) * TS16 + 8 * (g & 1) + r)];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int yj = 2 * tj + (g >> 1);
        const int ryj = region16(yj, w.Wy, w.nWy, d.shift) * 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int xj0 = 8 * (g & 1) + r;  // + 4 lh
          float sc = st[tj][4 * g + r] + bv[4 * g + r];
          sc -= (ryj + region16(xj0 + 4 * lh, w.Wx, w.nWx, d.shift) != ri) ? mpen : 0.f;
          st[tj][4 * g + r] = sc;
          m = fmaxf(m, sc);
        }
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int tj = 0; tj < NKT; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(st[tj][r] - m);
        st[tj][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.f / sum;
    if (d.lse && lh == 0) d.lse[(int64_t)w.wh * (W16 * W16) + 32 * ti + l31] = m + __logf(sum);

    // O[i][d] = sum_j P[i][j] V[j][d]; V column operand: key (yj, xj) = (2 tj + (g >> 1), 8 (g & 1) + 4 lh + r)
    f32x16 o;
    zero(o);
    if (PF > 1) load_v(1, vc[1]);
#pragma unroll
    for (int tj = 0; tj < NKT; ++tj) {
      if (tj + PF < NKT) load_v(tj + PF, vc[(tj + PF) % RING]);
      if (PF) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[tj][4 * g + r] * inv, vc[tj % RING][g][r], o, 0, 0, 0);
      if (PF) __builtin_amdgcn_sched_barrier(0);
    }
    if (l31 < hd) {
      float* ob = d.out + (int64_t)w.b * d.H * d.W * d.C + w.head * hd + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = rowb(2 * ti + (g >> 1));
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ob[(int64_t)(row + wrap(w.Wx * W16 + 8 * (g & 1) + 4 * lh + r + d.shift, d.W)) * d.C] = o[4 * g + r];
      }
    }
  }
}

}  // namespace

namespace neosr_wattn {

bool wave_ok(const neosr_wattn_desc& d) { return d.ws == WS && d.C % d.heads == 0 && d.C / d.heads <= 30; }

void launch_fwd(const neosr_wattn_desc& d, void* stream) {
  const int units = d.B * (d.H / WS) * (d.W / WS) * d.heads;
  int nwg = (2 * units + 3) / 4;
  if (nwg > 768) nwg = 768;  // three workgroups per CU (150 VGPRs); the waves walk the (window, head, query tile) list
  if (d.C / d.heads == 30)
    hipLaunchKernelGGL(wattn_wave_fwd_kernel<15>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d, units);
  else
    hipLaunchKernelGGL(wattn_wave_fwd_kernel<0>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d, units);
}

bool wave16_ok(const neosr_fattn_desc& d) {
  return d.ws == W16 && d.ks == W16 && d.C % d.heads == 0 && d.C / d.heads <= 30;
}

void launch16_fwd(const neosr_fattn_desc& d, void* stream) {
  const int units = d.B * (d.H / W16) * (d.W / W16) * d.heads;  // (window, head); 8 query tiles each
  int nwg = (8 * units + 3) / 4;
  // Two workgroups fit a CU (255 VGPRs): 512 resident.  Up to two rounds' worth of workgroups are launched one unit per
  // wave — the dispatcher hands a CU its next workgroup when one finishes, 3 units per SIMD at B = 4 instead of the 2 - 4 a
  // fixed 512-workgroup walk gives (57 -> 49 us, same box) — beyond that the 512 walk the unit list (B = 8: 87 vs 89 us).
  static const int cap = [] { const char* e = getenv("NEOSR_AMD_W16_NWG"); return e ? atoi(e) : 0; }();
  if (cap > 0) { if (nwg > cap) nwg = cap; }
  else if (nwg > 1024) nwg = 512;
  if (d.C / d.heads == 30)
    hipLaunchKernelGGL(wattn16_wave_fwd_kernel<15>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d, units);
  else
    hipLaunchKernelGGL(wattn16_wave_fwd_kernel<0>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, d, units);
}

}  // namespace neosr_wattn
