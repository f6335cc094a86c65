// attn.hip — LayerNorm and (shifted-)window multi-head self-attention of SwinIR for gfx950, plus
// the channels-last PixelShuffle of its upsampler.
//
// Window attention (neosr/archs/swinir_arch.py:150-212 WindowAttention.forward, :343-392
// SwinTransformerBlock.forward): window 8x8 -> N = 64 tokens, head_dim <= 32.  One 256-thread
// workgroup per (window, head).  Q (pre-scaled), K, V (and dO) tiles live in LDS with odd row
// strides; QK^T, PV and the four backward products run on v_mfma_f32_32x32x2_f32 with fragments read
// straight from those tiles (transposed operands are free: lanes walk the contiguous index).
// torch.roll(-shift) + window_partition + head split + window_reverse + torch.roll(+shift) are pure
// addressing: token n of window (Wy, Wx) is pixel ((Wy*8 + n/8 + shift) % H, (Wx*8 + n%8 + shift) % W).
// The relative-position index and the shifted-window mask are evaluated analytically.
// Softmax statistics (log-sum-exp per row) are the only thing kept for backward; P is recomputed.
#include "common.h"
#include "../../include/neosr_amd.h"
#include "prof.h"
#include "attn_wave.h"
#include "attn_rows.h"

namespace {

#ifdef WATTN_TL   // debug timeline of the SwinIR attention backward (tools/timeline_fattn.py swin): clocks of workgroup 0, wave W
__device__ unsigned long long g_wattn_tl[2][32];
#define WTL(i) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 128)) g_wattn_tl[threadIdx.x >> 7][i] = clock64(); } while (0)
#else
#define WTL(i) do {} while (0)
#endif

inline int grid_for(int64_t work_items, int cap = 4096) {
  int64_t g = (work_items + 255) / 256;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ------------------------------------------------------------------------------ LayerNorm
constexpr int LN_MAXI = 8;  // C <= 512 (NI = ceil(C/64) <= 8 values per lane)

// One wave per row, NI = ceil(C / 64) values per lane (compile time, so a row is exactly NI
// coalesced 256-byte loads); the next row's loads are issued before the current row is reduced.
template <int NI>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ y,
                                                            float* __restrict__ stats, int64_t rows,
                                                            int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float gm[NI], bt[NI], v[NI], nx[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    gm[i] = c < C ? gamma[c] : 0.f;
    bt[i] = c < C ? beta[c] : 0.f;
    v[i] = (row < rows && c < C) ? x[row * C + c] : 0.f;
  }
  for (; row < rows; row += stride) {
    const int64_t nrow = row + stride;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      nx[i] = (nrow < rows && c < C) ? x[nrow * C + c] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) s += v[i];
    const float mean = wave_allreduce_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float dlt = lane + 64 * i < C ? v[i] - mean : 0.f;
      q += dlt * dlt;
    }
    const float rstd = rsqrtf(wave_allreduce_sum(q) / C + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      if (c < C) y[row * C + c] = (v[i] - mean) * rstd * gm[i] + bt[i];
      v[i] = nx[i];
    }
    if (lane == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
  }
}

template <int NI>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ dx,
                                                            float* __restrict__ part, int64_t rows,
                                                            int C, const float* __restrict__ dres) {
  __shared__ float red[2][4][NI * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float gm[NI], dg[NI], db[NI], g[NI], xv[NI], ng[NI], nxv[NI], dr[NI], ndr[NI];   // dr: the shortcut gradient of the row
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = lane + 64 * i;
    const bool ok = row < rows && c < C;
    gm[i] = c < C ? gamma[c] : 0.f;
    dg[i] = db[i] = 0.f;
    g[i] = ok ? dy[row * C + c] : 0.f;
    xv[i] = ok ? x[row * C + c] : 0.f;
    dr[i] = (ok && dres) ? dres[row * C + c] : 0.f;
  }
  // (the row's (mean, rstd) pair travels with the row's prefetch: read at its use it was one exposed global round trip per row)
  float mean = row < rows ? stats[2 * row] : 0.f, rstd = row < rows ? stats[2 * row + 1] : 0.f;
  for (; row < rows; row += stride) {
    const int64_t nrow = row + stride;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      const bool ok = nrow < rows && c < C;
      ng[i] = ok ? dy[nrow * C + c] : 0.f;
      nxv[i] = ok ? x[nrow * C + c] : 0.f;
      ndr[i] = (ok && dres) ? dres[nrow * C + c] : 0.f;
    }
    const float nmean = nrow < rows ? stats[2 * nrow] : 0.f, nrstd = nrow < rows ? stats[2 * nrow + 1] : 0.f;
    float gy[NI], xh[NI];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      xh[i] = lane + 64 * i < C ? (xv[i] - mean) * rstd : 0.f;
      gy[i] = g[i] * gm[i];
      s1 += gy[i];
      s2 += gy[i] * xh[i];
      dg[i] += g[i] * xh[i];
      db[i] += g[i];
    }
    s1 = wave_allreduce_sum(s1) / C;
    s2 = wave_allreduce_sum(s2) / C;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = lane + 64 * i;
      // dres: gradient arriving over the shortcut that by-passed this norm (x -> x + f(norm(x))): summed here instead
      // of by a separate elementwise pass
      if (c < C) dx[row * C + c] = rstd * (gy[i] - s1 - xh[i] * s2) + dr[i];
      g[i] = ng[i];
      xv[i] = nxv[i];
      dr[i] = ndr[i];
    }
    mean = nmean;
    rstd = nrstd;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    red[0][wave][lane + 64 * i] = dg[i];
    red[1][wave][lane + 64 * i] = db[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part[((int64_t)blockIdx.x * 2 + 0) * C + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    part[((int64_t)blockIdx.x * 2 + 1) * C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  }
}

// ------------------------------------------------------------------------------ window attention
constexpr int WS = 8, NTOK = 64, HD_MAX = 32, QS = 33 /* q/k/v row stride */, PS = 65 /* P row stride */;
constexpr int NB = (2 * WS - 1) * (2 * WS - 1);  // 225 relative positions

// see attn_flash.hip: contiguous band of logical workgroup ids per XCD, so the heads of one window share an L2
__device__ __forceinline__ int xcd_bid() {
  const int n = gridDim.x, q = n >> 3, r = n & 7;
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct Win {
  int b, Wy, Wx, head;
};

__device__ __forceinline__ Win decode(const neosr_wattn_desc& d, int bid) {
  const int nWx = d.W / WS, nW = (d.H / WS) * nWx;
  Win w;
  w.head = bid % d.heads;
  const int t = bid / d.heads;
  const int wi = t % nW;
  w.b = t / nW;
  w.Wy = wi / nWx;
  w.Wx = wi - w.Wy * nWx;
  return w;
}

// Per-workgroup tables, built once by the first 64 / 225 threads so the hot loops do no integer
// division and no global gathers:
//   tok[n]  pixel row of window token n in the (B*H*W, .) matrices: the cyclic shift (torch.roll) and
//           window_partition / window_reverse are this one index
//   reg[n]  shifted-window mask region of token n (swinir_arch.py:313-341 calculate_mask): 3x3 regions
//           of the rolled image; pairs from different regions get -100
//   tab[k]  this head's column of relative_position_bias_table
struct Tables {
  int tok[NTOK];
  int reg[NTOK];
  float tab[NB + 31];
};

// pixel row of window token n (the cyclic shift and window_partition as one index)
__device__ __forceinline__ int window_token(const neosr_wattn_desc& d, const Win& w, int n) {
  int y = w.Wy * WS + (n >> 3) + d.shift, x = w.Wx * WS + (n & 7) + d.shift;  // shift < WS <= H, W: one conditional subtract == modulo
  if (y >= d.H) y -= d.H;
  if (x >= d.W) x -= d.W;
  return (w.b * d.H + y) * d.W + x;
}

__device__ __forceinline__ void build_tables(const neosr_wattn_desc& d, const Win& w, Tables& T) {
  const int n = threadIdx.x;
  if (n < NTOK) {
    const int ys = w.Wy * WS + (n >> 3), xs = w.Wx * WS + (n & 7);
    T.tok[n] = window_token(d, w, n);
    const int ry = ys < d.H - WS ? 0 : (ys < d.H - d.shift ? 1 : 2);
    const int rx = xs < d.W - WS ? 0 : (xs < d.W - d.shift ? 1 : 2);
    T.reg[n] = d.shift > 0 ? ry * 3 + rx : 0;
  }
  if (n < NB) T.tab[n] = d.rpb_table[n * d.heads + w.head];
}

// One [64 x hd] slice of the fused qkv matrix (or of dout / out) in two halves — request (8 registers per thread: token
// n = tid / 4, columns 8 (tid % 4)..) and store to LDS, zero padded to 32 columns — so that a kernel has the loads of ALL its
// slices in flight before it waits for the first.  The rows are raw-buffer loads (attn_rows.h): written as conditional loads
// from global pointers, hipcc put s_waitcnt vmcnt(0) between the slices (five dependent round trips in the backward's
// prologue; round 6, found in the ISA).  The columns past hd are zeroed by put_tile.
// (`tok` = window_token of the thread's own token n, worked out in registers: the requests do not wait for the tables)
__device__ __forceinline__ void fetch_tile(int tok, const Rows& R, int col0, int hd, float (&v)[8]) {
  load_row8_at(R, row_off(R, tok, col0, threadIdx.x & 3), row_al8(R, col0, hd), v);
}
__device__ __forceinline__ void put_tile(const float (&v)[8], float mul, float* dst, int hd) {
  const int n = threadIdx.x >> 2, part = threadIdx.x & 3;
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[n * QS + part * 8 + e] = part * 8 + e < hd ? v[e] * mul : 0.f;
}

// D[i][j] (one 32x32 tile: rows 32*ti.., cols 32*tj..) = sum_k A[i][k] * B[j][k], k < kdim (even)
// A, B in LDS with row strides sa, sb.  Returns the tile in MFMA layout (col j = lane&31).
__device__ __forceinline__ f32x16 tile_abt(const float* A, int sa, const float* B, int sb, int ti, int tj,
                                           int kdim, int l31, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = A + (32 * ti + l31) * sa + lh;
  const float* bp = B + (32 * tj + l31) * sb + lh;
  // head_dim 30 (every SwinIR variant: 180 / 6, 60 / 6 is 10) is 15 steps, written straight-line: a rolled loop is a chain of
  // read, wait, MFMA (`#pragma unroll 5` on the run-time trip count is refused by hipcc: "loop not unrolled") — 2 570 cycles per
  // product against 960 of MFMA (tools/timeline_wattn.py)
  if (kdim == 30) {
#pragma unroll
    for (int ks = 0; ks < 15; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
    return acc;
  }
  for (int ks = 0; ks < kdim / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
  return acc;
}

// D[i][j] = sum_k A[k][i] * B[k][j]  (A^T B), k < 64: lanes walk the contiguous index of both
__device__ __forceinline__ f32x16 tile_atb(const float* A, int sa, const float* B, int sb, int ti, int tj,
                                           int l31, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = A + lh * sa + 32 * ti + l31;
  const float* bp = B + lh * sb + 32 * tj + l31;
#pragma unroll 8
  for (int ks = 0; ks < NTOK / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks * sa], bp[2 * ks * sb], acc, 0, 0, 0);
  return acc;
}

// D[i][j] = sum_k A[i][k] * B[k][j], k < 64
__device__ __forceinline__ f32x16 tile_ab(const float* A, int sa, const float* B, int sb, int ti, int tj,
                                          int l31, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = A + (32 * ti + l31) * sa + lh;
  const float* bp = B + lh * sb + 32 * tj + l31;
#pragma unroll 8
  for (int ks = 0; ks < NTOK / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks * sb], acc, 0, 0, 0);
  return acc;
}

// scores tile -> LDS: s = q.k + rpb + mask; if lse_row != nullptr store exp(s - lse[i]) instead.
// Register r of the MFMA tile is query token i = 32 ti + (r&3) + 8 (r>>2) + 4 lh, the lane's column is
// key token j: the relative-position index (yi - yj + 7) * 15 + (xi - xj + 7) splits into a per-lane
// base and a compile-time term per register.
__device__ __forceinline__ void scores_to_lds(const Tables& T, const f32x16& acc, int ti, int tj, int l31,
                                              int lh, const float* lse_row, float* P) {
  const int j = 32 * tj + l31;
  const int rj = T.reg[j];
  const float* tb = T.tab + (WS - 1 - (j >> 3) + 4 * ti) * (2 * WS - 1) + (WS - 1 - (j & 7)) + 4 * lh;
  const int i0 = 32 * ti + 4 * lh;
  // (table values, query regions and row LSEs of the 16 scores in three batches of LDS reads, the mask as a select: the
  // per-score `if` form was a read + wait + branch chain)
  float bv[16], lv[16];
  int rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int di = (r & 3) + 8 * (r >> 2);  // i = i0 + di: yi = 4 ti + (r>>2), xi = (r&3) + 4 lh
    bv[r] = tb[(r >> 2) * (2 * WS - 1) + (r & 3)];
    rv[r] = T.reg[i0 + di];
    lv[r] = lse_row ? lse_row[i0 + di] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int di = (r & 3) + 8 * (r >> 2);
    float s = acc[r] + bv[r];
    s -= rv[r] != rj ? 100.f : 0.f;
    P[(i0 + di) * PS + j] = lse_row ? __expf(s - lv[r]) : s;
  }
}

__global__ __launch_bounds__(256) void window_attention_fwd_kernel(const neosr_wattn_desc d) {
  __shared__ float Qs[NTOK * QS], Ks[NTOK * QS], Vs[NTOK * QS], P[NTOK * PS];
  __shared__ Tables T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();
  const Win w = decode(d, bid);
  const int hd = d.C / d.heads, ld = 3 * d.C;
  {
    const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld);
    const int mytok = window_token(d, w, tid >> 2);
    float rq[8], rk[8], rv[8];
    fetch_tile(mytok, Rqkv, w.head * hd, hd, rq);
    fetch_tile(mytok, Rqkv, d.C + w.head * hd, hd, rk);
    fetch_tile(mytok, Rqkv, 2 * d.C + w.head * hd, hd, rv);
    build_tables(d, w, T);   // (its table loads queue behind the row requests: one round trip for everything)
    put_tile(rq, d.scale, Qs, hd);
    put_tile(rk, 1.f, Ks, hd);
    put_tile(rv, 1.f, Vs, hd);
  }
  __syncthreads();
  {
    const int ti = wave >> 1, tj = wave & 1;
    const f32x16 acc = tile_abt(Qs, QS, Ks, QS, ti, tj, (hd + 1) & ~1, l31, lh);
    scores_to_lds(T, acc, ti, tj, l31, lh, nullptr, P);
  }
  __syncthreads();
  {  // row softmax: 4 threads per row
    const int i = tid >> 2, q = tid & 3;
    float* row = P + i * PS + q * 16;
    float m = row[0];
#pragma unroll
    for (int c = 1; c < 16; ++c) m = fmaxf(m, row[c]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float s = 0.f, e[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      e[c] = __expf(row[c] - m);
      s += e[c];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = e[c] * inv;
    if (q == 0 && d.lse) d.lse[(int64_t)bid * NTOK + i] = m + __logf(s);
  }
  __syncthreads();
  if (wave < 2) {  // O = P V : rows 32*wave.., cols d
    const f32x16 acc = tile_ab(P, PS, Vs, QS, wave, 0, l31, lh);
    if (l31 < hd) {
      float* out = d.out + w.head * hd + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[(int64_t)T.tok[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh] * d.C] = acc[r];
    }
  }
}

// HAVE_O (the forward output is at hand, as in training): ONE 64 x 64 score buffer — P, then dS in place — instead of two:
// 52 KB of LDS, three workgroups per CU instead of two (-9 % on its own).  The products that read P (dV) run between P and
// the in-place dS; dQ and dK behind it are then one 32-MFMA product per wave pair (before: dV + dQ on two waves, dK on two).
template <bool HAVE_O>
__global__ __launch_bounds__(256) void window_attention_bwd_kernel(const neosr_wattn_desc d) {
  __shared__ float Qs[NTOK * QS], Ks[NTOK * QS], Vs[NTOK * QS], Gs[NTOK * QS];
  __shared__ float P[NTOK * PS];
  __shared__ float dS2[HAVE_O ? 1 : NTOK * PS];
  float* dS = HAVE_O ? P : dS2;
  __shared__ float lse_s[NTOK], delta_s[NTOK];
  __shared__ float zero_s[1];   // what the bias-bin walk reads for a pair outside the window
  __shared__ Tables T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();
  const Win w = decode(d, bid);
  const int hd = d.C / d.heads, ld = 3 * d.C, kq = (hd + 1) & ~1;
  if (tid == 0) zero_s[0] = 0.f;
  WTL(0);
  // all five slices (q, k, v, and below dO, O) are requested before anything is waited for — the thread's own token is worked
  // out in registers, the tables (and their loads) follow the requests
  float rq[8], rk[8], rv[8];
  const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld), Rdo = make_rows(d.dout, w.b, d.H * d.W, d.C);
  const int mytok = window_token(d, w, tid >> 2);
  fetch_tile(mytok, Rqkv, w.head * hd, hd, rq);
  fetch_tile(mytok, Rqkv, d.C + w.head * hd, hd, rk);
  fetch_tile(mytok, Rqkv, 2 * d.C + w.head * hd, hd, rv);
  constexpr bool have_o = HAVE_O;
  if (!have_o) {
    put_tile(rq, d.scale, Qs, hd);
    put_tile(rk, 1.f, Ks, hd);
    put_tile(rv, 1.f, Vs, hd);
  }
  if (have_o) {
    // delta[i] = sum_j P dP = sum_d dO[i][d] O[i][d]: with the forward output at hand the row sums come from two
    // 30-float rows instead of two 64 x 64 LDS tiles, and dS is finished in the score tile's registers.  The O row is
    // requested in the same batch as the dO row it multiplies (one round trip, no second pass over dO).
    const int n = tid >> 2, part = tid & 3;
    const Rows Rout = make_rows(d.out, w.b, d.H * d.W, d.C);
    float gv[8], ov[8];
    fetch_tile(mytok, Rdo, w.head * hd, hd, gv);
    fetch_tile(mytok, Rout, w.head * hd, hd, ov);
    build_tables(d, w, T);
    if (tid < NTOK) lse_s[tid] = d.lse[(int64_t)bid * NTOK + tid];
    WTL(1);
    put_tile(rq, d.scale, Qs, hd);
    WTL(2);
    put_tile(rk, 1.f, Ks, hd);
    put_tile(rv, 1.f, Vs, hd);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = part * 8 + e < hd;   // (columns past hd: the next head's values)
      Gs[n * QS + part * 8 + e] = in ? gv[e] : 0.f;
      s += in ? gv[e] * ov[e] : 0.f;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0) delta_s[n] = s;
  } else {
    float gv[8];
    fetch_tile(mytok, Rdo, w.head * hd, hd, gv);
    build_tables(d, w, T);
    if (tid < NTOK) lse_s[tid] = d.lse[(int64_t)bid * NTOK + tid];
    put_tile(gv, 1.f, Gs, hd);
  }
  WTL(3);
  __syncthreads();
  WTL(4);
  float* g = d.dqkv + w.head * hd + l31;
  {
    const int ti = wave >> 1, tj = wave & 1;
    const f32x16 s = tile_abt(Qs, QS, Ks, QS, ti, tj, kq, l31, lh);
    WTL(5);
    scores_to_lds(T, s, ti, tj, l31, lh, lse_s, P);                   // P = softmax (recomputed)
    WTL(6);
    const f32x16 dp = tile_abt(Gs, QS, Vs, QS, ti, tj, kq, l31, lh);  // dP = dO V^T
    WTL(7);
    const int i0 = 32 * ti + 4 * lh, j = 32 * tj + l31;
    if (HAVE_O) {
      __syncthreads();   // P complete
      WTL(8);
      if (wave >= 2) {   // dV[j][d] = sum_i P[i][j] dO[i][d]
        const f32x16 dv = tile_atb(P, PS, Gs, QS, wave - 2, 0, l31, lh);
        if (l31 < hd) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            g[(int64_t)T.tok[32 * (wave - 2) + (r & 3) + 8 * (r >> 2) + 4 * lh] * ld + 2 * d.C] = dv[r];
        }
      }
      WTL(9);
      __syncthreads();   // P has been read: dS over it, every lane on the elements it wrote
      WTL(10);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2);
        dS[i * PS + j] = P[i * PS + j] * (dp[r] - delta_s[i]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2);
        dS[i * PS + j] = dp[r];
      }
    }
  }
  WTL(11);
  __syncthreads();
  WTL(12);
  if (!have_o) {  // dS = P * (dP - sum_j P dP), 4 threads per row
    const int i = tid >> 2, q = tid & 3;
    const float* pr = P + i * PS + q * 16;
    float* gr = dS + i * PS + q * 16;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += pr[c] * gr[c];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
#pragma unroll
    for (int c = 0; c < 16; ++c) gr[c] = pr[c] * (gr[c] - s);
    __syncthreads();
  }
  // relative-position-bias gradient of this (window, head): bin sums in a fixed order
  if (tid < NB) {
    // all 64 query positions, straight-line: the pairs that leave the window read element 0 and add 0, so the LDS
    // reads are independent (a loop over the valid range is a chain of dependent read + add latencies); same
    // summation order (yi, then xi)
    const int dy = tid / (2 * WS - 1) - (WS - 1), dx = tid % (2 * WS - 1) - (WS - 1);
    // (all 64 reads in ONE batch in front of the additions: hipcc groups them by eight and waits for every group —
    // eight LDS round trips, 3 100 cycles of a 28 600-cycle workgroup; tools/timeline_wattn.py)
    float v[WS * WS];
#pragma unroll
    for (int yi = 0; yi < WS; ++yi) {
      const int yj = yi - dy;
      const bool oky = yj >= 0 && yj < WS;
#pragma unroll
      for (int xi = 0; xi < WS; ++xi) {
        const int xj = xi - dx;
        const bool ok = oky && xj >= 0 && xj < WS;
        // (a pair that leaves the window reads a word that holds 0: no predicate at the addition — 64 of them lived in
        // scalar register pairs across the batch and spilled)
        const float* src = ok ? dS + (yi * WS + xi) * PS + yj * WS + xj : zero_s;
        v[yi * WS + xi] = *src;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < WS * WS; ++i) s += v[i];
    d.workspace[((int64_t)(bid / d.heads) * NB + tid) * d.heads + w.head] = s;
  }
  WTL(13);
  if (wave < 2) {
    // dQ[i][d] = scale * sum_j dS[i][j] K[j][d]
    const f32x16 dq = tile_ab(dS, PS, Ks, QS, wave, 0, l31, lh);
    f32x16 dv;
    if (!HAVE_O) dv = tile_atb(P, PS, Gs, QS, wave, 0, l31, lh);   // dV[j][d] = sum_i P[i][j] dO[i][d]
    if (l31 < hd) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t t = (int64_t)T.tok[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh] * ld;
        if (!HAVE_O) g[t + 2 * d.C] = dv[r];
        g[t] = dq[r] * d.scale;
      }
    }
  } else {
    // dK[j][d] = sum_i dS[i][j] (scale * Q[i][d])   (Qs holds the scaled q)
    const f32x16 dk = tile_atb(dS, PS, Qs, QS, wave - 2, 0, l31, lh);
    if (l31 < hd) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        g[(int64_t)T.tok[32 * (wave - 2) + (r & 3) + 8 * (r >> 2) + 4 * lh] * ld + d.C] = dk[r];
    }
  }
  WTL(14);
}

#ifdef WATTN_TL
}  // namespace
extern "C" int neosr_debug_wattn_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wattn_tl), 64 * 8) == hipSuccess ? 0 : 1;
}
namespace {
#endif
// ------------------------------------------------------------------------------ misc
__global__ __launch_bounds__(256) void pixel_shuffle_nhwc_kernel(const float* __restrict__ in,
                                                                 float* __restrict__ out, int B, int H,
                                                                 int W, int C, int r, int inverse) {
  // lo (B,H,W,C*r*r)  <->  hi (B,H*r,W*r,C): hi[b, h*r+i, w*r+j, c] = lo[b, h, w, c*r*r + i*r + j]
  const int64_t total = (int64_t)B * H * W * C * r * r;
  const int Wh = W * r;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    int64_t t = e / C;
    const int xo = (int)(t % Wh);
    t /= Wh;
    const int yo = (int)(t % (H * r));
    const int b = (int)(t / (H * r));
    const int h = yo / r, i = yo - h * r, wq = xo / r, j = xo - wq * r;
    const int64_t lo = (((int64_t)b * H + h) * W + wq) * (C * r * r) + c * r * r + i * r + j;
    if (!inverse) out[e] = in[lo];
    else out[lo] = in[e];
  }
}

// r = 2 (every PixelShuffle of the x2 / x4 upsamplers), C * 4 low-resolution channels on 16-byte aligned storage: one thread
// per (low-resolution pixel, c) moves the quad lo[.., 4 c .. 4 c + 3] = the 2 x 2 output pixels of channel c — one 16-byte
// access on the low side, four dword accesses on the high side that are contiguous across the threads' c; 32-bit index
// arithmetic.  (The generic kernel above gathers 4 bytes at a 16-byte stride and divides five times in 64 bits per element:
// 76 us for the 33 MB map of swinir_medium's upsampler.)
__global__ __launch_bounds__(256) void pixel_shuffle2_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  int B, int H, int W, int C, int inverse) {
  const unsigned total = (unsigned)B * H * W * C;
  for (unsigned q = blockIdx.x * 256u + threadIdx.x; q < total; q += gridDim.x * 256u) {
    const unsigned c = q % C, pix = q / C;
    const unsigned w = pix % W, t = pix / W, h = t % H, b = t / H;
    const int64_t hi0 = (((int64_t)b * H * 2 + 2 * h) * (2 * W) + 2 * w) * C + c;   // (i, j) = (0, 0)
    const int64_t row = (int64_t)2 * W * C;
    if (!inverse) {
      const float4 v = reinterpret_cast<const float4*>(in)[q];
      out[hi0] = v.x;
      out[hi0 + C] = v.y;
      out[hi0 + row] = v.z;
      out[hi0 + row + C] = v.w;
    } else {
      reinterpret_cast<float4*>(out)[q] = make_float4(in[hi0], in[hi0 + C], in[hi0 + row], in[hi0 + row + C]);
    }
  }
}

__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     int64_t n, float shift, float scale) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = (in[e] + shift) * scale;
}

__global__ __launch_bounds__(256) void row_scale_kernel(const float* __restrict__ in,
                                                        const float* __restrict__ scale,
                                                        float* __restrict__ out, int64_t n, int cols,
                                                        int rows_per_scale) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
    out[e] = in[e] * scale[(e / cols) / rows_per_scale];
}

int wattn_check(const neosr_wattn_desc* d) {
  NEOSR_CHECK(d && d->qkv && d->rpb_table, "window_attention: null tensor");
  NEOSR_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->heads > 0, "window_attention: bad geometry");
  NEOSR_CHECK(d->ws == WS, "window_attention: only window_size 8 is implemented (got %d)", d->ws);
  NEOSR_CHECK(d->H % WS == 0 && d->W % WS == 0, "window_attention: H, W must be multiples of the window size");
  NEOSR_CHECK(d->C % d->heads == 0 && d->C / d->heads <= HD_MAX, "window_attention: head_dim must be <= 32");
  NEOSR_CHECK(d->shift >= 0 && d->shift < WS, "window_attention: 0 <= shift < window size");
  NEOSR_CHECK((int64_t)d->H * d->W * 3 * d->C * 4 <= (int64_t)ROW_DEAD && d->C / d->heads >= 2,
              "window_attention: one sample's qkv rows must stay under 3 GiB (32-bit row offsets)");
  return 0;
}

}  // namespace

extern "C" int neosr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                   float* stats, int64_t rows, int32_t C, float eps, void* stream) {
  NEOSR_CHECK(x && gamma && beta && y && stats && rows > 0 && C > 0 && C <= LN_MAXI * 64, "layernorm_fwd: bad args (C <= 512)");
  const dim3 grid(grid_for(rows * 16, 2048));  // 4 rows per wave
#define LN_FWD(NI)                                                                                            \
  hipLaunchKernelGGL(layernorm_fwd_kernel<NI>, grid, dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, \
                     stats, rows, C, eps)
  switch ((C + 63) / 64) {
    case 1: LN_FWD(1); break;
    case 2: LN_FWD(2); break;
    case 3: LN_FWD(3); break;
    case 4: LN_FWD(4); break;
    case 5: case 6: LN_FWD(6); break;
    default: LN_FWD(8); break;
  }
#undef LN_FWD
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma,
                                   float* dx, float* dgamma, float* dbeta, float* workspace, int64_t rows,
                                   int32_t C, int32_t accumulate, void* stream) {
  return neosr_layernorm_bwd_res(dy, x, stats, gamma, nullptr, dx, dgamma, dbeta, workspace, rows, C, accumulate, stream);
}

extern "C" int neosr_layernorm_bwd_res(const float* dy, const float* x, const float* stats, const float* gamma,
                                       const float* dres, float* dx, float* dgamma, float* dbeta, float* workspace,
                                       int64_t rows, int32_t C, int32_t accumulate, void* stream) {
  // dgamma == nullptr: leave the per-workgroup partials [return value][2 C] in `workspace` (dgamma | dbeta per row) for
  // the caller to reduce later, batched with others (neosr_colsum_many); the return value is then the row count
  const bool defer = !dgamma && !dbeta;
  NEOSR_CHECK(dy && x && stats && gamma && dx && (defer || (dgamma && dbeta)) && workspace && rows > 0 && C > 0 &&
                  C <= LN_MAXI * 64, "layernorm_bwd: bad args");
  int nblk = (int)((rows + 15) / 16);  // >= 4 rows per wave
  if (nblk > 1024) nblk = 1024;
#define LN_BWD(NI)                                                                                              \
  hipLaunchKernelGGL(layernorm_bwd_kernel<NI>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, x, stats, \
                     gamma, dx, workspace, rows, C, dres)
  switch ((C + 63) / 64) {
    case 1: LN_BWD(1); break;
    case 2: LN_BWD(2); break;
    case 3: LN_BWD(3); break;
    case 4: LN_BWD(4); break;
    case 5: case 6: LN_BWD(6); break;
    default: LN_BWD(8); break;
  }
#undef LN_BWD
  NEOSR_LAUNCH_CHECK();
  if (defer) return -nblk;  // (negative: not an error code)
  // per-workgroup partials [nblk][2][C] -> dgamma | dbeta; the column sums stage through the tail of
  // the workspace (behind the 2*1024*C partials)
  float* stage = workspace + (int64_t)2 * 1024 * C;
  if (dbeta == dgamma + C)  // adjacent outputs: one reduction over the 2C columns
    return neosr_colsum(workspace, dgamma, stage, nblk, 2 * C, 2 * C, accumulate, stream);
  if (int rc = neosr_colsum(workspace, dgamma, stage, nblk, C, 2 * C, accumulate, stream)) return rc;
  return neosr_colsum(workspace + C, dbeta, stage, nblk, C, 2 * C, accumulate, stream);
}

extern "C" int neosr_window_attention_fwd(const neosr_wattn_desc* d, void* stream) {
  if (int rc = wattn_check(d)) return rc;
  NEOSR_CHECK(d->out, "window_attention_fwd: out missing");
  const int nblk = d->B * (d->H / WS) * (d->W / WS) * d->heads;
  const bool prof = neosr_prof_on();
  const double tok = (double)d->B * d->H * d->W;  // Q.K^T + P.V: 4 * 64 * head_dim FLOP per (token, head)
  if (prof) neosr_prof_begin(NEOSR_PROF_ATTN_FWD, stream, 4.0 * WS * WS * tok * d->C, 4.0 * tok * 4 * d->C);
  static const bool staged = getenv("NEOSR_WATTN_STAGED") != nullptr;  // A/B switch: the workgroup-per-unit kernels
  if (neosr_wattn::wave_ok(*d) && !staged)
    neosr_wattn::launch_fwd(*d, stream);
  else
    hipLaunchKernelGGL(window_attention_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *d);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_window_attention_bwd(const neosr_wattn_desc* d, void* stream) {
  if (int rc = wattn_check(d)) return rc;
  NEOSR_CHECK(d->dout && d->dqkv && d->lse && d->d_rpb_table && d->workspace, "window_attention_bwd: null tensor");
  const int nbw = d->B * (d->H / WS) * (d->W / WS);
  const bool prof = neosr_prof_on();
  const double tok = (double)d->B * d->H * d->W;  // dP, dV, dQ, dK: four products (the recomputed P is not counted)
  if (prof) neosr_prof_begin(NEOSR_PROF_ATTN_BWD, stream, 8.0 * WS * WS * tok * d->C, 4.0 * tok * 7 * d->C);
  {
    static const bool rowpass = getenv("NEOSR_WATTN_ROWPASS") != nullptr;  // A/B: delta from the score tiles, not from O
    neosr_wattn_desc dd = *d;
    if (rowpass) dd.out = nullptr;
    if (dd.out) hipLaunchKernelGGL(window_attention_bwd_kernel<true>, dim3(nbw * d->heads), dim3(256), 0, (hipStream_t)stream, dd);
    else hipLaunchKernelGGL(window_attention_bwd_kernel<false>, dim3(nbw * d->heads), dim3(256), 0, (hipStream_t)stream, dd);
  }
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  // d_table[bin][head] (+)= column sums of the [nbw][bin*heads + head] per-window matrix (fixed order)
  const int cols = NB * d->heads;
  // accumulate_rpb == 2: leave the [nbw][cols] partials at the start of the workspace for a batched reduction
  // (neosr_colsum_many); returns -(rows)
  if (d->accumulate_rpb == 2) return -nbw;
  return neosr_colsum(d->workspace, d->d_rpb_table, d->workspace + (int64_t)nbw * cols, nbw, cols, cols,
                      d->accumulate_rpb, stream);
}

extern "C" int neosr_pixel_shuffle_nhwc(const float* in, float* out, int32_t B, int32_t H, int32_t W,
                                        int32_t C, int32_t r, int32_t inverse, void* stream) {
  NEOSR_CHECK(in && out && B > 0 && H > 0 && W > 0 && C > 0 && r > 0, "pixel_shuffle_nhwc: bad args");
  const float* lo = inverse ? out : in;
  if (r == 2 && ((uintptr_t)lo & 15) == 0 && (int64_t)B * H * W * C < (int64_t(1) << 31))
    hipLaunchKernelGGL(pixel_shuffle2_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                       in, out, B, H, W, C, inverse);
  else
    hipLaunchKernelGGL(pixel_shuffle_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * C * r * r)), dim3(256), 0,
                       (hipStream_t)stream, in, out, B, H, W, C, r, inverse);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_affine(const float* in, float* out, int64_t n, float shift, float scale, void* stream) {
  NEOSR_CHECK(in && out && n > 0, "affine: bad args");
  hipLaunchKernelGGL(affine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, shift, scale);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int neosr_row_scale(const float* in, const float* scale, float* out, int64_t rows, int32_t cols,
                               int32_t rows_per_scale, void* stream) {
  NEOSR_CHECK(in && scale && out && rows > 0 && cols > 0 && rows_per_scale > 0, "row_scale: bad args");
  hipLaunchKernelGGL(row_scale_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, in, scale,
                     out, rows * cols, cols, rows_per_scale);
  NEOSR_LAUNCH_CHECK();
  return 0;
}
