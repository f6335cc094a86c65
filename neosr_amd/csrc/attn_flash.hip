// attn_flash.hip — window attention of HAT for gfx950 (template <WS, KS>: 16/16, 16/24 = hat_s / m / l; 8/8, 8/12 =
// window_size 8): WS*WS query tokens against the WS*WS keys of the same (shifted) window (HAB,
// neosr/archs/hat_arch.py:168-216 inside :299-351) or against the KS*KS keys of the overlapping window (OCAB,
// hat_arch.py:445-516).  The comments below speak of the 16 / 24 case.
//
// A 256 x 576 score matrix does not fit in LDS, so the kernels stream 64-key blocks past a 64-query
// block with an online softmax (running max / sum per row, accumulator rescaled per block); one
// 256-thread workgroup per (window, head, 64-query block).  Q.K^T, P.V and the backward products run
// on v_mfma_f32_32x32x2_f32 from LDS tiles; the next key block's K / V rows are prefetched into
// registers while the current one is consumed.  torch.roll / window_partition / nn.Unfold (zero padded)
// / einops.rearrange / window_reverse are pure addressing: a key of the overlapping window is pixel
// (Wy*16 + yk - 4, Wx*16 + xk - 4) or, outside the image, a zero row that still takes part in the
// softmax with score = bias (that is what Unfold's zero padding does in the reference).
// Relative-position indices are evaluated analytically, including the reference's negative-index
// wrap-around of calculate_rpi_oca (hat_arch.py:1035-1068: indices in [-880, 640] into a 1521-row table).
// Backward: (A) per query block: dQ, and dS tiles dumped for the bias gradient; (B) per key block:
// dK, dV (written in place for HAB; into an unfolded buffer + a fixed-order fold for OCAB, where up to
// four windows overlap on a pixel); (C) bias gradient = column sums over windows, then a fixed-order
// gather per table row.  No float atomics anywhere.
#include <cstdlib>

#include "common.h"
#include "../../include/neosr_amd.h"
#include "prof.h"
#include "attn_wave.h"
#include "attn_rows.h"

namespace {

#ifdef FATTN_TL   // debug timeline of the fused backward (tools/timeline_fattn.py): clocks of workgroup 0, thread 0
__device__ unsigned long long g_fattn_tl[64];
#define FTL(i, cond) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (cond)) g_fattn_tl[i] = clock64(); } while (0)
#else
#define FTL(i, cond) do {} while (0)
#endif

constexpr int QB = 64;        // queries / keys per block
constexpr int QS = 33;        // q/k/v LDS row stride
constexpr int PS = 65;        // score tile row stride
constexpr int TAB_MAX = 1536; // >= (16+24-1)^2

template <int WS, int KS>
struct Geo {
  static constexpr int NQ = WS * WS, NK = KS * KS, NKB = (NK + QB - 1) / QB, NQB = NQ / QB;
  static constexpr int PAD = (KS - WS) / 2, L = WS + KS - 1, NBINS = L * L;
  static constexpr bool SELF = KS == WS;
};

// Workgroup ids go round-robin over the 8 XCDs (private L2 each): hand every XCD a contiguous band of the logical ids,
// so the (head, block) workgroups of one window — which read the same token rows — share one L2 instead of each XCD
// fetching the rows again (rocprofv3 FETCH_SIZE showed 6x the algorithmic bytes with the dispatch order).
__device__ __forceinline__ int xcd_bid() {
  const int n = gridDim.x, q = n >> 3, r = n & 7;
  const int x = blockIdx.x & 7, k = blockIdx.x >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct Win {
  int b, Wy, Wx, head, qb;
};

// blockIdx -> (batch, window, head, block): block fastest so the 4 query blocks of a (window, head) —
// which read the same K / V rows — sit next to each other
__device__ __forceinline__ Win decode(const neosr_fattn_desc& d, int bid, int nblk) {
  const int nWx = d.W / d.ws, nW = (d.H / d.ws) * nWx;
  Win w;
  w.qb = bid % nblk;
  int t = bid / nblk;
  w.head = t % d.heads;
  t /= d.heads;
  const int wi = t % nW;
  w.b = t / nW;
  w.Wy = wi / nWx;
  w.Wx = wi - w.Wy * nWx;
  return w;
}

// query token n (0..NQ) of the window -> pixel row of the (B*H*W, .) matrices, and mask region
template <int WS>
__device__ __forceinline__ void query_geom(const neosr_fattn_desc& d, const Win& w, int n, int& tok, int& reg) {
  const int ys = w.Wy * WS + n / WS, xs = w.Wx * WS + n % WS;
  int y = ys + d.shift, x = xs + d.shift;
  if (y >= d.H) y -= d.H;
  if (x >= d.W) x -= d.W;
  tok = (w.b * d.H + y) * d.W + x;
  const int ry = ys < d.H - WS ? 0 : (ys < d.H - d.shift ? 1 : 2);
  const int rx = xs < d.W - WS ? 0 : (xs < d.W - d.shift ? 1 : 2);
  reg = d.shift > 0 ? ry * 3 + rx : 0;
}

// key j (0..NK) -> pixel row (or -1: zero padding / past the end), region, bias key-term
template <int WS, int KS>
__device__ __forceinline__ void key_geom(const neosr_fattn_desc& d, const Win& w, int j, int& tok, int& reg,
                                         int& kterm, bool& exists) {
  using G = Geo<WS, KS>;
  exists = j < G::NK;
  const int yj = j / KS, xj = j % KS;
  if (G::SELF) {
    query_geom<WS>(d, w, exists ? j : 0, tok, reg);
    kterm = -(yj * G::L + xj);
  } else {
    const int y = w.Wy * WS + yj - G::PAD, x = w.Wx * WS + xj - G::PAD;
    const bool in = exists && y >= 0 && y < d.H && x >= 0 && x < d.W;
    tok = in ? (w.b * d.H + y) * d.W + x : -1;
    reg = 0;
    kterm = yj * G::L + xj;
  }
  if (!exists) tok = -1;
}

// bias query-term of query n: idx = kterm + qterm (wrapped into [0, NBINS) for the overlapping form)
template <int WS, int KS>
__device__ __forceinline__ int query_term(int n) {
  using G = Geo<WS, KS>;
  const int yi = n / WS, xi = n % WS;
  if (G::SELF) return yi * G::L + xi + (WS - 1) * G::L + (WS - 1);
  return -(yi * G::L + xi) + (WS - KS + 1) * (G::L + 1);
}

struct Shared {
  float Qs[QB * QS], Ks[QB * QS], Vs[QB * QS], P[QB * PS];
  float tab[TAB_MAX];
  int qpk[QB];    // qterm * 16 + region
  int kpk[QB];    // kterm * 16 + region, or INT_MIN for a key that does not exist (masked out)
  int qtok[QB];
  float alpha[QB];
};

constexpr int KEY_NONE = -2147483647 - 1;

__device__ __forceinline__ void zero_tail8(float (&v)[8], int hd, int part) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = part * 8 + e < hd ? v[e] : 0.f;
}
__device__ __forceinline__ void store_row8(float* dst, int n, int part, const float (&v)[8], float mul, int hd) {
#pragma unroll
  for (int e = 0; e < 8; ++e) dst[n * QS + part * 8 + e] = part * 8 + e < hd ? v[e] * mul : 0.f;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  return a;
}
// D[i][j] += sum_k A[i][k] B[j][k]   (rows 32 ti.., cols 32 tj..), k < kdim
__device__ __forceinline__ f32x16 mm_abt(f32x16 acc, const float* A, int sa, const float* B, int sb, int ti,
                                         int tj, int kdim, int l31, int lh) {
  const float* ap = A + (32 * ti + l31) * sa + lh;
  const float* bp = B + (32 * tj + l31) * sb + lh;
  if (kdim == 30) {   // hat_m / hat_l heads: 15 steps straight-line (the rolled loop is read, wait, MFMA per step)
#pragma unroll
    for (int ks = 0; ks < 15; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
    return acc;
  }
  for (int ks = 0; ks < kdim / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc, 0, 0, 0);
  return acc;
}
// two products over the same k range at once (S = Q K^T and dP = dO V^T of one tile): two independent accumulator
// chains, the LDS reads of a step batched ahead of its MFMAs
__device__ __forceinline__ void mm_abt_pair(f32x16& acc1, const float* A1, const float* B1, f32x16& acc2, const float* A2,
                                            const float* B2, int sa, int sb, int ti, int tj, int kdim, int l31, int lh) {
  const int ao = (32 * ti + l31) * sa + lh, bo = (32 * tj + l31) * sb + lh;
  if (kdim == 30) {   // (see mm_abt: S, dP of a tile 2 650 -> 2 200 cycles, tools/timeline_fattn.py)
#pragma unroll
    for (int ks = 0; ks < 15; ++ks) {
      const float a1 = A1[ao + 2 * ks], b1 = B1[bo + 2 * ks], a2 = A2[ao + 2 * ks], b2 = B2[bo + 2 * ks];
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc2, 0, 0, 0);
    }
    return;
  }
  for (int ks = 0; ks < kdim / 2; ++ks) {
    const float a1 = A1[ao + 2 * ks], b1 = B1[bo + 2 * ks], a2 = A2[ao + 2 * ks], b2 = B2[bo + 2 * ks];
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc2, 0, 0, 0);
  }
}
// D[i][j] += sum_k A[i][k] B[k][j], k < 64
__device__ __forceinline__ f32x16 mm_ab(f32x16 acc, const float* A, int sa, const float* B, int sb, int ti, int tj,
                                        int l31, int lh) {
  const float* ap = A + (32 * ti + l31) * sa + lh;
  const float* bp = B + lh * sb + 32 * tj + l31;
#pragma unroll 16
  for (int ks = 0; ks < QB / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks * sb], acc, 0, 0, 0);
  return acc;
}
// the same over the 32 reduction indices k0 .. k0 + 31 only
__device__ __forceinline__ f32x16 mm_ab_half(f32x16 acc, const float* A, int sa, const float* B, int sb, int ti, int tj,
                                             int l31, int lh, int k0) {
  const float* ap = A + (32 * ti + l31) * sa + k0 + lh;
  const float* bp = B + (k0 + lh) * sb + 32 * tj + l31;
#pragma unroll 16
  for (int ks = 0; ks < QB / 4; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks * sb], acc, 0, 0, 0);
  return acc;
}
// D[i][j] += sum_k A[k][i] B[k][j], k < 64
__device__ __forceinline__ f32x16 mm_atb(f32x16 acc, const float* A, int sa, const float* B, int sb, int ti,
                                         int tj, int l31, int lh) {
  const float* ap = A + lh * sa + 32 * ti + l31;
  const float* bp = B + lh * sb + 32 * tj + l31;
#pragma unroll 16
  for (int ks = 0; ks < QB / 2; ++ks)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks * sa], bp[2 * ks * sb], acc, 0, 0, 0);
  return acc;
}

// S tile (MFMA layout: lane = key column, registers = query rows) + bias + mask -> P (raw scores, or
// exp(s - lse[i]) when lse_row != nullptr); keys that do not exist get -inf / 0
template <int NBINS, bool SELF, bool PARTIAL>
__device__ __forceinline__ void scores_to_lds(const Shared& S, float* P, const f32x16& acc, int ti, int tj, int l31,
                                              int lh, const float* lse_row) {
  const int j = 32 * tj + l31;
  const int kp = S.kpk[j];
  const bool none = PARTIAL && kp == KEY_NONE;
  const int kterm = kp >> 4, kreg = kp & 15;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int qp = S.qpk[i];
    int idx = kterm + (qp >> 4);
    if (!SELF && idx < 0) idx += NBINS;
    float s = acc[r] + S.tab[none ? 0 : idx];
    if (SELF && (qp & 15) != kreg) s -= 100.f;
    if (lse_row)
      P[i * PS + j] = none ? 0.f : __expf(s - lse_row[i]);
    else
      P[i * PS + j] = none ? -INFINITY : s;
  }
}

// this head's column of the relative-position table -> LDS: every load of a thread in ONE batch (clamped index), then the
// stores.  (The loop `tab[k] = table[k * heads + head]` compiles to load, wait, store per trip: four to six dependent memory
// round trips at the start of every workgroup; round 6, found in the ISA.)
template <int NBINS>
__device__ __forceinline__ void stage_table(float* tab, const float* table, int heads, int head, int tid) {
  constexpr int NT = (NBINS + 255) / 256;
  float t[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int k = tid + 256 * i;
    t[i] = table[(k < NBINS ? k : NBINS - 1) * heads + head];
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int k = tid + 256 * i;
    if (k < NBINS) tab[k] = t[i];
  }
}

template <int WS, int KS>
__device__ __forceinline__ void setup_block(const neosr_fattn_desc& d, const Win& w, Shared& S) {
  using G = Geo<WS, KS>;
  const int tid = threadIdx.x;
  if (tid < QB) {
    int tok, reg;
    query_geom<WS>(d, w, w.qb * QB + tid, tok, reg);
    S.qtok[tid] = tok;
    S.qpk[tid] = query_term<WS, KS>(w.qb * QB + tid) * 16 + reg;
  }
  stage_table<G::NBINS>(S.tab, d.rpb_table, d.heads, w.head, tid);
}

// ------------------------------------------------------------------------------------------ forward
template <int WS, int KS>
__global__ __launch_bounds__(256) void flash_wattn_fwd_kernel(const neosr_fattn_desc d) {
  using G = Geo<WS, KS>;
  __shared__ Shared S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();
  const Win w = decode(d, bid, G::NQB);
  const int hd = d.C / d.heads, ld = 3 * d.C, kq = (hd + 1) & ~1;
  const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld);
  [[maybe_unused]] const Rows Rdo = make_rows(d.dout, w.b, d.H * d.W, d.C), Rout = make_rows(d.out, w.b, d.H * d.W, d.C);
  const int n = tid >> 2, part = tid & 3;
  // the thread's query row and the rows of key block 0 are requested before the tables are built (their pixels worked out
  // in registers): one memory round trip for rows and table together instead of two
  float q[8], kr[8], vr[8];
  int ktok, kreg, kterm;
  bool kex;
  {
    int qtok, qreg;
    query_geom<WS>(d, w, w.qb * QB + n, qtok, qreg);
    load_row8(Rqkv, qtok, w.head * hd, hd, part, q);
  }
  key_geom<WS, KS>(d, w, n, ktok, kreg, kterm, kex);
  load_row8(Rqkv, ktok, d.C + w.head * hd, hd, part, kr);
  load_row8(Rqkv, ktok, 2 * d.C + w.head * hd, hd, part, vr);
  setup_block<WS, KS>(d, w, S);
  store_row8(S.Qs, n, part, q, d.scale, hd);
  float m_run = -INFINITY, l_run = 0.f;  // row n, replicated in its 4 threads
  f32x16 o = zero16();                   // waves 0,1: rows 32 wave.., cols d
  for (int kb = 0; kb < G::NKB; ++kb) {
    __syncthreads();  // previous P.V finished with Ks / Vs / P
    store_row8(S.Ks, n, part, kr, 1.f, hd);
    store_row8(S.Vs, n, part, vr, 1.f, hd);
    if (part == 0) S.kpk[n] = kex ? kterm * 16 + kreg : KEY_NONE;
    __syncthreads();
    if (kb + 1 < G::NKB) {
      key_geom<WS, KS>(d, w, (kb + 1) * QB + n, ktok, kreg, kterm, kex);
      load_row8(Rqkv, ktok, d.C + w.head * hd, hd, part, kr);
      load_row8(Rqkv, ktok, 2 * d.C + w.head * hd, hd, part, vr);
    }
    {
      const int ti = wave >> 1, tj = wave & 1;
      const f32x16 acc = mm_abt(zero16(), S.Qs, QS, S.Ks, QS, ti, tj, kq, l31, lh);
      scores_to_lds<G::NBINS, G::SELF, G::NK % QB != 0>(S, S.P, acc, ti, tj, l31, lh, nullptr);
    }
    __syncthreads();
    {  // online softmax over this block's 64 columns: 4 threads per row, 16 columns each
      float* row = S.P + n * PS + part * 16;
      float m = row[0];
#pragma unroll
      for (int c = 1; c < 16; ++c) m = fmaxf(m, row[c]);
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const float m_new = fmaxf(m_run, m);
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float e = __expf(row[c] - m_new);
        row[c] = e;
        s += e;
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      const float alpha = __expf(m_run - m_new);  // 0 on the first block (m_run = -inf)
      l_run = l_run * alpha + s;
      m_run = m_new;
      if (part == 0) S.alpha[n] = alpha;
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= S.alpha[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh];
      o = mm_ab(o, S.P, PS, S.Vs, QS, wave, 0, l31, lh);
    }
  }
  __syncthreads();
  if (part == 0) {
    S.alpha[n] = 1.f / l_run;
    if (d.lse) d.lse[(int64_t)bid * QB + n] = m_run + __logf(l_run);
  }
  __syncthreads();
  if (wave < 2 && l31 < hd) {
    float* out = d.out + w.head * hd + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[(int64_t)S.qtok[i] * d.C] = o[r] * S.alpha[i];
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct SharedBwd {
  float Qs[QB * QS], Ks[QB * QS], Vs[QB * QS], Gs[QB * QS], P[QB * PS], dS[QB * PS];
  float tab[TAB_MAX];
  alignas(16) int qpk[QB];
  alignas(16) float lse[QB], dsum[QB];
  int kpk[QB], qtok[QB], ktok[QB];
#ifdef FATTN_TL
  int tl_on;
#endif
  float bins[31 * 31];  // self-attention dQ pass: this workgroup's share of the relative-position-bias gradient
                        // ((2 WS - 1)^2 bins, WS <= 16)
};

struct BwdWs {  // carve-up of the backward workspace (floats)
  int64_t dsum, ds_full, dense, stage, unf, total;
};
__host__ __device__ inline BwdWs bwd_ws(const neosr_fattn_desc& d) {
  const int64_t bw = (int64_t)d.B * (d.H / d.ws) * (d.W / d.ws), nq = d.ws * d.ws, nk = d.ks * d.ks;
  BwdWs w;
  w.dsum = 0;
  w.ds_full = w.dsum + bw * d.heads * nq;
  w.dense = w.ds_full + bw * d.heads * nq * nk;
  w.stage = w.dense + d.heads * nq * nk;
  w.unf = w.stage + ((bw + 1023) / 1024) * d.heads * nq * nk;
  w.total = w.unf + (d.ks > d.ws ? bw * nk * 2 * d.C : 0) + 64;
  return w;
}

// shared by both backward kernels: P = exp(S - lse), dP = dO V^T, dS = P (dP - D) for the current
// (query block in Qs/Gs, key block in Ks/Vs); leaves P and dS tiles in LDS
// NEED_P = false (the dQ kernel only consumes dS): the P tile is not written
// PARTIAL: the last key block has keys past the window's end (NK % 64 != 0: only the 8 / 12 windows) — without it `none`
// is a compile-time false and the exponentials are straight-line code (hipcc puts `none ? 0 : exp(..)` behind a branch per
// pair of scores)
template <int NBINS, bool SELF, bool PARTIAL, bool NEED_P = true>
__device__ __forceinline__ void recompute_p_ds(SharedBwd& S, int kq, int wave, int l31, int lh) {
  {
    const int ti = wave >> 1, tj = wave & 1;
    f32x16 s = zero16(), dp = zero16();
    mm_abt_pair(s, S.Qs, S.Ks, dp, S.Gs, S.Vs, QS, QS, ti, tj, kq, l31, lh);
    FTL(30, S.tl_on);
    // scores_to_lds wants the forward Shared layout: replicate its body on SharedBwd fields
    const int j = 32 * tj + l31;
    const int kp = S.kpk[j];
    const bool none = PARTIAL && kp == KEY_NONE;
    const int kterm = kp >> 4, kreg = kp & 15;
    // (the per-row terms of 8 scores at a time — packed query geometry, LSE, D — in one batch of LDS reads, then the table
    // values they index in a second: written as one loop over the scores this is four dependent LDS round trips per score;
    // 16 at a time spilled registers in the fused kernel)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int qp[8];
      float ls[8], dsu[8], tv[8];
      // (the 16 rows of a lane are four runs of 4 consecutive queries: one 16-byte read per run and array — 6 LDS reads per
      // half instead of 24)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int i0 = 32 * ti + 8 * (2 * h + g) + 4 * lh;
        const i32x4_t q4 = *reinterpret_cast<const i32x4_t*>(&S.qpk[i0]);
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&S.lse[i0]), d4 = *reinterpret_cast<const f32x4*>(&S.dsum[i0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qp[4 * g + e] = q4[e];
          ls[4 * g + e] = l4[e];
          dsu[4 * g + e] = d4[e];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int idx = kterm + (qp[q] >> 4);
        if (!SELF && idx < 0) idx += NBINS;
        tv[q] = S.tab[none ? 0 : idx];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 8 * h + q;
        const int i = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = s[r] + tv[q];
        v -= (SELF && (qp[q] & 15) != kreg) ? 100.f : 0.f;
        // dS = P (dP - D) with D = rowsum(dO . O) already in LDS: finished in the tile's own registers (a separate row
        // pass over the two LDS tiles cost 32 reads + 16 writes per thread and one more barrier per key block)
        const float p = none ? 0.f : __expf(v - ls[q]);
        if (NEED_P) S.P[i * PS + j] = p;
        S.dS[i * PS + j] = p * (dp[r] - dsu[q]);
      }
    }
    FTL(31, S.tl_on);
  }
  __syncthreads();
}

// Bias gradient of one 64-query x 64-key dS tile, owner-computes (kernel (A) explains the geometry): thread `tid` sums the
// pairs of ITS bin (dyi, dx) of the tile in a fixed order.  The WS reads of a window-row pair go out as ONE batch in front
// of their additions (sched_barrier): left to itself hipcc reuses one register pair for all of them — 32 LDS round trips in
// a row, each waited for with lgkmcnt(0), ~1.5 us of a 7.5 us tile (round 6, found in the ISA).  Same additions, same order.
template <int WS>
__device__ __forceinline__ float tile_bin_sum(const float* dS, int tid, int& dyi, int& dx) {
  constexpr int RQ = QB / WS, NB1 = 2 * WS - 1;
  dyi = tid / NB1;
  dx = tid % NB1 - (WS - 1);
  const int xi0 = dx > 0 ? dx : 0, xj0 = dx < 0 ? -dx : 0, len = WS - (dx < 0 ? -dx : dx);
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < RQ; ++a) {
    const int b = a - (dyi - (RQ - 1));  // key row of the tile paired with query row a
    const bool oky = b >= 0 && b < RQ;
    const int bc = b < 0 ? 0 : (b >= RQ ? RQ - 1 : b);
    const float* base = dS + (a * WS + xi0) * PS + bc * WS + xj0;
    float v[WS];
#pragma unroll
    for (int t = 0; t < WS; ++t) v[t] = base[t * (PS + 1)];
    __builtin_amdgcn_sched_barrier(0);
    float sa = 0.f;
#pragma unroll
    for (int t = 0; t < WS; ++t) sa += t < len ? v[t] : 0.f;  // (a select, not a 0 / 1 factor: past the diagonal the read may hit anything)
    s += oky ? sa : 0.f;
    __builtin_amdgcn_sched_barrier(0);
  }
  return s;
}

// (A) one workgroup per (window, head, query block): dQ, D = rowsum(dO * O), and the dS tiles
template <int WS, int KS>
__global__ __launch_bounds__(256, KS > WS ? 2 : 1) void flash_wattn_bwd_dq_kernel(const neosr_fattn_desc d, const BwdWs ws) {
  using G = Geo<WS, KS>;
  __shared__ SharedBwd S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();
  const Win w = decode(d, bid, G::NQB);
  const int hd = d.C / d.heads, ld = 3 * d.C, kq = (hd + 1) & ~1;
  const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld);
  [[maybe_unused]] const Rows Rdo = make_rows(d.dout, w.b, d.H * d.W, d.C), Rout = make_rows(d.out, w.b, d.H * d.W, d.C);
  const int n = tid >> 2, part = tid & 3;
  // the thread's query rows (q, dO, O) and the rows of key block 0 are requested before the tables are built (their pixels
  // worked out in registers): one memory round trip for rows, table and LSE together instead of two
  float q[8], g[8], o[8], kr[8], vr[8];
  int ktok, kreg, kterm;
  bool kex;
  {
    int tok, reg;
    query_geom<WS>(d, w, w.qb * QB + n, tok, reg);
    load_row8(Rqkv, tok, w.head * hd, hd, part, q);
    load_row8(Rdo, tok, w.head * hd, hd, part, g);
    load_row8(Rout, tok, w.head * hd, hd, part, o);
  }
  key_geom<WS, KS>(d, w, n, ktok, kreg, kterm, kex);
  load_row8(Rqkv, ktok, d.C + w.head * hd, hd, part, kr);
  load_row8(Rqkv, ktok, 2 * d.C + w.head * hd, hd, part, vr);
  if (tid < QB) {
    int tok, reg;
    query_geom<WS>(d, w, w.qb * QB + tid, tok, reg);
    S.qtok[tid] = tok;
    S.qpk[tid] = query_term<WS, KS>(w.qb * QB + tid) * 16 + reg;
    S.lse[tid] = d.lse[(int64_t)bid * QB + tid];
  }
  stage_table<G::NBINS>(S.tab, d.rpb_table, d.heads, w.head, tid);
  if (G::SELF)
    for (int k = tid; k < G::NBINS; k += 256) S.bins[k] = 0.f;
  {
    store_row8(S.Qs, n, part, q, d.scale, hd);
    store_row8(S.Gs, n, part, g, 1.f, hd);
    float ds = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ds += part * 8 + e < hd ? g[e] * o[e] : 0.f;   // (columns past hd: the next head's values)
    ds += __shfl_xor(ds, 1, 64);
    ds += __shfl_xor(ds, 2, 64);
    if (part == 0) {
      S.dsum[n] = ds;
      d.workspace[ws.dsum + (int64_t)bid * QB + n] = ds;
    }
  }
  f32x16 dq = zero16();
  // dS dump: [(b, window, head)][query 256][key NK]
  float* dump = d.workspace + ws.ds_full + ((int64_t)(bid / G::NQB) * G::NQ + w.qb * QB) * G::NK;
  for (int kb = 0; kb < G::NKB; ++kb) {
    __syncthreads();
    store_row8(S.Ks, n, part, kr, 1.f, hd);
    store_row8(S.Vs, n, part, vr, 1.f, hd);
    if (part == 0) S.kpk[n] = kex ? kterm * 16 + kreg : KEY_NONE;
    __syncthreads();
    if (kb + 1 < G::NKB) {
      key_geom<WS, KS>(d, w, (kb + 1) * QB + n, ktok, kreg, kterm, kex);
      load_row8(Rqkv, ktok, d.C + w.head * hd, hd, part, kr);
      load_row8(Rqkv, ktok, 2 * d.C + w.head * hd, hd, part, vr);
    }
    recompute_p_ds<G::NBINS, G::SELF, G::NK % QB != 0, false>(S, kq, wave, l31, lh);
    if (G::SELF) {
      // bias gradient of this 64-query x 64-key tile, owner-computes: the tile is RQ x RQ window rows of WS (4 x 4 rows of
      // 16, or the whole 8 x 8 window), so it touches (2 RQ - 1) x (2 WS - 1) bins (dy = yi - yj, dx = xi - xj) and thread
      // t sums the pairs of ITS bin in a fixed order (64 predicated LDS reads, out-of-window pairs read element 0 and
      // add 0) into the workgroup's accumulator — no dS dump to HBM (it was 256 x 256 floats per (window, head): 100 MB
      // per call at B = 4)
      constexpr int RQ = QB / WS, NB1 = 2 * WS - 1;
      static_assert((2 * RQ - 1) * NB1 <= 256, "one bin of the tile per thread");
      if (tid < (2 * RQ - 1) * NB1) {
        // The pairs of bin (dy, dx) inside window rows (a, b) lie on ONE diagonal of that WS x WS sub-block:
        // (xi, xj) = (xi0 + t, xj0 + t), t < WS - |dx| — a fixed start address per (thread, a) and a constant stride
        // PS + 1, so the reads carry immediate offsets and validity is ONE predicate per t, shared by all a (t < len),
        // plus one per a — 16 + 4 masks instead of the 64 per-pair masks that spilled 170 SGPRs in this kernel; a key
        // row outside the tile is clamped for the address and its sum discarded.
        int dyi, dx;
        const float s = tile_bin_sum<WS>(S.dS, tid, dyi, dx);
        const int dy = RQ * (w.qb - kb) - (RQ - 1) + dyi;
        if (dy > -WS && dy < WS) S.bins[(dy + WS - 1) * NB1 + dx + WS - 1] += s;
      }
    } else {
      // dump this dS tile (rows n, 16 columns per thread) for the bias gradient.  (Round 4 tried the owner-computes bin
      // walk of the self-attention form here — the index is linear in (yj - yi, xj - xi), (RQ + 3) x 39 bins per tile in
      // two passes — to get rid of the 226 MB dump per call at B = 4 (3.9x the kernel's algorithmic traffic): the call
      // went 501 -> 597 us.  Sixty-four predicated LDS reads with per-element key-row arithmetic cost more than writing
      // and re-reading the dump at HBM rate; the dump stays.)
      const float* gr = S.dS + n * PS + part * 16;
      float* o = dump + (int64_t)n * G::NK + kb * QB + part * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (kb * QB + part * 16 + c < G::NK) o[c] = gr[c];
    }
    // dQ[i][d] += sum_j dS[i][j] K[j][d]: all four waves — wave (tile = wave & 1, key half = wave >> 1) takes 32 of the
    // block's 64 keys (before: the two tiles on waves 0, 1 while waves 2, 3 idled); the halves meet once, at the end
    dq = mm_ab_half(dq, S.dS, PS, S.Ks, QS, wave & 1, 0, l31, lh, 32 * (wave >> 1));
  }
  __syncthreads();   // last block's products have read P / dS
  if (G::SELF) {  // partial bins of (window, query block): row (bw index, qb) of a [rows][bin][head] matrix
    float* row = d.workspace + ws.ds_full +
                 ((int64_t)(bid / (d.heads * G::NQB)) * G::NQB + w.qb) * G::NBINS * d.heads + w.head;
    for (int k = tid; k < G::NBINS; k += 256) row[(int64_t)k * d.heads] = S.bins[k];
  }
  if (wave >= 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) S.P[(wave - 2) * 1024 + r * 64 + lane] = dq[r];
  }
  __syncthreads();
  if (wave < 2 && l31 < hd) {
    float* g = d.dqkv + w.head * hd + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      g[(int64_t)S.qtok[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh] * ld] =
          (dq[r] + S.P[wave * 1024 + r * 64 + lane]) * d.scale;
  }
}

// (B) one workgroup per (window, head, key block): dK, dV over the 4 query blocks
template <int WS, int KS>
__global__ __launch_bounds__(256) void flash_wattn_bwd_dkv_kernel(const neosr_fattn_desc d, const BwdWs ws) {
  using G = Geo<WS, KS>;
  __shared__ SharedBwd S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();
  Win w = decode(d, bid, G::NKB);
  const int kb = w.qb;  // decode()'s innermost index is the key block here
  const int hd = d.C / d.heads, ld = 3 * d.C, kq = (hd + 1) & ~1;
  const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld);
  [[maybe_unused]] const Rows Rdo = make_rows(d.dout, w.b, d.H * d.W, d.C), Rout = make_rows(d.out, w.b, d.H * d.W, d.C);
  const int n = tid >> 2, part = tid & 3;
  const int64_t wh = bid / G::NKB;  // (b, window, head)
  // the key block's rows and the first query block's q / dO rows are requested before the tables are built (every thread
  // works out its own rows' pixels): one memory round trip for rows and table together
  float kr[8], vr[8], qn[8], gn[8];
  {
    int tok, reg, kterm;
    bool ex;
    key_geom<WS, KS>(d, w, kb * QB + n, tok, reg, kterm, ex);
    load_row8(Rqkv, tok, d.C + w.head * hd, hd, part, kr);
    load_row8(Rqkv, tok, 2 * d.C + w.head * hd, hd, part, vr);
    query_geom<WS>(d, w, n, tok, reg);
    load_row8(Rqkv, tok, w.head * hd, hd, part, qn);
    load_row8(Rdo, tok, w.head * hd, hd, part, gn);
  }
  if (tid < QB) {
    int tok, reg, kterm;
    bool ex;
    key_geom<WS, KS>(d, w, kb * QB + tid, tok, reg, kterm, ex);
    S.ktok[tid] = tok;
    S.kpk[tid] = ex ? kterm * 16 + reg : KEY_NONE;
  }
  stage_table<G::NBINS>(S.tab, d.rpb_table, d.heads, w.head, tid);
  store_row8(S.Ks, n, part, kr, 1.f, hd);
  store_row8(S.Vs, n, part, vr, 1.f, hd);
  f32x16 acc = zero16();  // waves 0,1: dV rows 32 wave..; waves 2,3: dK rows 32 (wave-2)..
  // (the next query block's q / dO rows travel in registers while the current block is consumed)
  for (int qb = 0; qb < G::NQB; ++qb) {
    __syncthreads();  // previous products finished with Qs / Gs / P / dS
    w.qb = qb;
    if (tid < QB) {
      int tok, reg;
      query_geom<WS>(d, w, qb * QB + tid, tok, reg);
      S.qtok[tid] = tok;
      S.qpk[tid] = query_term<WS, KS>(qb * QB + tid) * 16 + reg;
      S.lse[tid] = d.lse[(wh * G::NQB + qb) * QB + tid];
      S.dsum[tid] = d.workspace[ws.dsum + (wh * G::NQB + qb) * QB + tid];
    }
    store_row8(S.Qs, n, part, qn, d.scale, hd);
    store_row8(S.Gs, n, part, gn, 1.f, hd);
    __syncthreads();
    if (qb + 1 < G::NQB) {
      int tok, reg;
      query_geom<WS>(d, w, (qb + 1) * QB + n, tok, reg);
      load_row8(Rqkv, tok, w.head * hd, hd, part, qn);
      load_row8(Rdo, tok, w.head * hd, hd, part, gn);
    }
    recompute_p_ds<G::NBINS, G::SELF, G::NK % QB != 0>(S, kq, wave, l31, lh);
    if (wave < 2)
      acc = mm_atb(acc, S.P, PS, S.Gs, QS, wave, 0, l31, lh);       // dV[j][d] += sum_i P[i][j] dO[i][d]
    else
      acc = mm_atb(acc, S.dS, PS, S.Qs, QS, wave - 2, 0, l31, lh);  // dK[j][d] += sum_i dS[i][j] (scale q)[i][d]
  }
  if (l31 < hd) {
    const int which = wave < 2 ? 2 : 1;  // v : k
    const int jt = wave & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (G::SELF) {
        d.dqkv[(int64_t)S.ktok[j] * ld + which * d.C + w.head * hd + l31] = acc[r];
      } else if (kb * QB + j < G::NK) {
        const int64_t bwin = wh / d.heads;
        d.workspace[ws.unf + ((bwin * G::NK + kb * QB + j) * 2 + (which - 1)) * d.C + w.head * hd + l31] = acc[r];
      }
    }
  }
}


// (A+B) self-attention form, ONE pass: one workgroup per (window, head) walks the query blocks (outer) and key blocks
// (inner); every 64 x 64 tile's S = Q K^T, dP = dO V^T, P and dS are formed ONCE and feed all three gradients — dQ of the
// query block (one accumulator, as in kernel (A)), dV += P^T dO and dK += dS^T Q of the key block (NKB accumulators per
// wave, resident across the outer loop: waves 0, 1 hold dV, waves 2, 3 dK).  5 products per tile instead of the 7 of the
// two recompute passes (A) + (B), half the Q / K / V / dO tile loads, half the exponentials; no rowsum(dO . O) round trip
// through the workspace.  Every output accumulates its tiles in the order the two-pass kernels use (dQ over key blocks,
// dK / dV over query blocks; bias bins per query block, written as the same [window, query block] partial rows):
// bit-identical results (tests/test_hip_hat.py).  LDS as kernel (A): two workgroups per CU.
template <int WS, int KS>
__global__ __launch_bounds__(256, 2) void flash_wattn_bwd_fused_kernel(const neosr_fattn_desc d, const BwdWs ws) {
  using G = Geo<WS, KS>;
  static_assert(G::SELF, "the overlapping form folds dK / dV over windows: two-pass kernels");
  __shared__ SharedBwd S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int bid = xcd_bid();            // (b, window, head)
  Win w = decode(d, bid, 1);
  const int hd = d.C / d.heads, ld = 3 * d.C, kq = (hd + 1) & ~1;
  const Rows Rqkv = make_rows(d.qkv, w.b, d.H * d.W, ld);
  [[maybe_unused]] const Rows Rdo = make_rows(d.dout, w.b, d.H * d.W, d.C), Rout = make_rows(d.out, w.b, d.H * d.W, d.C);
  const int n = tid >> 2, part = tid & 3;
  FTL(0, true);
  stage_table<G::NBINS>(S.tab, d.rpb_table, d.heads, w.head, tid);
  // The key rows of the window are the same for every query block: this thread's row offset and the packed bias / mask
  // term of its key in each key block are worked out ONCE (the geometry + address arithmetic of the next block's prefetch
  // took ~700 of a tile's ~13 000 cycles; tools/timeline_fattn.py)
  unsigned koff[G::NKB];
  int kpkr[G::NKB];
  const bool kal8 = row_al8(Rqkv, d.C + w.head * hd, hd) && (d.C & 1) == 0;
#pragma unroll
  for (int c = 0; c < G::NKB; ++c) {
    int tok, reg, kterm;
    bool ex;
    key_geom<WS, KS>(d, w, c * QB + n, tok, reg, kterm, ex);
    koff[c] = row_off(Rqkv, tok, d.C + w.head * hd, part);
    kpkr[c] = ex ? kterm * 16 + reg : KEY_NONE;
  }
  auto key_rows = [&](int kb, float (&kr)[8], float (&vr)[8], int& kp) {   // (kb is a run-time index: selects, not scratch)
    unsigned off = koff[0];
    kp = kpkr[0];
#pragma unroll
    for (int c = 1; c < G::NKB; ++c) {
      off = kb == c ? koff[c] : off;
      kp = kb == c ? kpkr[c] : kp;
    }
    load_row8_at(Rqkv, off, kal8, kr);
    load_row8_at(Rqkv, off + (unsigned)d.C * 4u, kal8, vr);
  };
  f32x16 dkv[G::NKB];   // waves 0, 1: dV rows 32 wave..; waves 2, 3: dK rows 32 (wave - 2)..  of key block kb
#pragma unroll
  for (int kb = 0; kb < G::NKB; ++kb) dkv[kb] = zero16();
  for (int qb = 0; qb < G::NQB; ++qb) {
    w.qb = qb;
    // this thread's query row (q, dO, O) is requested before the barriers of the block's set-up: its pixel is worked out in
    // registers, not read back from the table the set-up writes
    float q[8], g[8], o[8];
    {
      int tok, reg;
      query_geom<WS>(d, w, qb * QB + n, tok, reg);
      load_row8(Rqkv, tok, w.head * hd, hd, part, q);
      load_row8(Rdo, tok, w.head * hd, hd, part, g);
      load_row8(Rout, tok, w.head * hd, hd, part, o);
    }
    __syncthreads();   // the previous query block's epilogue has read S.P / S.qtok
    if (tid < QB) {
      int tok, reg;
      query_geom<WS>(d, w, qb * QB + tid, tok, reg);
      S.qtok[tid] = tok;
      S.qpk[tid] = query_term<WS, KS>(qb * QB + tid) * 16 + reg;
      S.lse[tid] = d.lse[((int64_t)bid * G::NQB + qb) * QB + tid];
    }
    for (int k = tid; k < G::NBINS; k += 256) S.bins[k] = 0.f;
    __syncthreads();
    {
      store_row8(S.Qs, n, part, q, d.scale, hd);
      store_row8(S.Gs, n, part, g, 1.f, hd);
      float ds = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ds += part * 8 + e < hd ? g[e] * o[e] : 0.f;   // (columns past hd: the next head's values)
      ds += __shfl_xor(ds, 1, 64);
      ds += __shfl_xor(ds, 2, 64);
      if (part == 0) S.dsum[n] = ds;
    }
    float kr[8], vr[8];
    int kp;
    key_rows(0, kr, vr, kp);
    f32x16 dq = zero16();
    // (the loop stays rolled — unrolled, its four copies of the bias-bin walk spilled 29 VGPRs at the 256 a wave may hold
    // with two workgroups per CU; only the accumulation into the key block's own registers is spelled out per block)
#pragma unroll 1
    for (int kb = 0; kb < G::NKB; ++kb) {
#ifdef FATTN_TL
      if (tid == 0) S.tl_on = qb == 1 && kb == 1;
#endif
      FTL(1, qb == 1 && kb == 1);
      FTL(9, qb == 1 && kb == 2);
      __syncthreads();
      FTL(2, qb == 1 && kb == 1);
      store_row8(S.Ks, n, part, kr, 1.f, hd);
      store_row8(S.Vs, n, part, vr, 1.f, hd);
      if (part == 0) S.kpk[n] = kp;
      __syncthreads();
      FTL(3, qb == 1 && kb == 1);
      if (kb + 1 < G::NKB) key_rows(kb + 1, kr, vr, kp);
      FTL(4, qb == 1 && kb == 1);
      recompute_p_ds<G::NBINS, G::SELF, G::NK % QB != 0, true>(S, kq, wave, l31, lh);
      FTL(5, qb == 1 && kb == 1);
      {  // bias gradient of the tile (see kernel (A))
        constexpr int RQ = QB / WS, NB1 = 2 * WS - 1;
        static_assert((2 * RQ - 1) * NB1 <= 256, "one bin of the tile per thread");
        if (tid < (2 * RQ - 1) * NB1) {
          int dyi, dx;
          const float s = tile_bin_sum<WS>(S.dS, tid, dyi, dx);
          const int dy = RQ * (qb - kb) - (RQ - 1) + dyi;
          if (dy > -WS && dy < WS) S.bins[(dy + WS - 1) * NB1 + dx + WS - 1] += s;
        }
      }
      FTL(6, qb == 1 && kb == 1);
      {
        // waves 0, 1: dV[j][d] += sum_i P[i][j] dO[i][d];  waves 2, 3: dK[j][d] += sum_i dS[i][j] (scale q)[i][d]
        const float* A = wave < 2 ? S.P : S.dS;
        const float* Bm = wave < 2 ? S.Gs : S.Qs;
        const int ti = wave & 1;
#pragma unroll
        for (int c = 0; c < G::NKB; ++c)
          if (kb == c) dkv[c] = mm_atb(dkv[c], A, PS, Bm, QS, ti, 0, l31, lh);
      }
      FTL(7, qb == 1 && kb == 1);
      dq = mm_ab_half(dq, S.dS, PS, S.Ks, QS, wave & 1, 0, l31, lh, 32 * (wave >> 1));
      FTL(8, qb == 1 && kb == 1);
    }
    FTL(10 + qb, true);
    __syncthreads();   // the last block's products have read P / dS
    {  // partial bins of (window, query block): row (bw index, qb) of a [rows][bin][head] matrix
      float* row = d.workspace + ws.ds_full + ((int64_t)(bid / d.heads) * G::NQB + qb) * G::NBINS * d.heads + w.head;
      for (int k = tid; k < G::NBINS; k += 256) row[(int64_t)k * d.heads] = S.bins[k];
    }
    if (wave >= 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S.P[(wave - 2) * 1024 + r * 64 + lane] = dq[r];
    }
    __syncthreads();
    if (wave < 2 && l31 < hd) {
      float* g = d.dqkv + w.head * hd + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        g[(int64_t)S.qtok[32 * wave + (r & 3) + 8 * (r >> 2) + 4 * lh] * ld] =
            (dq[r] + S.P[wave * 1024 + r * 64 + lane]) * d.scale;
    }
  }
  FTL(20, true);
  // dK / dV of every key block
  if (l31 < hd) {
    const int which = wave < 2 ? 2 : 1;  // v : k
    const int jt = wave & 1;
#pragma unroll
    for (int kb = 0; kb < G::NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kb * QB + 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        int tok, reg;
        query_geom<WS>(d, w, j, tok, reg);   // (self-attention: key j is query token j)
        d.dqkv[(int64_t)tok * ld + which * d.C + w.head * hd + l31] = dkv[kb][r];
      }
    }
  }
}

// OCAB: nn.Unfold's adjoint — every pixel sums the k / v gradients of the (up to 4) overlapping
// windows that contain it, in (Wy, Wx) order
template <int WS, int KS>
__global__ __launch_bounds__(256) void fold_dkv_kernel(const neosr_fattn_desc d, const BwdWs ws) {
  using G = Geo<WS, KS>;
  const int nWy = d.H / WS, nWx = d.W / WS, C2 = 2 * d.C;
  const int64_t total = (int64_t)d.B * d.H * d.W * C2;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c = (int)(e % C2);
    int64_t p = e / C2;
    const int x = (int)(p % d.W);
    p /= d.W;
    const int y = (int)(p % d.H), b = (int)(p / d.H);
    float s = 0.f;
    for (int Wy = max(0, (y - (WS + G::PAD) + WS) / WS); Wy < nWy && Wy * WS - G::PAD <= y; ++Wy) {
      const int yk = y - Wy * WS + G::PAD;
      if (yk < 0 || yk >= KS) continue;
      for (int Wx = max(0, (x - (WS + G::PAD) + WS) / WS); Wx < nWx && Wx * WS - G::PAD <= x; ++Wx) {
        const int xk = x - Wx * WS + G::PAD;
        if (xk < 0 || xk >= KS) continue;
        const int64_t bwin = ((int64_t)b * nWy + Wy) * nWx + Wx;
        s += d.workspace[ws.unf + ((bwin * G::NK + yk * KS + xk) * 2 + c / d.C) * d.C + c % d.C];
      }
    }
    d.dqkv[((int64_t)(b * d.H + y) * d.W + x) * 3 * d.C + d.C + c] = s;
  }
}

// d_table[bin][head] (+)= sum over the (query, key) pairs mapped to `bin` of the dense bias gradient:
// one wave per (bin, head), each lane takes 4 of the 256 query positions, butterfly-free fixed-order
// shuffle reduction
template <int WS, int KS>
__global__ __launch_bounds__(256) void rpb_bins_kernel(const float* __restrict__ dense, float* __restrict__ dtab,
                                                       int heads, int accumulate) {
  using G = Geo<WS, KS>;
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= G::NBINS * heads) return;
  const int bin = e / heads, head = e - bin * heads;
  // key - query offsets (dy, dx) of this bin
  int dy, dx;
  if (G::SELF) {  // bin = (yi - yj + WS-1) L + (xi - xj + WS-1)
    dy = -(bin / G::L - (WS - 1));
    dx = -(bin % G::L - (WS - 1));
  } else {  // unwrap the negative-index wrap-around, then t = (yj-yi+WS-KS+1) L + (xj-xi+WS-KS+1)
    const int hi = (WS - 1 + WS - KS + 1) * G::L + (WS - 1 + WS - KS + 1) + (KS - WS) * (G::L + 1);  // max t
    const int t = bin <= hi ? bin : bin - G::NBINS;
    const int off = KS - 2;  // digits (xj - xi + WS-KS+1) lie in [2-KS, WS-1]: + off -> [0, L)
    const int u = (t + off * G::L + off) / G::L, v = (t + off * G::L + off) % G::L;
    dy = u - off - (WS - KS + 1);
    dx = v - off - (WS - KS + 1);
  }
  const float* base = dense + (int64_t)head * G::NQ * G::NK;
  float s = 0.f;
  for (int q = lane; q < G::NQ; q += 64) {
    const int yi = q / WS, xi = q % WS;
    const int yj = yi + dy, xj = xi + dx;
    if (yj >= 0 && yj < KS && xj >= 0 && xj < KS) s += base[(int64_t)q * G::NK + yj * KS + xj];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) dtab[e] = accumulate ? dtab[e] + s : s;
}

#ifdef FATTN_TL
}  // namespace
extern "C" int neosr_debug_fattn_timeline(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fattn_tl), 64 * 8) == hipSuccess ? 0 : 1;
}
namespace {
#endif
int g_fattn_fused = -1;   // -1: read NEOSR_AMD_FATTN_FUSED on first use (default on)
bool fattn_fused_on() {
  if (g_fattn_fused < 0) {
    const char* e = getenv("NEOSR_AMD_FATTN_FUSED");
    g_fattn_fused = (e && e[0] == '0') ? 0 : 1;
  }
  return g_fattn_fused == 1;
}

template <int WS, int KS>
int launch_bwd(const neosr_fattn_desc& d, hipStream_t st) {
  using G = Geo<WS, KS>;
  const BwdWs ws = bwd_ws(d);
  const int bw = d.B * (d.H / WS) * (d.W / WS);
  const bool prof = neosr_prof_on();
  const double tok = (double)d.B * d.H * d.W;  // dP, dV, dQ, dK over KS*KS keys per query (recompute not counted)
  if (prof) neosr_prof_begin(NEOSR_PROF_ATTN_BWD, (void*)st, 8.0 * KS * KS * tok * d.C, 4.0 * tok * 8 * d.C);
  // self-attention: the one-pass kernel (NEOSR_AMD_FATTN_FUSED=0 keeps the two recompute passes: same bits)
  if constexpr (G::SELF) {
    if (fattn_fused_on()) {
      hipLaunchKernelGGL((flash_wattn_bwd_fused_kernel<WS, KS>), dim3(bw * d.heads), dim3(256), 0, st, d, ws);
    } else {
      hipLaunchKernelGGL((flash_wattn_bwd_dq_kernel<WS, KS>), dim3(bw * d.heads * G::NQB), dim3(256), 0, st, d, ws);
      NEOSR_LAUNCH_CHECK();
      hipLaunchKernelGGL((flash_wattn_bwd_dkv_kernel<WS, KS>), dim3(bw * d.heads * G::NKB), dim3(256), 0, st, d, ws);
    }
  } else {
    hipLaunchKernelGGL((flash_wattn_bwd_dq_kernel<WS, KS>), dim3(bw * d.heads * G::NQB), dim3(256), 0, st, d, ws);
    NEOSR_LAUNCH_CHECK();
    hipLaunchKernelGGL((flash_wattn_bwd_dkv_kernel<WS, KS>), dim3(bw * d.heads * G::NKB), dim3(256), 0, st, d, ws);
  }
  if (prof) neosr_prof_end((void*)st);
  NEOSR_LAUNCH_CHECK();
  if (!G::SELF) {
    const int64_t total = (int64_t)d.B * d.H * d.W * 2 * d.C;
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL((fold_dkv_kernel<WS, KS>), dim3(g), dim3(256), 0, st, d, ws);
    NEOSR_LAUNCH_CHECK();
  }
  if (G::SELF) {  // bias gradient: fixed-order column sums of the per-(window, query block) bin partials
    const int cols = G::NBINS * d.heads;
    // accumulate_rpb == 2: leave the [bw NQB][cols] partials (at float offset bw heads ws^2 of the workspace) for a
    // batched reduction (neosr_colsum_many); returns -(rows)
    if (d.accumulate_rpb == 2) return -(bw * G::NQB);
    return neosr_colsum(d.workspace + ws.ds_full, d.d_rpb_table, d.workspace + ws.stage, bw * G::NQB, cols, cols,
                        d.accumulate_rpb, (void*)st);
  }
  // bias gradient: sum the dS dump over (batch, window), then gather per table row
  const int cols = d.heads * G::NQ * G::NK;
  if (int rc = neosr_colsum(d.workspace + ws.ds_full, d.workspace + ws.dense, d.workspace + ws.stage, bw, cols,
                            cols, 0, (void*)st))
    return rc;
  hipLaunchKernelGGL((rpb_bins_kernel<WS, KS>), dim3(ceil_div(G::NBINS * d.heads, 4)), dim3(256), 0, st,
                     d.workspace + ws.dense, d.d_rpb_table, d.heads, d.accumulate_rpb);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

int check(const neosr_fattn_desc* d) {
  NEOSR_CHECK(d && d->qkv && d->rpb_table, "flash_window_attention: null tensor");
  NEOSR_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->heads > 0, "flash_window_attention: bad geometry");
  NEOSR_CHECK((d->ws == 16 && (d->ks == 16 || d->ks == 24)) || (d->ws == 8 && (d->ks == 8 || d->ks == 12)),
              "flash_window_attention: window 16 or 8 with the same key window (self) or 1.5x (overlapping) only "
              "(got %d / %d)", d->ws, d->ks);
  NEOSR_CHECK(d->H % d->ws == 0 && d->W % d->ws == 0, "flash_window_attention: H, W must be multiples of the window");
  NEOSR_CHECK(d->C % d->heads == 0 && d->C / d->heads <= 32, "flash_window_attention: head_dim must be <= 32");
  NEOSR_CHECK((int64_t)d->H * d->W * 3 * d->C * 4 <= (int64_t)ROW_DEAD && d->C / d->heads >= 2,
              "flash_window_attention: one sample's qkv rows must stay under 3 GiB (32-bit row offsets)");
  NEOSR_CHECK(d->shift >= 0 && d->shift < d->ws && (d->ks == d->ws || d->shift == 0),
              "flash_window_attention: bad shift");
  return 0;
}

}  // namespace

extern "C" int neosr_flash_window_attention_fwd(const neosr_fattn_desc* d, void* stream) {
  if (int rc = check(d)) return rc;
  NEOSR_CHECK(d->out, "flash_window_attention_fwd: out missing");
  const int nblk = d->B * (d->H / d->ws) * (d->W / d->ws) * d->heads * (d->ws * d->ws / QB);
  const bool prof = neosr_prof_on();
  if (prof)
    neosr_prof_begin(NEOSR_PROF_ATTN_FWD, stream, 4.0 * d->ks * d->ks * (double)d->B * d->H * d->W * d->C,
                     4.0 * (double)d->B * d->H * d->W * 4 * d->C);
  const dim3 grid(nblk), blk(256);
  hipStream_t st = (hipStream_t)stream;
  static const bool streaming = getenv("NEOSR_FATTN_STREAMING") != nullptr;  // A/B switch
  if (neosr_wattn::wave16_ok(*d) && !streaming) neosr_wattn::launch16_fwd(*d, stream);
  else if (d->ws == 16 && d->ks == 16) hipLaunchKernelGGL((flash_wattn_fwd_kernel<16, 16>), grid, blk, 0, st, *d);
  else if (d->ws == 16) hipLaunchKernelGGL((flash_wattn_fwd_kernel<16, 24>), grid, blk, 0, st, *d);
  else if (d->ks == 8) hipLaunchKernelGGL((flash_wattn_fwd_kernel<8, 8>), grid, blk, 0, st, *d);
  else hipLaunchKernelGGL((flash_wattn_fwd_kernel<8, 12>), grid, blk, 0, st, *d);
  if (prof) neosr_prof_end(stream);
  NEOSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t neosr_flash_window_attention_workspace_bytes(const neosr_fattn_desc* d) {
  if (!d || check(d) || d->B <= 0) return 0;
  return bwd_ws(*d).total * 4;
}

extern "C" int neosr_set_fattn_fused(int on) {
  const int prev = fattn_fused_on() ? 1 : 0;
  g_fattn_fused = on ? 1 : 0;
  return prev;
}

extern "C" int neosr_flash_window_attention_bwd(const neosr_fattn_desc* d, void* stream) {
  if (int rc = check(d)) return rc;
  NEOSR_CHECK(d->out && d->dout && d->dqkv && d->lse && d->d_rpb_table && d->workspace,
              "flash_window_attention_bwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  if (d->ws == 16) return d->ks == 16 ? launch_bwd<16, 16>(*d, st) : launch_bwd<16, 24>(*d, st);
  return d->ks == 8 ? launch_bwd<8, 8>(*d, st) : launch_bwd<8, 12>(*d, st);
}
