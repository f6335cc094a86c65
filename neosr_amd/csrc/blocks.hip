// blocks.hip — host-side plans for ONE transformer block of the SwinIR / HAT generators: every kernel of the block's
// forward (or backward) enqueued by a single C-ABI call, from a descriptor of plain pointers.
//
//   SwinIR  SwinTransformerBlock   neosr/archs/swinir_arch.py:231-392   norm1 -> qkv -> (shifted-)window attention ->
//           proj + DropPath + shortcut -> norm2 -> fc1 -> GELU -> fc2 + DropPath + shortcut
//   HAT     OCAB                   neosr/archs/hat_arch.py:393-515      the same chain with the overlapping cross-attention
//   HAT     HAB                    neosr/archs/hat_arch.py:218-350      the same chain + the CAB branch on norm1's output
//                                                                       (conv3x3 -> GELU -> conv3x3 -> channel attention,
//                                                                       scaled by conv_scale) added in front of norm2
// The reference dispatches ~40 ATen ops per block and direction; rounds 1-3 composed the block from Python, one
// autograd.Function + one ctypes call per fused op (6-9 forward, 12-20 backward launches per block, ~6 000 host dispatches
// per hat_l step: host enqueue 83 of 102 ms).  Here a block is TWO calls per step.  The plans call the library's own entry
// points (neosr_layernorm_*, neosr_gemm, neosr_*window_attention_*, neosr_conv3x3*, neosr_colsum_many) with exactly the
// descriptors the Python fronts (hip/transformer.py, hip/layers.py) build, in the same order: bit-identical results.
// No allocation, no synchronisation: activations kept for backward live in a caller-owned `save` buffer, temporaries of
// the backward pass in a caller-owned workspace (sizes from the *_floats functions).
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>

#include "common.h"
#include "../../include/neosr_amd.h"

namespace {

struct Carve {
  float* base;
  int64_t off = 0;
  explicit Carve(float* b) : base(b) {}
  float* take(int64_t n) {
    float* p = base ? base + off : nullptr;
    off += (n + 63) & ~(int64_t)63;   // 256-byte granules: every piece satisfies the kernels' 16-byte alignment
    return p;
  }
};

struct SaveLayout {   // activations the backward pass reads again
  float *stats1, *y1, *qkv, *lse, *att, *x2, *stats2, *y2, *pre, *h;
  float *u0, *t0, *t1, *pooled, *gate, *hidden, *x3, *bcs;   // CAB (HAB only)
  int64_t total;
};

int64_t tokens(const neosr_tblock_desc& d) { return (int64_t)d.B * d.H * d.W; }

SaveLayout save_layout(const neosr_tblock_desc& d, float* base) {
  Carve c(base);
  SaveLayout s;
  memset(&s, 0, sizeof(s));
  const int64_t M = tokens(d);
  s.stats1 = c.take(2 * M);
  s.y1 = c.take(M * d.C);
  s.qkv = c.take(M * 3 * d.C);
  s.lse = c.take(M * d.heads);
  s.att = c.take(M * d.C);
  s.x2 = c.take(M * d.C);
  if (d.cab_mid > 0) {
    s.u0 = c.take(M * d.cab_mid);
    s.t0 = c.take(M * d.cab_mid);
    s.t1 = c.take(M * d.C);
    s.pooled = c.take((int64_t)d.B * d.C);
    s.gate = c.take((int64_t)d.B * d.C);
    s.hidden = c.take((int64_t)d.B * d.cab_sq);
    s.x3 = c.take(M * d.C);
    s.bcs = c.take((int64_t)d.B * 128 * d.C);   // scratch of the forward pooling pass
  }
  s.stats2 = c.take(2 * M);
  s.y2 = c.take(M * d.C);
  s.pre = c.take(M * d.hidden);
  s.h = c.take(M * d.hidden);
  s.total = c.off;
  return s;
}

int check(const neosr_tblock_desc* d) {
  NEOSR_CHECK(d, "tblock: null descriptor");
  NEOSR_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->heads > 0 && d->ws > 0 && d->hidden > 0 &&
                  d->C % d->heads == 0 && d->C % 4 == 0 && d->hidden % 4 == 0,
              "tblock: bad geometry");
  NEOSR_CHECK(d->attn == 0 || d->attn == 1, "tblock: attn must be 0 (window attention) or 1 (flash window attention)");
  NEOSR_CHECK(d->n1_w && d->n1_b && d->rpb && d->qkv_w && d->proj_w && d->n2_w && d->n2_b && d->fc1_w && d->fc1_b &&
                  d->fc2_w && d->fc2_b,
              "tblock: missing parameter");
  NEOSR_CHECK(d->cab_mid == 0 || (d->cab_mid > 0 && d->cab_sq > 0 && d->c0_w && d->c0_b && d->c2_w && d->c2_b &&
                                  d->ca1_w && d->ca1_b && d->ca2_w && d->ca2_b),
              "tblock: missing CAB parameter");
  return 0;
}

neosr_gemm_desc gemm_desc(int mode, const float* A, const float* B, float* C, int M, int N, int K) {
  neosr_gemm_desc g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K;
  g.lda = mode != NEOSR_GEMM_TN ? K : M;
  g.ldb = mode == NEOSR_GEMM_NT ? K : N;
  g.ldc = N; g.ldres = N; g.ldaux = N;
  g.mode = mode;
  return g;
}

// plain 3x3 convolution launch as hip/ops.py:conv3x3 describes it (forward / backward-data of a CAB convolution)
// (gelu_out2: forward with GELU in the epilogue, the pre-activation written to gelu_out2; gelu_mask: backward-data multiplied by
// GELU'(gelu_mask) in the epilogue — both only on the F(4x4,3x3) kernel, i.e. when the launch carries a wino4 image)
int conv_launch(const neosr_tblock_desc& d, int mode, const float* in, int in_cs, const float* w, int w_cout, int w_cin,
                const float* bias, float* out, const float* res1, const float* pack, const float* wino, const float* wino4,
                void* stream, float* gelu_out2 = nullptr, const float* gelu_mask = nullptr) {
  neosr_conv_desc c;
  memset(&c, 0, sizeof(c));
  c.in = in; c.in_cs = in_cs; c.w = w; c.bias = bias; c.out = out;
  c.B = d.B; c.H = d.H; c.W = d.W;
  c.K = mode == NEOSR_CONV_FWD ? w_cin : w_cout;
  c.N = mode == NEOSR_CONV_FWD ? w_cout : w_cin;
  c.out_cs = c.N;
  c.w_cout = w_cout; c.w_cin = w_cin;
  c.mode = mode;
  c.mask_slope = 1.f; c.alpha = 1.f; c.alpha2 = 1.f; c.out_mask_slope = 1.f;
  if (res1) { c.res1 = res1; c.res1_cs = c.N; c.res1_nch = c.N; }
  c.w_pack = pack; c.w_wino = wino; c.w_wino4 = wino4;
  if (gelu_out2) { c.act = NEOSR_ACT_GELU; c.out2 = gelu_out2; c.out2_cs = c.N; }
  if (gelu_mask) { c.out_mask = gelu_mask; c.out_mask_cs = c.N; c.out_mask_gelu = 1; }
  return neosr_conv3x3(&c, stream);
}

int wgrad_launch(const neosr_tblock_desc& d, const float* in, int K, const float* g, int N, float* dw, float* db, float* ws,
                 void* stream) {
  neosr_wgrad_desc w;
  memset(&w, 0, sizeof(w));
  w.in = in; w.in_cs = K; w.g = g; w.g_cs = N; w.dw = dw; w.db = db; w.workspace = ws;
  w.B = d.B; w.H = d.H; w.W = d.W; w.K = K; w.N = N;
  w.mask_slope = 1.f; w.scale = 1.f;
  return neosr_conv3x3_wgrad(&w, stream);
}

#define TB_RUN(expr)            \
  do {                          \
    if (int rc__ = (expr)) return rc__; \
  } while (0)

// attention forward / backward with the block's geometry
int attn_fwd(const neosr_tblock_desc& d, const SaveLayout& s, void* stream) {
  if (d.attn == 0) {
    neosr_wattn_desc a;
    memset(&a, 0, sizeof(a));
    a.qkv = s.qkv; a.rpb_table = d.rpb; a.out = s.att; a.lse = s.lse;
    a.B = d.B; a.H = d.H; a.W = d.W; a.C = d.C; a.heads = d.heads; a.ws = d.ws; a.shift = d.shift; a.scale = d.scale;
    return neosr_window_attention_fwd(&a, stream);
  }
  neosr_fattn_desc a;
  memset(&a, 0, sizeof(a));
  a.qkv = s.qkv; a.rpb_table = d.rpb; a.out = s.att; a.lse = s.lse;
  a.B = d.B; a.H = d.H; a.W = d.W; a.C = d.C; a.heads = d.heads; a.ws = d.ws; a.ks = d.ks; a.shift = d.shift; a.scale = d.scale;
  return neosr_flash_window_attention_fwd(&a, stream);
}

neosr_fattn_desc fattn_bwd_desc(const neosr_tblock_desc& d) {
  neosr_fattn_desc a;
  memset(&a, 0, sizeof(a));
  a.B = d.B; a.H = d.H; a.W = d.W; a.C = d.C; a.heads = d.heads; a.ws = d.ws; a.ks = d.ks; a.shift = d.shift; a.scale = d.scale;
  return a;
}

// temporaries of the backward pass
struct BwdLayout {
  float *gpre, *gy2, *dx2, *gatt, *dqkv, *gy1;
  float *tn_fc2, *tn_fc1, *tn_proj, *tn_qkv, *ln2, *ln1, *attn_ws, *many;
  float *gx3, *gt1, *dattn, *dpooled, *gt0, *gu0, *gy1c, *bcs, *wg0, *wg2;   // CAB
  int64_t tn_fc2_n, tn_fc1_n, tn_proj_n, tn_qkv_n, attn_n, many_n;
  int64_t total;
};

int64_t tn_ws_floats(int M, int N, int K) {
  neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_TN, nullptr, nullptr, nullptr, M, N, K);
  return neosr_gemm_workspace_bytes(&g) / 4;
}

constexpr int MAX_JOBS = 12;

BwdLayout bwd_layout(const neosr_tblock_desc& d, float* base) {
  Carve c(base);
  BwdLayout b;
  memset(&b, 0, sizeof(b));
  const int64_t M = tokens(d);
  const int C = d.C, Hd = d.hidden;
  b.gpre = c.take(M * Hd);
  b.gy2 = c.take(M * C);
  b.dx2 = c.take(M * C);
  b.gatt = c.take(M * C);
  b.dqkv = c.take(M * 3 * C);
  b.gy1 = c.take(M * C);
  b.tn_fc2_n = tn_ws_floats(C, Hd, (int)M);
  b.tn_fc1_n = tn_ws_floats(Hd, C, (int)M);
  b.tn_proj_n = tn_ws_floats(C, C, (int)M);
  b.tn_qkv_n = tn_ws_floats(3 * C, C, (int)M);
  b.tn_fc2 = c.take(b.tn_fc2_n);
  b.tn_fc1 = c.take(b.tn_fc1_n);
  b.tn_proj = c.take(b.tn_proj_n);
  b.tn_qkv = c.take(b.tn_qkv_n);
  b.ln2 = c.take((int64_t)(2 * 1024 + 512) * C);
  b.ln1 = c.take((int64_t)(2 * 1024 + 512) * C);
  const int nW = (d.H / d.ws) * (d.W / d.ws);
  if (d.attn == 0) {
    b.attn_n = ((int64_t)d.B * nW + 256) * d.heads * (2 * d.ws - 1) * (2 * d.ws - 1);
  } else {
    neosr_fattn_desc a = fattn_bwd_desc(d);
    a.qkv = d.qkv_w; a.rpb_table = d.rpb;   // (the size query validates the descriptor: any non-null pointers)
    b.attn_n = neosr_flash_window_attention_workspace_bytes(&a) / 4;
  }
  b.attn_ws = c.take(b.attn_n);
  // neosr_colsum_many: its workspace depends on the jobs' shapes only; upper bound over this block's jobs
  {
    neosr_colsum_item it[MAX_JOBS];
    memset(it, 0, sizeof(it));
    int n = 0;
    auto job = [&](int rows, int64_t cols) { it[n].rows = rows; it[n].cols = (int)cols; it[n].ld = (int)cols; ++n; };
    // (row counts are bounded by the split counts the kernels may return: 256 splits / 1024 LN partial rows / windows)
    job(256, (int64_t)C * Hd + C);
    job(256, (int64_t)Hd * C + Hd);
    job(256, (int64_t)C * C + C);
    job(256, (int64_t)3 * C * C + 3 * C);
    job(1024, 2 * C);
    job(1024, 2 * C);
    const int kk = d.attn ? d.ks : d.ws;
    job(d.B * nW * (d.attn ? (d.ws * d.ws / 64 > 0 ? d.ws * d.ws / 64 : 1) : 1), (int64_t)d.heads * (d.ws + kk - 1) * (d.ws + kk - 1));
    b.many_n = 2 * neosr_colsum_many_workspace_floats(it, n) + 4096;
  }
  b.many = c.take(b.many_n);
  if (d.cab_mid > 0) {
    const int mid = d.cab_mid;
    b.gt1 = c.take(M * C);
    b.dattn = c.take((int64_t)d.B * C);
    b.dpooled = c.take((int64_t)d.B * C);
    b.gt0 = c.take(M * mid);
    b.gu0 = c.take(M * mid);
    b.gy1c = c.take(M * C);
    b.bcs = c.take((int64_t)d.B * 128 * C);
    b.wg2 = c.take(neosr_conv3x3_wgrad_workspace_bytes(d.B, d.H, d.W, mid, C) / 4 + 64);
    b.wg0 = c.take(neosr_conv3x3_wgrad_workspace_bytes(d.B, d.H, d.W, C, mid) / 4 + 64);
  }
  b.total = c.off;
  return b;
}

// Side stream of the backward plan.  The four weight-gradient GEMMs (and the two weight gradients of the CAB convolutions)
// of a block feed nothing inside the block — only the optimizer reads them — while the data-gradient chain
// (GEMM -> LayerNorm -> GEMM -> attention -> GEMM -> LayerNorm) is a sequence of DEPENDENT launches that each pay their
// fill / first-load / drain in the open (~15 us of a 30 us launch at M = 16 384: DESIGN §7).  They run on a library-owned
// stream, forked / joined with events inside the call (each waits for the event behind the kernel that produced its
// operand; the caller's stream waits for the side stream before the block's batched column sums), so their workgroups
// fill the CUs the chain leaves idle at every launch boundary.  Same kernels, same operands: bit-identical results.
// Mode 2 (NEOSR_AMD_BLOCK_STREAMS=2 / neosr_set_tblock_streams(2)), not the default: measured on MI355X in round 4,
// swinir_medium (B = 8) 38.52 vs 38.54 ms per step (nothing), hat_l (B = 4) 97.9 -> 101.9 ms — these GEMMs fill the chip
// by themselves (928 workgroups), there is nothing for them to fill, and the 14 event calls per block cost host time.
struct Side {
  int dev = -1;
  hipStream_t s = nullptr;
  hipStream_t tail = nullptr;   // stream of the backward plans' weight-gradient TAILS (below)
  hipEvent_t ev[16] = {};   // (EV_TAIL_DONE .. + 7: one per tail in flight, round robin)
};
enum { EV_FORK = 0, EV_GPRE, EV_DX2, EV_GT1, EV_GU0, EV_DQKV, EV_JOIN, EV_TAIL_FORK, EV_TAIL_DONE /* .. EV_TAIL_DONE + 7 */ };
constexpr int TAIL_RING = 8;
// The TAIL of a block's backward plan — the grouped weight-gradient GEMMs and the batched column sums that finish the
// parameter gradients (~75 us per block at B = 8 / 64 x 64) — feeds nothing in the backward pass: only the optimizer (and
// the gradient exchange) read what it writes, while the NEXT block's data-gradient chain waits behind it on the caller's
// stream.  With NEOSR_AMD_BLOCK_TAIL (default on, modes 3 only) it runs on a second library stream and the call returns with
// it in flight (round 6: swinir_medium +4.0 %, hat_l +3.3 %, same-box).  What that costs the caller:
//   * everything the tail reads or writes — the workspace, the block's save buffer, the incoming gradient — must stay
//     untouched until the tail is done.  Nothing is put on the caller's stream for that (a wait per block on the critical
//     chain cost two thirds of the gain): tail n records event n mod 8 on the tail stream and the caller asks
//     neosr_tblock_tail_done(n) — a host-side event query — before it lets go of the buffers of call n (hip/transformer.py
//     keeps them in a short ring and joins if the ring grows past six entries);
//   * the parameter gradients are complete only behind neosr_tblock_tail_join(stream), which makes `stream` wait for every
//     tail issued so far: the Python side calls it at the end of the backward pass (an autograd engine callback), before
//     a data-parallel bucket leaves (utils/grad_sync.py), and at once when a gradient is going to be ACCUMULATED into an
//     existing one.
std::atomic<int> g_block_tail{-1};      // -1: read NEOSR_AMD_BLOCK_TAIL on first use (default 1)
std::atomic<long> g_tails{0};           // tails issued so far
std::atomic<long> g_tails_joined{0};    // ... covered by the last neosr_tblock_tail_join
std::atomic<int> g_block_streams{-1};   // (read by the forward thread and the autograd thread, written by the setter)
std::atomic<long> g_side_forks{0};      // forks onto the side stream so far (tests ask: was the path taken?)
// mode 3 (NEOSR_AMD_BLOCK_STREAMS=3): the CAB branch of a HAB — a chain of SMALL launches (B = 4: 128-192 twelve-wave
// workgroups per convolution on 256 CUs, a one-workgroup channel-attention kernel, ~65 us forward / ~150 us backward per
// block) that only meets the attention branch at the sum in front of norm2 (forward) and at norm1's input gradient
// (backward) — runs on the side stream beside the attention branch: fork behind norm1 (forward) / behind norm2's
// backward, join in front of the sum / the qkv data-gradient GEMM.  Two event pairs per block and direction.  THE DEFAULT:
// hat_l (configs[4], B = 4) 95.2 -> 89.1 ms per step, same-box A/B; rocprofv3: 4 896 of 19 336 launches on the second
// queue, two kernels in flight for 93 of 365 ms of kernel time (tools/trace_overlap.sh).  Same kernels, same operands,
// same order inside each chain: bit-identical (tests/test_hip_blocks.py).  A stream under hipGraph capture keeps the
// whole block on itself.
// Joins the caller's stream with the side stream when the call leaves — on the normal path through join(), on an error
// return through the destructor: the caller frees `save` / the workspace right after an error, and the caching allocator
// may hand them out again while side-stream kernels still use them.  Errors inside the guard are swallowed:
// the call is already returning one.
struct SideJoin {
  Side* side = nullptr;
  hipStream_t caller = nullptr;
  bool armed = false;
  void arm(Side* s, void* st) { side = s; caller = (hipStream_t)st; armed = true; }
  int join() {   // record behind everything the side stream holds, make the caller's stream wait for it
    if (!armed) return 0;
    armed = false;
    NEOSR_HIP(hipEventRecord(side->ev[EV_JOIN], side->s));
    NEOSR_HIP(hipStreamWaitEvent(caller, side->ev[EV_JOIN], 0));
    return 0;
  }
  ~SideJoin() {
    if (armed && hipEventRecord(side->ev[EV_JOIN], side->s) == hipSuccess)
      (void)hipStreamWaitEvent(caller, side->ev[EV_JOIN], 0);
  }
};

// CAB: GELU / GELU' inside the epilogues of its two convolutions (only the F(4x4,3x3) kernel has them: launches that carry
// its image); NEOSR_AMD_CAB_GELU_FUSED=0 keeps the elementwise passes.  Same expressions (gelu.h): bit-identical.
bool cab_gelu_fused(const float* wino4_image) {
  static const bool on = [] { const char* e = getenv("NEOSR_AMD_CAB_GELU_FUSED"); return !(e && e[0] == '0'); }();
  return on && wino4_image != nullptr && neosr_get_winograd() == 2;
}

bool not_capturing(void* stream) {
  if (stream) {   // (the null stream cannot be captured; a stream under hipGraph capture keeps the block on itself)
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    if (st != hipStreamCaptureStatusNone) return false;
  }
  return true;
}
bool cab_on_side(const Side* side, const neosr_tblock_desc& d, void* stream) {
  if (!side || g_block_streams != 3 || d.cab_mid <= 0) return false;
  return not_capturing(stream);
}
bool tail_on_side(const Side* side, void* stream) {
  if (g_block_tail < 0) {
    const char* e = getenv("NEOSR_AMD_BLOCK_TAIL");
    g_block_tail = (e && e[0] == '0') ? 0 : 1;
  }
  return side && side->tail && g_block_streams == 3 && g_block_tail == 1 && not_capturing(stream);
}

Side* side_get() {
  static Side a;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_block_streams < 0) {
    const char* e = getenv("NEOSR_AMD_BLOCK_STREAMS");
    const int v = e ? atoi(e) : 3;
    g_block_streams = (v == 2 || v == 3) ? v : 1;
  }
  if (g_block_streams < 2) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (a.s && a.dev != dev) a = Side();   // streams / events belong to the device they were created on
  a.dev = dev;
  if (!a.s) {
    if (hipStreamCreateWithFlags(&a.s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipStreamCreateWithFlags(&a.tail, hipStreamNonBlocking) != hipSuccess) return nullptr;
    for (auto& e : a.ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  return &a;
}

}  // namespace

extern "C" int64_t neosr_tblock_side_forks(void) { return g_side_forks; }

extern "C" int neosr_set_tblock_streams(int n) {
  const int cur = g_block_streams.load();
  const int prev = cur < 0 ? 3 : cur;
  g_block_streams = (n == 2 || n == 3) ? n : 1;
  return prev;
}

extern "C" int64_t neosr_tblock_save_floats(const neosr_tblock_desc* d) {
  if (check(d)) return -1;
  return save_layout(*d, nullptr).total;
}

extern "C" int64_t neosr_tblock_bwd_workspace_floats(const neosr_tblock_desc* d) {
  if (check(d)) return -1;
  return bwd_layout(*d, nullptr).total;
}

extern "C" int neosr_tblock_forward(const neosr_tblock_desc* dp, const float* x, float* out, float* save, void* stream) {
  if (check(dp)) return 1;
  const neosr_tblock_desc& d = *dp;
  NEOSR_CHECK(x && out && save, "tblock forward: null buffer");
  const SaveLayout s = save_layout(d, save);
  const int64_t M64 = tokens(d);
  NEOSR_CHECK(M64 * 3 * d.C < (int64_t(1) << 31), "tblock: token matrix too large for 32-bit GEMM extents");
  const int M = (int)M64, C = d.C, Hd = d.hidden, rps = d.H * d.W;
  // norm1 (the shortcut is x itself)
  TB_RUN(neosr_layernorm_fwd(x, d.n1_w, d.n1_b, s.y1, s.stats1, M, C, d.eps1, stream));
  // HAB: x3 = x2 + conv_scale * CAB(norm1(x))   (hat_arch.py:15-52, 347).  The CAB chain is enqueued first: on the side
  // stream (mode 3) it then runs beside the attention branch; on the caller's stream the order of two independent chains
  // does not matter
  Side* side = side_get();
  const bool cab_side = cab_on_side(side, d, stream);
  SideJoin sj;
  if (d.cab_mid > 0) {
    const int mid = d.cab_mid;
    if (cab_side) {
      NEOSR_HIP(hipEventRecord(side->ev[EV_FORK], (hipStream_t)stream));
      NEOSR_HIP(hipStreamWaitEvent(side->s, side->ev[EV_FORK], 0));
      sj.arm(side, stream);
      ++g_side_forks;
    }
    void* cs = cab_side ? (void*)side->s : stream;
    if (cab_gelu_fused(d.c0_wino4_f)) {   // GELU in the convolution's epilogue, the pre-activation as its second output
      TB_RUN(conv_launch(d, NEOSR_CONV_FWD, s.y1, C, d.c0_w, mid, C, d.c0_b, s.t0, nullptr, d.c0_pack_f, d.c0_wino_f,
                         d.c0_wino4_f, cs, s.u0));
    } else {
      TB_RUN(conv_launch(d, NEOSR_CONV_FWD, s.y1, C, d.c0_w, mid, C, d.c0_b, s.u0, nullptr, d.c0_pack_f, d.c0_wino_f,
                         d.c0_wino4_f, cs));
      TB_RUN(neosr_gelu(s.u0, nullptr, s.t0, (int64_t)M * mid, cs));
    }
    TB_RUN(conv_launch(d, NEOSR_CONV_FWD, s.t0, mid, d.c2_w, C, mid, d.c2_b, s.t1, nullptr, d.c2_pack_f, d.c2_wino_f,
                       d.c2_wino4_f, cs));
    TB_RUN(neosr_batched_colsum(s.t1, nullptr, s.pooled, s.bcs, d.B, rps, C, 1.0f / rps, cs));
    TB_RUN(neosr_channel_attention_fwd(s.pooled, d.ca1_w, d.ca1_b, d.ca2_w, d.ca2_b, s.hidden, s.gate, d.B, C, d.cab_sq,
                                       cs));
  }
  // qkv
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NT, s.y1, d.qkv_w, s.qkv, M, 3 * C, C);
    g.bias = d.qkv_b;
    TB_RUN(neosr_gemm(&g, stream));
  }
  TB_RUN(attn_fwd(d, s, stream));
  // proj + DropPath + shortcut
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NT, s.att, d.proj_w, s.x2, M, C, C);
    g.bias = d.proj_b; g.res = x; g.row_scale = d.drop_scale; g.rows_per_scale = d.drop_scale ? rps : 0;
    TB_RUN(neosr_gemm(&g, stream));
  }
  const float* xm = s.x2;
  if (d.cab_mid > 0) {
    TB_RUN(sj.join());
    TB_RUN(neosr_scale_channels_add(s.t1, s.gate, s.x2, s.x3, d.B, rps, C, d.conv_scale, stream));
    xm = s.x3;
  }
  TB_RUN(neosr_layernorm_fwd(xm, d.n2_w, d.n2_b, s.y2, s.stats2, M, C, d.eps2, stream));
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NT, s.y2, d.fc1_w, s.h, M, Hd, C);
    g.bias = d.fc1_b; g.aux_out = s.pre; g.gelu = 1;
    TB_RUN(neosr_gemm(&g, stream));
  }
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NT, s.h, d.fc2_w, out, M, C, Hd);
    g.bias = d.fc2_b; g.res = xm; g.row_scale = d.drop_scale2; g.rows_per_scale = d.drop_scale2 ? rps : 0;
    TB_RUN(neosr_gemm(&g, stream));
  }
  return 0;
}

// Backward.  `grads` points at the block's parameter gradients in ONE buffer laid out like the parameters in the
// network's flat arena (named_parameters order, every tensor start rounded up to 4 floats): each (weight, bias) and
// (gamma, beta) pair is then contiguous, which lets ONE fixed-order column-sum job finish both (neosr_gemm TN with
// accumulate == 2, neosr_layernorm_bwd_res without targets).  All column sums of the block — four split-K weight
// gradients, two LayerNorm affine gradients, the relative-position-bias bins — run as one neosr_colsum_many at the end.
extern "C" int neosr_tblock_backward(const neosr_tblock_desc* dp, const float* x, const float* dout, const float* save,
                                     float* dx, const neosr_tblock_grads* gp, float* workspace, void* stream) {
  if (check(dp)) return 1;
  const neosr_tblock_desc& d = *dp;
  NEOSR_CHECK(x && dout && save && dx && gp && workspace, "tblock backward: null buffer");
  const neosr_tblock_grads& G = *gp;
  NEOSR_CHECK(G.n1_w && G.rpb && G.qkv_w && G.proj_w && G.n2_w && G.fc1_w && G.fc2_w, "tblock backward: missing gradient target");
  const SaveLayout s = save_layout(d, const_cast<float*>(save));
  const BwdLayout b = bwd_layout(d, workspace);
  const int M = (int)tokens(d), C = d.C, Hd = d.hidden, rps = d.H * d.W;
  const float* rs = d.drop_scale;      // attention branch
  const int rsn = rs ? rps : 0;
  const float* rs2 = d.drop_scale2;    // MLP branch
  const int rsn2 = rs2 ? rps : 0;
  NEOSR_CHECK(G.n1_b == G.n1_w + C && G.n2_b == G.n2_w + C && G.fc1_b == G.fc1_w + (int64_t)Hd * C &&
                  G.fc2_b == G.fc2_w + (int64_t)C * Hd && G.proj_b == G.proj_w + (int64_t)C * C &&
                  (!d.qkv_b || G.qkv_b == G.qkv_w + (int64_t)3 * C * C),
              "tblock backward: (weight, bias) / (gamma, beta) gradient pairs must be contiguous");
  Side* side_any = side_get();
  const bool cab_side = cab_on_side(side_any, d, stream);          // mode 3: the CAB branch on the side stream
  const bool tail_side = tail_on_side(side_any, stream);           // mode 3: the weight-gradient tail on the tail stream
  Side* side = (side_any && g_block_streams == 2) ? side_any : nullptr;   // mode 2: the weight gradients on the side stream
  void* sw = side ? (void*)side->s : stream;   // the stream of the weight gradients
  SideJoin sj;                                 // armed by the first fork of either mode
  // `after(e)`: the side stream continues behind what the caller's stream has enqueued so far
  auto after = [&](int e) -> int {
    if (!side) return 0;
    NEOSR_HIP(hipEventRecord(side->ev[e], (hipStream_t)stream));
    NEOSR_HIP(hipStreamWaitEvent(side->s, side->ev[e], 0));
    if (!sj.armed) sj.arm(side, stream);
    ++g_side_forks;
    return 0;
  };
  TB_RUN(after(EV_FORK));
  neosr_colsum_item jobs[MAX_JOBS];
  int nj = 0;
  auto add_job = [&](const float* part, int rows, int64_t cols, float* out) {
    jobs[nj].x = part; jobs[nj].out = out; jobs[nj].rows = rows; jobs[nj].cols = (int)cols; jobs[nj].ld = (int)cols;
    jobs[nj].accumulate = 0;
    ++nj;
  };
  // TN GEMM with its split reduction left for the batched pass.  Without the side stream the four weight-gradient GEMMs
  // of the block are only COLLECTED here and go out as ONE launch behind the data-gradient chain (neosr_gemm_tn_group:
  // one ramp / drain instead of four; NEOSR_AMD_TN_GROUP=0 launches them where they are described)
  static const bool group_on = [] { const char* e = getenv("NEOSR_AMD_TN_GROUP"); return !(e && e[0] == '0'); }();
  neosr_gemm_desc tn[4];
  int ntn = 0;
  auto wgrad = [&](const float* dy, const float* xin, float* dw, float* db, int N, int K, float* ws, const float* rscale,
                   int rscale_n) -> int {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_TN, dy, xin, dw, N, K, M);
    g.colsum_a = db; g.workspace = ws; g.accumulate = 2; g.row_scale = rscale; g.rows_per_scale = rscale_n;
    if (group_on && !side && ntn < 4) {
      tn[ntn++] = g;
      return 0;
    }
    const int rc = neosr_gemm(&g, sw);
    if (rc >= 0) return rc ? rc : (neosr_set_error("tblock: TN gemm did not defer its reduction"), 1);
    add_job(ws, -rc, (int64_t)N * K + (db ? N : 0), dw);
    return 0;
  };
  const float* xm = d.cab_mid > 0 ? s.x3 : s.x2;   // input of norm2
  NEOSR_CHECK(d.cab_mid == 0 || (G.c0_w && G.c0_b && G.c2_w && G.c2_b && G.ca1_w && G.ca1_b && G.ca2_w && G.ca2_b),
              "tblock backward: missing CAB gradient target");
  // ---- Mlp (hip/transformer.py: Mlp.backward)
  TB_RUN(wgrad(dout, s.h, G.fc2_w, G.fc2_b, C, Hd, b.tn_fc2, rs2, rsn2));
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NN, dout, d.fc2_w, b.gpre, M, Hd, C);   // (g W2) GELU'(pre)
    g.aux_in = s.pre; g.row_scale = rs2; g.rows_per_scale = rsn2;
    TB_RUN(neosr_gemm(&g, stream));
  }
  TB_RUN(after(EV_GPRE));
  TB_RUN(wgrad(b.gpre, s.y2, G.fc1_w, G.fc1_b, Hd, C, b.tn_fc1, nullptr, 0));
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NN, b.gpre, d.fc1_w, b.gy2, M, C, Hd);
    TB_RUN(neosr_gemm(&g, stream));
  }
  // ---- norm2 with the shortcut gradient (= dout) summed in the same pass
  {
    const int rc = neosr_layernorm_bwd_res(b.gy2, xm, s.stats2, d.n2_w, dout, b.dx2, nullptr, nullptr, b.ln2, M, C, 0, stream);
    if (rc >= 0) return rc ? rc : (neosr_set_error("tblock: layernorm did not defer its reduction"), 1);
    add_job(b.ln2, -rc, 2 * C, G.n2_w);
  }
  TB_RUN(after(EV_DX2));
  // ---- CAB branch (HAB): the gradient that reached x3 (b.dx2) also is the gradient of x2 (x3 = x2 + ...)
  if (d.cab_mid > 0) {
    const int mid = d.cab_mid;
    // mode 3: the whole branch (its weight gradients included) on the side stream, behind norm2's backward; the caller's
    // stream goes on with proj / attention and waits for it in front of the qkv data-gradient GEMM (which adds gy1c)
    void* cs = cab_side ? (void*)side_any->s : stream;
    void* cw = cab_side ? cs : sw;
    if (cab_side) {
      NEOSR_HIP(hipEventRecord(side_any->ev[EV_FORK], (hipStream_t)stream));
      NEOSR_HIP(hipStreamWaitEvent(side_any->s, side_any->ev[EV_FORK], 0));
      sj.arm(side_any, stream);
      ++g_side_forks;
    }
    // channel gate (hip/transformer.py: ChannelGate.backward)
    TB_RUN(neosr_batched_colsum(b.dx2, s.t1, b.dattn, b.bcs, d.B, rps, C, d.conv_scale, cs));
    TB_RUN(neosr_channel_attention_bwd(b.dattn, s.gate, s.hidden, s.pooled, d.ca1_w, d.ca2_w, b.dpooled, G.ca1_w, G.ca1_b,
                                       G.ca2_w, G.ca2_b, d.B, C, d.cab_sq, cs));
    TB_RUN(neosr_scale_channels_bwd(b.dx2, s.gate, b.dpooled, b.gt1, d.B, rps, C, d.conv_scale, cs));
    // second convolution: weight + bias gradient (side stream), data gradient (hip/layers.py: Conv3x3.backward)
    TB_RUN(after(EV_GT1));
    TB_RUN(wgrad_launch(d, s.t0, mid, b.gt1, C, G.c2_w, G.c2_b, b.wg2, cw));
    if (cab_gelu_fused(d.c2_wino4_d)) {   // GELU'(u0) in the data gradient's epilogue
      TB_RUN(conv_launch(d, NEOSR_CONV_DGRAD, b.gt1, C, d.c2_w, C, mid, nullptr, b.gu0, nullptr, d.c2_pack_d, d.c2_wino_d,
                         d.c2_wino4_d, cs, nullptr, s.u0));
    } else {
      TB_RUN(conv_launch(d, NEOSR_CONV_DGRAD, b.gt1, C, d.c2_w, C, mid, nullptr, b.gt0, nullptr, d.c2_pack_d, d.c2_wino_d,
                         d.c2_wino4_d, cs));
      TB_RUN(neosr_gelu(s.u0, b.gt0, b.gu0, (int64_t)M * mid, cs));
    }
    // first convolution; its data gradient is one of the two contributions to norm1's output (the other comes from qkv,
    // whose GEMM adds this one in its epilogue below: a + b in either order, as autograd's accumulation would)
    TB_RUN(after(EV_GU0));
    TB_RUN(wgrad_launch(d, s.y1, C, b.gu0, mid, G.c0_w, G.c0_b, b.wg0, cw));
    TB_RUN(conv_launch(d, NEOSR_CONV_DGRAD, b.gu0, mid, d.c0_w, mid, C, nullptr, b.gy1c, nullptr, d.c0_pack_d, d.c0_wino_d,
                       d.c0_wino4_d, cs));
  }
  // ---- proj (Linear.backward): data gradient with the DropPath scale in the epilogue, weight gradient with it on the rows
  TB_RUN(wgrad(b.dx2, s.att, G.proj_w, G.proj_b, C, C, b.tn_proj, rs, rsn));
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NN, b.dx2, d.proj_w, b.gatt, M, C, C);
    g.row_scale = rs; g.rows_per_scale = rsn;
    TB_RUN(neosr_gemm(&g, stream));
  }
  // ---- attention
  {
    const int bins = (d.attn == 0 || d.ks == d.ws) ? (2 * d.ws - 1) * (2 * d.ws - 1) : 0;
    int rc;
    const float* part = b.attn_ws;
    if (d.attn == 0) {
      neosr_wattn_desc a;
      memset(&a, 0, sizeof(a));
      a.qkv = s.qkv; a.rpb_table = d.rpb; a.out = s.att; a.lse = s.lse; a.dout = b.gatt; a.dqkv = b.dqkv;
      a.d_rpb_table = G.rpb; a.workspace = b.attn_ws;
      a.B = d.B; a.H = d.H; a.W = d.W; a.C = C; a.heads = d.heads; a.ws = d.ws; a.shift = d.shift; a.scale = d.scale;
      a.accumulate_rpb = d.ks == d.ws ? 2 : 0;   // (the overlapping form gathers its bins from a dense sum itself)
      rc = neosr_window_attention_bwd(&a, stream);
    } else {
      neosr_fattn_desc a = fattn_bwd_desc(d);
      a.qkv = s.qkv; a.rpb_table = d.rpb; a.out = s.att; a.lse = s.lse; a.dout = b.gatt; a.dqkv = b.dqkv;
      a.d_rpb_table = G.rpb; a.workspace = b.attn_ws;
      a.accumulate_rpb = d.ks == d.ws ? 2 : 0;   // (the overlapping form gathers its bins from a dense sum itself)
      rc = neosr_flash_window_attention_bwd(&a, stream);
      part = b.attn_ws + (int64_t)d.B * (d.H / d.ws) * (d.W / d.ws) * d.heads * d.ws * d.ws;
    }
    if (rc < 0) add_job(part, -rc, (int64_t)d.heads * bins, G.rpb);
    else if (rc) return rc;
  }
  // ---- qkv
  // (mode 3: the side stream holds only the CAB branch — recording the join event here, behind the attention launches of
  // the caller's stream, orders the same work as recording it right behind the branch)
  if (cab_side) TB_RUN(sj.join());
  TB_RUN(after(EV_DQKV));
  TB_RUN(wgrad(b.dqkv, s.y1, G.qkv_w, d.qkv_b ? G.qkv_b : nullptr, 3 * C, C, b.tn_qkv, nullptr, 0));
  {
    neosr_gemm_desc g = gemm_desc(NEOSR_GEMM_NN, b.dqkv, d.qkv_w, b.gy1, M, C, 3 * C);
    if (d.cab_mid > 0) g.res = b.gy1c;
    TB_RUN(neosr_gemm(&g, stream));
  }
  // ---- norm1 with the shortcut gradient (= what reached x2 = dx2)
  {
    const int rc = neosr_layernorm_bwd_res(b.gy1, x, s.stats1, d.n1_w, b.dx2, dx, nullptr, nullptr, b.ln1, M, C, 0, stream);
    if (rc >= 0) return rc ? rc : (neosr_set_error("tblock: layernorm did not defer its reduction"), 1);
    add_job(b.ln1, -rc, 2 * C, G.n1_w);
  }
  // ---- the tail (see the comment at g_block_tail): behind everything the caller's stream holds so far
  void* ts = stream;
  if (tail_side) {
    NEOSR_HIP(hipEventRecord(side_any->ev[EV_TAIL_FORK], (hipStream_t)stream));
    NEOSR_HIP(hipStreamWaitEvent(side_any->tail, side_any->ev[EV_TAIL_FORK], 0));
    ts = (void*)side_any->tail;
  }
  if (ntn) {   // the collected weight-gradient GEMMs: one launch (or, if a shape does not qualify, one each)
    int32_t ns[4];
    int rc = neosr_gemm_tn_group(tn, ntn, ns, ts);
    if (rc > 0) return rc;
    for (int i = 0; i < ntn; ++i) {
      if (rc < 0) {
        const int r1 = neosr_gemm(&tn[i], ts);
        if (r1 >= 0) return r1 ? r1 : (neosr_set_error("tblock: TN gemm did not defer its reduction"), 1);
        ns[i] = -r1;
      }
      add_job(tn[i].workspace, ns[i], (int64_t)tn[i].M * tn[i].N + (tn[i].colsum_a ? tn[i].M : 0), tn[i].C);
    }
  }
  if (side) TB_RUN(sj.join());   // the column sums below read the partials of both streams
  NEOSR_CHECK(neosr_colsum_many_workspace_floats(jobs, nj) <= b.many_n, "tblock backward: column-sum workspace too small");
  TB_RUN(neosr_colsum_many(jobs, nj, b.many, ts));
  if (tail_side) {
    const long n = g_tails.fetch_add(1);
    NEOSR_HIP(hipEventRecord(side_any->ev[EV_TAIL_DONE + (n % TAIL_RING)], side_any->tail));
  }
  return 0;
}

// `stream` waits for every tail issued so far; returns how many were outstanding
extern "C" int64_t neosr_tblock_tail_join(void* stream) {
  const long n = g_tails.load();
  const long pending = n - g_tails_joined.load();
  if (pending <= 0) return 0;
  Side* side = side_get();
  if (!side) return -1;
  // (in-order stream: the newest tail's event covers all of them)
  if (hipStreamWaitEvent((hipStream_t)stream, side->ev[EV_TAIL_DONE + ((n - 1) % TAIL_RING)], 0) != hipSuccess) return -1;
  g_tails_joined = n;
  return pending;
}
extern "C" int64_t neosr_tblock_tails(void) { return g_tails; }
// 1: tail number `index` (0-based, in issue order) has finished; 0: still running; host-side query, no synchronisation.
// Only the newest TAIL_RING - 1 tails can be asked about individually: an older index answers for the tail that took its
// event slot since (callers keep at most six buffers: hip/transformer.py).
extern "C" int neosr_tblock_tail_done(int64_t index) {
  const long n = g_tails.load();
  if (index < 0 || index >= n) return 1;
  Side* side = side_get();
  if (!side) return 1;
  const hipError_t e = hipEventQuery(side->ev[EV_TAIL_DONE + (index % TAIL_RING)]);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  return 1;
}
extern "C" int neosr_set_tblock_tail(int on) {
  const int prev = g_block_tail < 0 ? 1 : g_block_tail.load();
  g_block_tail = on ? 1 : 0;
  return prev;
}

