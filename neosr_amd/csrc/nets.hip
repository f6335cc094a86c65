// nets.hip — host-side execution plans: whole-network forward/backward for the generators on the
// hot path, expressed as a fixed sequence of kernel launches on one HIP stream.  No per-layer
// Python, no torch.cat (the RDB concat lives in one 192-channel channels-last buffer), no
// allocation (the caller hands in one workspace), no host synchronisation -> graph-capturable.
//
// Reference behaviour restated here: neosr/archs/esrgan_arch.py:82-214 (ResidualDenseBlock, RRDB,
// esrgan.forward) and neosr/archs/compact_arch.py:11-85 (compact.forward), plus their autograd
// backward.
#include "common.h"
#include "conv_pack.h"
#include "conv_wino4_chain.h"
namespace neosr_conv {
int wino_mode();                  // conv_wino.hip: 0 direct, 1 F(2x2,3x3), 2 F(4x4,3x3)
extern int g_wino4_concurrency;   // conv_wino4.hip: launch chains running side by side (fill estimate of the F(4x4) kernel)
}
#include "../../include/neosr_amd.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>

extern "C" int neosr_fill(float* p, int64_t n, float v, void* stream);
extern "C" int neosr_pool2x2_sum_masked(const float* in, float* out, const float* mask, int32_t B, int32_t H,
                                        int32_t W, int32_t C, int32_t in_cs, int32_t out_cs, int32_t mask_cs,
                                        float mask_slope, void* stream);
extern "C" int neosr_axpy_slice(float* out, const float* in, int64_t npix, int32_t C,
                                int32_t out_cs, int32_t in_cs, float alpha, void* stream);

namespace {

struct Bump {
  char* base;
  int64_t off = 0;
  explicit Bump(void* b) : base((char*)b) {}
  float* take(int64_t nfloats) {
    off = (off + 255) & ~(int64_t)255;
    float* p = base ? (float*)(base + off) : nullptr;
    off += nfloats * 4;
    return p;
  }
};

inline int pad4(int c) { return (c + 3) & ~3; }

neosr_conv_desc conv_base(int B, int H, int W) {
  neosr_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B;
  d.H = H;
  d.W = W;
  d.alpha = 1.f;
  d.alpha2 = 1.f;
  d.mask_slope = 1.f;
  return d;
}

neosr_wgrad_desc wgrad_base(int B, int H, int W) {
  neosr_wgrad_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B;
  d.H = H;
  d.W = W;
  d.scale = 1.f;
  d.mask_slope = 1.f;
  return d;
}

#define RUN(expr)                \
  do {                           \
    if (int rc__ = (expr)) return rc__; \
  } while (0)

// the F(4x4,3x3) kernel's fill estimate counts the workgroups of `n` launch chains together while this is alive
struct ChainHint {
  int prev;
  explicit ChainHint(int n) : prev(neosr_conv::g_wino4_concurrency) { neosr_conv::g_wino4_concurrency = n; }
  ~ChainHint() { neosr_conv::g_wino4_concurrency = prev; }
};

// ------------------------------------------------------------------------------ RRDBNet
struct RrdbLayout {
  int B, H, W, Cin, Cout, F, G, NB, CC, cin_cs, cout_cs, nact;
  int64_t np1, np2, np4;
  float* x_nhwc;
  std::vector<float*> act;
  float *trunk, *fea, *u1, *u2, *hr, *y_nhwc;
  float *gy_nhwc, *g_hr, *g_u2, *g_up2in, *g_u1, *g_up1in, *g_fea, *g_trunk, *gx_nhwc;
  float* gb[4];
  float *wg_ws, *wg_ws2;
  // packed weight images of the RDB convs (conv_pack.h): forward = one image per conv; backward = one
  // image per gradient slice of the concat buffer, rows concatenating every conv that consumes it
  float *wpack_f, *wpack_d;
  int64_t pf_off[5], pd_off[5], pf_total, pd_total;
  // Winograd images of the same launches (conv_wino.hip), same indexing
  float *wwino_f, *wwino_d;
  int64_t wf_off[5], wd_off[5], wf_total, wd_total;
  // direct + Winograd images of conv_body / conv_up1 / conv_up2 / conv_hr (forward) and of their backward-data
  float *tail_pf, *tail_wf, *tail_pd, *tail_wd;
  int64_t tail_p, tail_w;  // floats per image
  // Winograd F(4x4,3x3) images (conv_wino4.hip), same indexing; only the set the current neosr_set_winograd mode uses is packed
  float *wwino4_f, *wwino4_d, *tail_w4f, *tail_w4d;
  int64_t w4f_off[5], w4d_off[5], w4f_total, w4d_total, tail_w4;
  // trunk launches take the F(4x4,3x3) kernel: mode 2 and enough 16 x 16-pixel tiles x 32-cout blocks over the WHOLE
  // batch (the launch chains run side by side) to fill the chip — NEOSR_WINO4_MIN_WGS; else F(2x2,3x3)
  bool w4_trunk;
  // the trunk launches can still reach the direct-to-LDS kernel (Winograd switched off, or buffers of 2 GB and more that
  // the F(4x4) kernel's 32-bit offsets refuse): only then are the direct images worth packing
  bool direct_trunk;
  // conv3x3_wino4_chain_kernel (conv_wino4_chain.hip): one launch per RRDB and direction (the fifteen layer records travel
  // in the kernel arguments); per launch one flag word per 16 x 16-pixel tile
  // A batch with more tiles than CUs runs as `chain_groups` launches over consecutive groups of `chain_gb` samples.
  float* chain_flags;
  int64_t chain_flag_words;
  int chain_groups, chain_gb;
  int64_t total;
};

int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

RrdbLayout rrdb_layout(const neosr_rrdbnet_cfg& c, void* ws) {
  RrdbLayout L;
  L.B = c.B; L.H = c.H; L.W = c.W;
  L.Cin = c.num_in_ch; L.Cout = c.num_out_ch; L.F = c.num_feat; L.G = c.num_grow_ch;
  L.NB = c.num_block;
  L.CC = L.F + 4 * L.G;
  L.cin_cs = pad4(L.Cin);
  L.cout_cs = pad4(L.Cout);
  L.np1 = (int64_t)c.B * c.H * c.W;
  L.np2 = L.np1 * 4;
  L.np4 = L.np1 * 16;
  Bump b(ws);
  L.x_nhwc = b.take(L.np1 * L.cin_cs);
  L.nact = c.training ? 3 * L.NB : 5;  // inference: act[0] (conv_first, needed by the skip) + ring of 4
  L.act.resize(L.nact);
  for (int i = 0; i < L.nact; ++i) L.act[i] = b.take(L.np1 * L.CC);
  L.trunk = b.take(L.np1 * L.F);
  L.fea = b.take(L.np1 * L.F);
  L.u1 = b.take(L.np2 * L.F);
  L.u2 = b.take(L.np4 * L.F);
  L.hr = b.take(L.np4 * L.F);
  L.y_nhwc = b.take(L.np4 * L.cout_cs);
  if (c.training) {
    L.gy_nhwc = L.y_nhwc;  // y_nhwc is dead after the final layout transform
    L.g_hr = b.take(L.np4 * L.F);
    L.g_u2 = b.take(L.np4 * L.F);
    L.g_up2in = L.g_hr;    // g_hr is dead once conv_hr's dgrad/wgrad have run
    L.g_u1 = b.take(L.np2 * L.F);
    L.g_up1in = b.take(L.np2 * L.F);
    L.g_fea = b.take(L.np1 * L.F);
    L.g_trunk = b.take(L.np1 * L.F);
    L.gx_nhwc = b.take(L.np1 * L.cin_cs);
    for (int i = 0; i < 4; ++i) L.gb[i] = b.take(L.np1 * L.CC);
    int64_t w = 0;
    const int B = c.B, H = c.H, W = c.W, F = L.F, G = L.G;
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, H, W, L.Cin, F));
    for (int k = 0; k < 4; ++k) w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, H, W, F + k * G, G));
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, H, W, L.CC, F));
    {  // the five convs of one RDB go in a single multi-conv launch
      neosr_wgrad_desc wd[5];
      for (int k = 0; k < 5; ++k) {
        wd[k] = wgrad_base(B, H, W);
        wd[k].in = wd[k].g = (const float*)16; wd[k].dw = (float*)16;
        wd[k].K = (k < 4) ? F + k * G : L.CC;
        wd[k].N = (k < 4) ? G : F;
      }
      w = max64(w, neosr_conv3x3_wgrad_multi_workspace_bytes(wd, 5));
      // ... and the fifteen of one RRDB (behind a chain launch, see rrdb_backward_impl)
      neosr_wgrad_desc wd3[15];
      for (int k = 0; k < 15; ++k) wd3[k] = wd[k % 5];
      w = max64(w, neosr_conv3x3_wgrad_multi_workspace_bytes(wd3, 15));
    }
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, H, W, F, F));
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, 2 * H, 2 * W, F, F));
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, 4 * H, 4 * W, F, F));
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(B, 4 * H, 4 * W, F, L.Cout));
    L.wg_ws = b.take(w / 4 + 64);
    L.wg_ws2 = b.take(w / 4 + 64);  // second half-batch chain
  }
  {
    const int F = L.F, G = L.G;
    int64_t o = 0;
    for (int k = 0; k < 5; ++k) {
      L.pf_off[k] = o;
      o += neosr_pack::image_floats(k < 4 ? G : F, F + k * G);
    }
    L.pf_total = o;
    o = 0;
    for (int j = 0; j < 5; ++j) {  // slice 0 = x (F channels, all five convs); slice j = x_j
      L.pd_off[j] = o;
      o += neosr_pack::image_floats(j == 0 ? F : G, F + (j == 0 ? 4 : 4 - j) * G);
    }
    L.pd_total = o;
    L.wpack_f = b.take(3 * L.NB * L.pf_total);
    L.wpack_d = c.training ? b.take(3 * L.NB * L.pd_total) : nullptr;
    o = 0;
    for (int k = 0; k < 5; ++k) {
      L.wf_off[k] = o;
      o += neosr_pack::wino_image_floats(k < 4 ? G : F, F + k * G);
    }
    L.wf_total = o;
    o = 0;
    for (int j = 0; j < 5; ++j) {
      L.wd_off[j] = o;
      o += neosr_pack::wino_image_floats(j == 0 ? F : G, F + (j == 0 ? 4 : 4 - j) * G);
    }
    L.wd_total = o;
    L.wwino_f = b.take(3 * L.NB * L.wf_total);
    L.wwino_d = c.training ? b.take(3 * L.NB * L.wd_total) : nullptr;
    L.tail_p = neosr_pack::image_floats(F, F);
    L.tail_w = neosr_pack::wino_image_floats(F, F);
    L.tail_pf = b.take(4 * L.tail_p);
    L.tail_wf = b.take(4 * L.tail_w);
    L.tail_pd = c.training ? b.take(4 * L.tail_p) : nullptr;
    L.tail_wd = c.training ? b.take(4 * L.tail_w) : nullptr;
    o = 0;
    for (int k = 0; k < 5; ++k) {
      L.w4f_off[k] = o;
      o += neosr_pack::wino4_image_floats(k < 4 ? G : F, F + k * G);
    }
    L.w4f_total = o;
    o = 0;
    for (int j = 0; j < 5; ++j) {
      L.w4d_off[j] = o;
      o += neosr_pack::wino4_image_floats(j == 0 ? F : G, F + (j == 0 ? 4 : 4 - j) * G);
    }
    L.w4d_total = o;
    L.wwino4_f = b.take(3 * L.NB * L.w4f_total);
    L.wwino4_d = c.training ? b.take(3 * L.NB * L.w4d_total) : nullptr;
    L.tail_w4 = neosr_pack::wino4_image_floats(F, F);
    L.tail_w4f = b.take(4 * L.tail_w4);
    L.tail_w4d = c.training ? b.take(4 * L.tail_w4) : nullptr;
    L.w4_trunk = neosr_conv::wino_mode() == 2 &&
                 (int64_t)c.B * ((c.H + 15) / 16) * ((c.W + 15) / 16) * ((G + 31) / 32) >= NEOSR_WINO4_MIN_WGS;
    L.direct_trunk = neosr_conv::wino_mode() == 0 || (int64_t)c.B * c.H * c.W * L.CC * 4 >= (int64_t(1) << 31);
  }
  {
    const int64_t tps = (int64_t)((c.H + 15) / 16) * ((c.W + 15) / 16);   // tiles per sample
    const int64_t cap = neosr_conv::chain_max_tiles();
    L.chain_gb = tps <= cap ? (int)(cap / tps < c.B ? cap / tps : c.B) : 0;  // samples per launch (0: a sample alone is too big)
    L.chain_groups = L.chain_gb ? (c.B + L.chain_gb - 1) / L.chain_gb : 0;
    L.chain_flag_words = (tps * (L.chain_gb ? L.chain_gb : 1) + 63) & ~(int64_t)63;
    const int ng = L.chain_groups ? L.chain_groups : 1;
    L.chain_flags = b.take(2 * L.NB * ng * L.chain_flag_words);
  }
  L.total = ((b.off + 255) & ~(int64_t)255);
  return L;
}

int rrdb_check(const neosr_rrdbnet_cfg* c) {
  NEOSR_CHECK(c, "rrdbnet: null cfg");
  NEOSR_CHECK(c->B > 0 && c->H > 0 && c->W > 0 && c->num_in_ch > 0 && c->num_out_ch > 0 &&
                  c->num_feat > 0 && c->num_block > 0 && c->num_grow_ch > 0,
              "rrdbnet: bad cfg");
  NEOSR_CHECK(c->num_feat % 4 == 0 && c->num_grow_ch % 4 == 0,
              "rrdbnet: num_feat and num_grow_ch must be multiples of 4");
  return 0;
}

// activation buffer of RDB i (i = 3*n + r); inference keeps slot 0 and recycles a ring of 4
inline int act_idx(const RrdbLayout& L, int i) {
  if (L.nact == 3 * L.NB || i == 0) return i;
  return 1 + ((i - 1) & 3);
}

// parameter index helpers (named_parameters order)
inline int p_first() { return 0; }
inline int p_rdb(int n, int r, int k) { return 2 + ((n * 3 + r) * 5 + k) * 2; }
inline int p_tail(const RrdbLayout& L, int i) { return 2 + L.NB * 30 + 2 * i; }  // body,up1,up2,hr,last

// Second stream for the RRDB trunk: at B = 16 a 64x64 conv is ONE round of 512 workgroups, so every
// launch pays its fill / drain / first-load latency in full (~8 us of 30-60).  Running the two halves of
// the batch as independent launch chains on two streams lets one chain's matrix work cover the other's
// launch gap (measured on the forward chain: 110 -> 121 TFLOP/s).  Fork / join with events only.
constexpr int MAX_CHAINS = 4;
struct Aux {
  int dev = -1;
  hipStream_t sc[MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};  // sc[1..]: chains 1.. (chain 0 = the caller's stream)
  hipStream_t s3 = nullptr;                                            // weight gradients
  hipEvent_t fork = nullptr, join[MAX_CHAINS] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> ev;
};
int g_wgrad_rrdb = -1;   // -1: read NEOSR_AMD_WGRAD_RRDB on first use (default on), see neosr_set_wgrad_rrdb
int g_num_streams = -1;  // -1: read NEOSR_AMD_STREAMS on first use (default 2)

Aux* aux_get(int nev) {
  static Aux a;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (g_num_streams < 0) {
    const char* e = getenv("NEOSR_AMD_STREAMS");
    const int n = e ? atoi(e) : 2;
    g_num_streams = n < 1 ? 2 : (n > MAX_CHAINS ? MAX_CHAINS : n);
  }
  if (g_num_streams < 2) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (a.s3 && a.dev != dev) {  // streams / events belong to the device they were created on
    a = Aux();
  }
  a.dev = dev;
  if (!a.s3) {
    for (int i = 1; i < MAX_CHAINS; ++i)
      if (hipStreamCreateWithFlags(&a.sc[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipStreamCreateWithFlags(&a.s3, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&a.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < MAX_CHAINS; ++i)
      if (hipEventCreateWithFlags(&a.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  while ((int)a.ev.size() < nev) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    a.ev.push_back(e);
  }
  return &a;
}

// Gradient buffer of one RDB (CC channels, "G order"): [g5 (F) | g4 (G) | g3 | g2 | g1], g_m = gradient
// wrt the pre-activation output of conv m in units of the RDB's residual scale.  Every slice of the
// concat buffer is then consumed by a PREFIX of it: x_j (j = 1..4) by conv5..conv(j+1) = channels
// [0, F + (4-j) G), x by all five.
inline int g_off(int F, int G, int m) { return m == 5 ? 0 : F + (4 - m) * G; }

// images of the four F -> F convs behind the trunk (conv_body, conv_up1, conv_up2, conv_hr), direct then Winograd
int rrdb_pack_tail(const RrdbLayout& L, const float* const* P, int mode, void* st) {
  neosr_pack::Image imgs[4];
  memset(imgs, 0, sizeof(imgs));
  for (int i = 0; i < 4; ++i) {
    neosr_pack::Image& im = imgs[i];
    im.dst = (mode == NEOSR_CONV_FWD ? L.tail_pf : L.tail_pd) + i * L.tail_p;
    im.N = L.F; im.K = L.F; im.mode = mode; im.nseg = 1;
    im.seg[0].w = P[2 + L.NB * 30 + 2 * i];
    im.seg[0].w_cin = L.F;
    im.seg[0].k_cnt = L.F;
  }
  RUN(neosr_pack::launch(imgs, 4, st));
  for (int i = 0; i < 4; ++i) imgs[i].dst = (mode == NEOSR_CONV_FWD ? L.tail_wf : L.tail_wd) + i * L.tail_w;
  RUN(neosr_pack::launch_wino(imgs, 4, st));  // (small launches — e.g. B = 1 — fall back to F(2x2,3x3): both kinds are kept)
  if (neosr_conv::wino_mode() != 2) return 0;
  for (int i = 0; i < 4; ++i) imgs[i].dst = (mode == NEOSR_CONV_FWD ? L.tail_w4f : L.tail_w4d) + i * L.tail_w4;
  return neosr_pack::launch_wino4(imgs, 4, st);
}

int rrdb_pack_fwd(const RrdbLayout& L, const float* const* P, void* st) {
  RUN(rrdb_pack_tail(L, P, NEOSR_CONV_FWD, st));
  std::vector<neosr_pack::Image> imgs(3 * L.NB * 5);
  memset(imgs.data(), 0, imgs.size() * sizeof(neosr_pack::Image));
  for (int i = 0; i < 3 * L.NB; ++i)
    for (int k = 0; k < 5; ++k) {
      neosr_pack::Image& im = imgs[i * 5 + k];
      im.dst = L.wpack_f + i * L.pf_total + L.pf_off[k];
      im.N = k < 4 ? L.G : L.F;
      im.K = L.F + k * L.G;
      im.mode = NEOSR_CONV_FWD;
      im.nseg = 1;
      im.seg[0].w = P[p_rdb(i / 3, i % 3, k)];
      im.seg[0].w_cin = im.K;
      im.seg[0].k_cnt = im.K;
    }
  if (L.direct_trunk) RUN(neosr_pack::launch(imgs.data(), (int)imgs.size(), st));
  for (int i = 0; i < 3 * L.NB; ++i)
    for (int k = 0; k < 5; ++k)
      imgs[i * 5 + k].dst = L.w4_trunk ? L.wwino4_f + i * L.w4f_total + L.w4f_off[k]
                                       : L.wwino_f + i * L.wf_total + L.wf_off[k];
  return L.w4_trunk ? neosr_pack::launch_wino4(imgs.data(), (int)imgs.size(), st)
                    : neosr_pack::launch_wino(imgs.data(), (int)imgs.size(), st);
}

int rrdb_pack_dgrad(const RrdbLayout& L, const float* const* P, void* st) {
  RUN(rrdb_pack_tail(L, P, NEOSR_CONV_DGRAD, st));
  const int F = L.F, G = L.G;
  std::vector<neosr_pack::Image> imgs(3 * L.NB * 5);
  memset(imgs.data(), 0, imgs.size() * sizeof(neosr_pack::Image));
  for (int i = 0; i < 3 * L.NB; ++i)
    for (int j = 0; j < 5; ++j) {
      neosr_pack::Image& im = imgs[i * 5 + j];
      im.dst = L.wpack_d + i * L.pd_total + L.pd_off[j];
      im.N = j == 0 ? F : G;
      im.K = F + (j == 0 ? 4 : 4 - j) * G;
      im.mode = NEOSR_CONV_DGRAD;
      const int n_lo = j == 0 ? 0 : F + (j - 1) * G;
      int ns = 0;
      for (int m = 5; m > j && m >= 1; --m) {  // consumers of this slice, in G order
        neosr_pack::Seg& sg = im.seg[ns++];
        sg.w = P[p_rdb(i / 3, i % 3, m - 1)];
        sg.w_cin = F + (m - 1) * G;
        sg.k_lo = g_off(F, G, m);
        sg.k_cnt = m == 5 ? F : G;
        sg.n_lo = n_lo;
      }
      im.nseg = ns;
    }
  if (L.direct_trunk) RUN(neosr_pack::launch(imgs.data(), (int)imgs.size(), st));
  for (int i = 0; i < 3 * L.NB; ++i)
    for (int j = 0; j < 5; ++j)
      imgs[i * 5 + j].dst = L.w4_trunk ? L.wwino4_d + i * L.w4d_total + L.w4d_off[j]
                                       : L.wwino_d + i * L.wd_total + L.wd_off[j];
  return L.w4_trunk ? neosr_pack::launch_wino4(imgs.data(), (int)imgs.size(), st)
                    : neosr_pack::launch_wino(imgs.data(), (int)imgs.size(), st);
}

}  // namespace

extern "C" int32_t neosr_rrdbnet_num_params(const neosr_rrdbnet_cfg* c) {
  return 2 + c->num_block * 30 + 10;
}

extern "C" int64_t neosr_rrdbnet_workspace_bytes(const neosr_rrdbnet_cfg* c) {
  if (rrdb_check(c)) return -1;
  return rrdb_layout(*c, nullptr).total;
}

extern "C" int neosr_rrdbnet_forward(const neosr_rrdbnet_cfg* c, const float* const* P,
                                     const float* x, float* y, void* ws, void* st) {
  RUN(rrdb_check(c));
  NEOSR_CHECK(P && x && y && ws, "rrdbnet_forward: null pointer");
  const RrdbLayout L = rrdb_layout(*c, ws);
  const int B = L.B, H = L.H, W = L.W, F = L.F, G = L.G, CC = L.CC;
  RUN(neosr_nchw_to_nhwc(x, L.x_nhwc, B, L.Cin, H, W, L.cin_cs, st));
  {
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.x_nhwc; d.in_cs = L.cin_cs; d.K = L.Cin;
    d.w = P[0]; d.bias = P[1]; d.w_cout = F; d.w_cin = L.Cin;
    d.out = L.act[0]; d.out_cs = CC; d.N = F;
    RUN(neosr_conv3x3(&d, st));
  }
  RUN(rrdb_pack_fwd(L, P, st));
  // trunk.  One descriptor per (RRDB n, RDB r, conv k) over the samples [b0, b0 + nb)
  auto trunk_desc = [&](int n, int r, int k, int b0, int nb) {
    const float* pk = L.wpack_f + (int64_t)(3 * n + r) * L.pf_total;
    const float* wk = L.wwino_f + (int64_t)(3 * n + r) * L.wf_total;
    const float* wk4 = L.wwino4_f + (int64_t)(3 * n + r) * L.w4f_total;
    const int64_t po = (int64_t)b0 * H * W;  // pixel offset of these samples
    float* A = L.act[act_idx(L, 3 * n + r)] + po * CC;
    neosr_conv_desc d = conv_base(nb, H, W);
    d.in = A; d.in_cs = CC; d.K = F + k * G;
    d.w = P[p_rdb(n, r, k)]; d.bias = P[p_rdb(n, r, k) + 1]; d.w_cin = F + k * G;
    d.w_pack = pk + L.pf_off[k];
    d.w_wino = L.w4_trunk ? nullptr : wk + L.wf_off[k];   // (only the image kind that was packed is offered)
    d.w_wino4 = L.w4_trunk ? wk4 + L.w4f_off[k] : nullptr;
    if (k < 4) {
      d.w_cout = G;
      d.out = A + F + k * G; d.out_cs = CC; d.N = G;
      d.act = NEOSR_ACT_LRELU; d.slope = 0.2f;
    } else {
      d.w_cout = F;
      const bool last = (n == L.NB - 1 && r == 2);
      d.out = last ? L.trunk + po * F : L.act[act_idx(L, 3 * n + r + 1)] + po * CC;
      d.out_cs = last ? F : CC;
      d.N = F;
      d.alpha = 0.2f; d.res1 = A; d.res1_cs = CC; d.res1_nch = F;
      if (r == 2) {
        d.alpha2 = 0.2f; d.res2 = L.act[act_idx(L, 3 * n)] + po * CC; d.res2_cs = CC; d.res2_nch = F;
      }
    }
    return d;
  };
  // (a) every RRDB as ONE launch of the chain kernel (conv_wino4_chain.hip): conv k of an RDB depends on the previous
  // layer only through the newest slice of the concat buffer = the chunk that starts at channel F + (k - 1) G; conv1
  // reads the previous RDB's output: everything is new (dep 0)
  bool chained = false;
  if (L.w4_trunk && L.chain_groups && neosr_conv::chain_enabled()) {
    const int ng = L.chain_groups;
    NEOSR_HIP(hipMemsetAsync(L.chain_flags, 0, (size_t)L.NB * ng * L.chain_flag_words * 4, (hipStream_t)st));
    for (int n = 0; n < L.NB && (n == 0 || chained); ++n)
      for (int gi = 0; gi < ng; ++gi) {
        const int b0 = gi * L.chain_gb, nb = B - b0 < L.chain_gb ? B - b0 : L.chain_gb;
        neosr_conv_desc dd[15];
        int dep[15];
        for (int r = 0; r < 3; ++r)
          for (int k = 0; k < 5; ++k) {
            dd[r * 5 + k] = trunk_desc(n, r, k, b0, nb);
            dep[r * 5 + k] = k == 0 ? 0 : (F + (k - 1) * G) / 32;
          }
        const int rc = neosr_conv::launch_wino4_chain(dd, dep, 15, (unsigned*)L.chain_flags + (n * ng + gi) * L.chain_flag_words, st);
        if (rc > 0) return rc;
        if (rc < 0) {
          NEOSR_CHECK(n == 0 && gi == 0, "rrdbnet_forward: the chain kernel refused RRDB %d after taking RRDB 0", n);
          break;
        }
        chained = true;
      }
  }
  // (b) one launch per conv: the two halves of the batch are independent launch chains (see Aux)
  Aux* ax = (!chained && B >= 2) ? aux_get(0) : nullptr;
  const int nhalf = ax ? (g_num_streams < B ? g_num_streams : B) : 1;  // number of launch chains
  if (ax) {
    NEOSR_HIP(hipEventRecord(ax->fork, (hipStream_t)st));
    for (int h = 1; h < nhalf; ++h) NEOSR_HIP(hipStreamWaitEvent(ax->sc[h], ax->fork, 0));
  }
  if (!chained) {
  ChainHint hint(nhalf);
  for (int n = 0; n < L.NB; ++n) {
    for (int r = 0; r < 3; ++r) {
      for (int h = 0; h < nhalf; ++h) {
        const int b0 = (int)((int64_t)B * h / nhalf), nb = (int)((int64_t)B * (h + 1) / nhalf) - b0;
        void* sh = h ? (void*)ax->sc[h] : st;
        for (int k = 0; k < 5; ++k) {
          const neosr_conv_desc d = trunk_desc(n, r, k, b0, nb);
          RUN(neosr_conv3x3(&d, sh));
        }
      }
    }
  }
  }
  if (ax)
    for (int h = 1; h < nhalf; ++h) {
      NEOSR_HIP(hipEventRecord(ax->join[h], ax->sc[h]));
      NEOSR_HIP(hipStreamWaitEvent((hipStream_t)st, ax->join[h], 0));
    }
  {  // conv_body + skip
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.trunk; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 0)]; d.bias = P[p_tail(L, 0) + 1]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pf; d.w_wino = L.tail_wf; d.w_wino4 = L.tail_w4f;
    d.out = L.fea; d.out_cs = F; d.N = F;
    d.res1 = L.act[0]; d.res1_cs = CC; d.res1_nch = F;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_up1 on nearest x2
    neosr_conv_desc d = conv_base(B, 2 * H, 2 * W);
    d.ups = 1; d.in = L.fea; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 1)]; d.bias = P[p_tail(L, 1) + 1]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pf + L.tail_p; d.w_wino = L.tail_wf + L.tail_w; d.w_wino4 = L.tail_w4f + L.tail_w4;
    d.out = L.u1; d.out_cs = F; d.N = F; d.act = NEOSR_ACT_LRELU; d.slope = 0.2f;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_up2 on nearest x2
    neosr_conv_desc d = conv_base(B, 4 * H, 4 * W);
    d.ups = 1; d.in = L.u1; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 2)]; d.bias = P[p_tail(L, 2) + 1]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pf + 2 * L.tail_p; d.w_wino = L.tail_wf + 2 * L.tail_w; d.w_wino4 = L.tail_w4f + 2 * L.tail_w4;
    d.out = L.u2; d.out_cs = F; d.N = F; d.act = NEOSR_ACT_LRELU; d.slope = 0.2f;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_hr
    neosr_conv_desc d = conv_base(B, 4 * H, 4 * W);
    d.in = L.u2; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 3)]; d.bias = P[p_tail(L, 3) + 1]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pf + 3 * L.tail_p; d.w_wino = L.tail_wf + 3 * L.tail_w; d.w_wino4 = L.tail_w4f + 3 * L.tail_w4;
    d.out = L.hr; d.out_cs = F; d.N = F; d.act = NEOSR_ACT_LRELU; d.slope = 0.2f;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_last
    neosr_conv_desc d = conv_base(B, 4 * H, 4 * W);
    d.in = L.hr; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 4)]; d.bias = P[p_tail(L, 4) + 1]; d.w_cout = L.Cout; d.w_cin = F;
    d.out = L.y_nhwc; d.out_cs = L.cout_cs; d.N = L.Cout;
    RUN(neosr_conv3x3(&d, st));
  }
  RUN(neosr_nhwc_to_nchw(L.y_nhwc, y, B, L.Cout, 4 * H, 4 * W, L.cout_cs, st));
  return 0;
}

namespace {
int rrdb_backward_impl(const neosr_rrdbnet_cfg* c, const float* const* P, float* const* Gp, const float* gy,
                       float* gx, void* ws, void* st, int n_marks, const int32_t* mark_block,
                       void* const* mark_event);
}

extern "C" int neosr_rrdbnet_backward(const neosr_rrdbnet_cfg* c, const float* const* P,
                                      float* const* Gp, const float* gy, float* gx, void* ws,
                                      void* st) {
  return rrdb_backward_impl(c, P, Gp, gy, gx, ws, st, 0, nullptr, nullptr);
}

extern "C" int neosr_rrdbnet_backward_marked(const neosr_rrdbnet_cfg* c, const float* const* P,
                                             float* const* Gp, const float* gy, float* gx, void* ws,
                                             void* st, int32_t n_marks, const int32_t* mark_block,
                                             void* const* mark_event) {
  NEOSR_CHECK(n_marks >= 0 && (n_marks == 0 || (mark_block && mark_event)), "rrdbnet_backward_marked: bad marks");
  for (int i = 0; i < n_marks; ++i)
    NEOSR_CHECK(c && mark_block[i] >= 0 && mark_block[i] < c->num_block && mark_event[i],
                "rrdbnet_backward_marked: mark %d out of range", i);
  return rrdb_backward_impl(c, P, Gp, gy, gx, ws, st, n_marks, mark_block, mark_event);
}

namespace {
int rrdb_backward_impl(const neosr_rrdbnet_cfg* c, const float* const* P, float* const* Gp, const float* gy,
                       float* gx, void* ws, void* st, int n_marks, const int32_t* mark_block,
                       void* const* mark_event) {
  RUN(rrdb_check(c));
  NEOSR_CHECK(P && Gp && gy && ws, "rrdbnet_backward: null pointer");
  NEOSR_CHECK(c->training, "rrdbnet_backward: cfg.training must be set (activations are needed)");
  const RrdbLayout L = rrdb_layout(*c, ws);
  const int B = L.B, H = L.H, W = L.W, F = L.F, G = L.G, CC = L.CC;
  const int H2 = 2 * H, W2 = 2 * W, H4 = 4 * H, W4 = 4 * W;

  RUN(neosr_nchw_to_nhwc(gy, L.gy_nhwc, B, L.Cout, H4, W4, L.cout_cs, st));
  RUN(rrdb_pack_dgrad(L, P, st));
  // Behind the trunk every gradient leaves its producer as dL/d(pre-activation): the LeakyReLU derivative of the layer
  // below is applied in the producer's epilogue (out_mask) or in the 2x2 pool, so no consumer masks on load and the
  // backward-data / weight-gradient launches are plain -> packed / Winograd kernels (as in the RDB trunk).
  {  // conv_last: g_hr = dgrad * lrelu'(hr)
    neosr_wgrad_desc w = wgrad_base(B, H4, W4);
    w.in = L.hr; w.in_cs = F; w.K = F; w.g = L.gy_nhwc; w.g_cs = L.cout_cs; w.N = L.Cout;
    w.dw = Gp[p_tail(L, 4)]; w.db = Gp[p_tail(L, 4) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H4, W4);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.gy_nhwc; d.in_cs = L.cout_cs; d.K = L.Cout;
    d.w = P[p_tail(L, 4)]; d.w_cout = L.Cout; d.w_cin = F;
    d.out = L.g_hr; d.out_cs = F; d.N = F;
    d.out_mask = L.hr; d.out_mask_cs = F; d.out_mask_slope = 0.2f;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_hr: g_u2 = dgrad * lrelu'(u2)
    neosr_wgrad_desc w = wgrad_base(B, H4, W4);
    w.in = L.u2; w.in_cs = F; w.K = F; w.g = L.g_hr; w.g_cs = F; w.N = F;
    w.dw = Gp[p_tail(L, 3)]; w.db = Gp[p_tail(L, 3) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H4, W4);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_hr; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 3)]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pd + 3 * L.tail_p; d.w_wino = L.tail_wd + 3 * L.tail_w; d.w_wino4 = L.tail_w4d + 3 * L.tail_w4;
    d.out = L.g_u2; d.out_cs = F; d.N = F;
    d.out_mask = L.u2; d.out_mask_cs = F; d.out_mask_slope = 0.2f;
    RUN(neosr_conv3x3(&d, st));
  }
  {  // conv_up2 (input = nearest x2 of u1): g_u1 = pool(dgrad) * lrelu'(u1)
    neosr_wgrad_desc w = wgrad_base(B, H4, W4);
    w.ups = 1; w.in = L.u1; w.in_cs = F; w.K = F; w.g = L.g_u2; w.g_cs = F; w.N = F;
    w.dw = Gp[p_tail(L, 2)]; w.db = Gp[p_tail(L, 2) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H4, W4);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_u2; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 2)]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pd + 2 * L.tail_p; d.w_wino = L.tail_wd + 2 * L.tail_w; d.w_wino4 = L.tail_w4d + 2 * L.tail_w4;
    d.out = L.g_up2in; d.out_cs = F; d.N = F;
    RUN(neosr_conv3x3(&d, st));
    RUN(neosr_pool2x2_sum_masked(L.g_up2in, L.g_u1, L.u1, B, H2, W2, F, F, F, F, 0.2f, st));
  }
  {  // conv_up1 (input = nearest x2 of fea; fea has no activation)
    neosr_wgrad_desc w = wgrad_base(B, H2, W2);
    w.ups = 1; w.in = L.fea; w.in_cs = F; w.K = F; w.g = L.g_u1; w.g_cs = F; w.N = F;
    w.dw = Gp[p_tail(L, 1)]; w.db = Gp[p_tail(L, 1) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H2, W2);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_u1; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 1)]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pd + L.tail_p; d.w_wino = L.tail_wd + L.tail_w; d.w_wino4 = L.tail_w4d + L.tail_w4;
    d.out = L.g_up1in; d.out_cs = F; d.N = F;
    RUN(neosr_conv3x3(&d, st));
    RUN(neosr_pool2x2_sum(L.g_up1in, L.g_fea, B, H, W, F, F, F, 0, st));
  }
  {  // conv_body
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.trunk; w.in_cs = F; w.K = F; w.g = L.g_fea; w.g_cs = F; w.N = F;
    w.dw = Gp[p_tail(L, 0)]; w.db = Gp[p_tail(L, 0) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H, W);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_fea; d.in_cs = F; d.K = F;
    d.w = P[p_tail(L, 0)]; d.w_cout = F; d.w_cin = F;
    d.w_pack = L.tail_pd; d.w_wino = L.tail_wd; d.w_wino4 = L.tail_w4d;
    d.out = L.gb[0]; d.out_cs = CC; d.N = F;  // = g5 of the last RDB
    RUN(neosr_conv3x3(&d, st));
  }
  // trunk: 23 x RRDB, reversed
  // Gather form: each gradient slice of the concat buffer is produced ONCE, by a forward-shaped
  // convolution over a prefix of the RDB's gradient buffer (see g_off), with the LeakyReLU derivative
  // of that slice applied in the epilogue -> no read-modify-write, no masks on load, K = 64..192.
  // Two launch chains again (see Aux) for the data gradients; the weight gradients (one full-batch
  // launch per RDB, independent of the chain that follows) run on a third stream behind them.  The ring
  // of four gradient buffers bounds how far the chains may run ahead: RDB t+3 overwrites the g5 slot of
  // the buffer RDB t's weight gradient reads.
  const int NR = 3 * L.NB;
  // data-gradient descriptor of (RRDB n, RDB r, slice j) over the samples [b0, b0 + nb); gbi = ring slot of the RDB's buffer
  auto dgrad_desc = [&](int n, int r, int j, int gbi, const float* dOut, int b0, int nb) {
    const float* pk = L.wpack_d + (int64_t)(3 * n + r) * L.pd_total;
    const float* wk = L.wwino_d + (int64_t)(3 * n + r) * L.wd_total;
    const float* wk4 = L.wwino4_d + (int64_t)(3 * n + r) * L.w4d_total;
    const int64_t po = (int64_t)b0 * H * W * CC;
    const float* A = L.act[3 * n + r] + po;
    float* GB = L.gb[gbi] + po;
    float* NG = L.gb[(gbi + 1) & 3] + po;
    neosr_conv_desc d = conv_base(nb, H, W);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = GB; d.in_cs = CC;
    d.w_pack = pk + L.pd_off[j];
    d.w_wino = L.w4_trunk ? nullptr : wk + L.wd_off[j];
    d.w_wino4 = L.w4_trunk ? wk4 + L.w4d_off[j] : nullptr;
    if (j >= 1) {  // g_j = lrelu'(x_j) * sum over conv5..conv(j+1)
      d.K = F + (4 - j) * G;
      d.out = GB + g_off(F, G, j); d.out_cs = CC; d.N = G;
      d.out_mask = A + F + (j - 1) * G; d.out_mask_cs = CC; d.out_mask_slope = 0.2f;
    } else {       // gradient wrt the RDB input -> g5 slot of the next RDB's buffer
      d.K = CC;
      d.out = NG; d.out_cs = CC; d.N = F;
      d.alpha = 0.2f; d.res1 = GB; d.res1_cs = CC; d.res1_nch = F;
      if (r == 2) d.alpha2 = 0.2f;
      if (r == 0) { d.res2 = dOut + po; d.res2_cs = CC; d.res2_nch = F; }
    }
    return d;
  };
  auto wgrad_desc_of = [&](int n, int r, int gbi, int m) {
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.act[3 * n + r]; w.in_cs = CC; w.K = F + (m - 1) * G;
    w.g = L.gb[gbi] + g_off(F, G, m); w.g_cs = CC; w.N = m == 5 ? F : G;
    w.scale = (r == 2) ? 0.04f : 0.2f;
    w.dw = Gp[p_rdb(n, r, m - 1)]; w.db = Gp[p_rdb(n, r, m - 1) + 1];
    return w;
  };
  auto wgrad_rdb = [&](int n, int r, int gbi, void* sw_) -> int {  // all five weight gradients of this RDB in one launch
    neosr_wgrad_desc wd[5];
    for (int m = 1; m <= 5; ++m) wd[m - 1] = wgrad_desc_of(n, r, gbi, m);
    return neosr_conv3x3_wgrad_multi(wd, 5, L.wg_ws, sw_);
  };
  if (g_wgrad_rrdb < 0) {
    const char* e = getenv("NEOSR_AMD_WGRAD_RRDB");
    g_wgrad_rrdb = (e && e[0] == '0') ? 0 : 1;
  }
  const bool wgrad15 = g_wgrad_rrdb == 1;
  int gbi = 0, t = 0;
  const float* dOut = nullptr;  // gradient wrt the output of the current RRDB (g5 slot of its last RDB)
  float* prev = nullptr;
  // (a) the fifteen data-gradient convolutions of an RRDB as ONE launch of the chain kernel (conv_wino4_chain.hip),
  // its three weight-gradient launches behind it on the same stream (a chain workgroup and a weight-gradient workgroup
  // do not fit one CU together, so a second stream could only interleave them workgroup by workgroup).  Slice j of an
  // RDB's gradient buffer is produced from the prefix in front of it: the newest slice is the last chunk again, and
  // g4 of the next RDB reads the g5 slot the previous layer wrote (dep 0).
  bool chained = false;
  if (L.w4_trunk && L.chain_groups && neosr_conv::chain_enabled()) {
    const int ng = L.chain_groups;
    unsigned* flags = (unsigned*)L.chain_flags + (int64_t)L.NB * ng * L.chain_flag_words;
    NEOSR_HIP(hipMemsetAsync(flags, 0, (size_t)L.NB * ng * L.chain_flag_words * 4, (hipStream_t)st));
    for (int n = L.NB - 1; n >= 0; --n) {
      const int g0 = gbi;
      dOut = L.gb[g0];
      int rc = 0;
      for (int gi = 0; gi < ng && rc == 0; ++gi) {
        const int b0 = gi * L.chain_gb, nb = B - b0 < L.chain_gb ? B - b0 : L.chain_gb;
        neosr_conv_desc dd[15];
        int dep[15];
        for (int r = 2, i = 0; r >= 0; --r)
          for (int j = 4; j >= 0; --j, ++i) {
            dd[i] = dgrad_desc(n, r, j, (g0 + (2 - r)) & 3, dOut, b0, nb);
            dep[i] = j == 4 ? 0 : (F + (j == 0 ? 3 : 3 - j) * G) / 32;
          }
        rc = neosr_conv::launch_wino4_chain(dd, dep, 15, flags + (n * ng + gi) * L.chain_flag_words, st);
        if (rc < 0) NEOSR_CHECK(n == L.NB - 1 && gi == 0, "rrdbnet_backward: the chain kernel refused RRDB %d after taking the last one", n);
      }
      if (rc > 0) return rc;
      if (rc < 0) break;
      chained = true;
      if (wgrad15) {  // the fifteen weight gradients of the RRDB in one launch
        neosr_wgrad_desc wd[15];
        for (int r = 2, i = 0; r >= 0; --r)
          for (int m = 1; m <= 5; ++m, ++i) wd[i] = wgrad_desc_of(n, r, (g0 + (2 - r)) & 3, m);
        RUN(neosr_conv3x3_wgrad_multi(wd, 15, L.wg_ws, st));
      } else {
        for (int r = 2; r >= 0; --r) RUN(wgrad_rdb(n, r, (g0 + (2 - r)) & 3, st));
      }
      for (int i = 0; i < n_marks; ++i)
        if (mark_block[i] == n) NEOSR_HIP(hipEventRecord((hipEvent_t)mark_event[i], (hipStream_t)st));
      gbi = (g0 + 3) & 3;
      prev = L.gb[gbi];
    }
  }
  // (b) one launch per convolution.
  // Two launch chains again (see Aux) for the data gradients; the weight gradients (one full-batch
  // launch per RDB, independent of the chain that follows) run on a third stream behind them.  The ring
  // of four gradient buffers bounds how far the chains may run ahead: RDB t+3 overwrites the g5 slot of
  // the buffer RDB t's weight gradient reads.
  Aux* ax = (!chained && B >= 2) ? aux_get((MAX_CHAINS + 1) * NR) : nullptr;
  const int nhalf = ax ? (g_num_streams < B ? g_num_streams : B) : 1;  // number of launch chains
  void* sw = ax ? (void*)ax->s3 : st;  // weight-gradient stream
  if (ax) {
    NEOSR_HIP(hipEventRecord(ax->fork, (hipStream_t)st));
    for (int h = 1; h < nhalf; ++h) NEOSR_HIP(hipStreamWaitEvent(ax->sc[h], ax->fork, 0));
    NEOSR_HIP(hipStreamWaitEvent(ax->s3, ax->fork, 0));
  }
  if (!chained) {
  ChainHint hint(nhalf);
  for (int n = L.NB - 1; n >= 0; --n) {
    for (int r = 2; r >= 0; --r, ++t) {
      if (r == 2) dOut = L.gb[gbi];
      for (int h = 0; h < nhalf; ++h) {
        const int b0 = (int)((int64_t)B * h / nhalf), nb = (int)((int64_t)B * (h + 1) / nhalf) - b0;
        void* sh = h ? (void*)ax->sc[h] : st;
        for (int j = 4; j >= 0; --j) {
          if (j == 0 && ax && t >= 3) NEOSR_HIP(hipStreamWaitEvent((hipStream_t)sh, ax->ev[MAX_CHAINS * NR + t - 3], 0));
          const neosr_conv_desc d = dgrad_desc(n, r, j, gbi, dOut, b0, nb);
          RUN(neosr_conv3x3(&d, sh));
        }
        if (ax) {
          NEOSR_HIP(hipEventRecord(ax->ev[h * NR + t], (hipStream_t)sh));
          NEOSR_HIP(hipStreamWaitEvent(ax->s3, ax->ev[h * NR + t], 0));
        }
      }
      RUN(wgrad_rdb(n, r, gbi, sw));
      if (ax) NEOSR_HIP(hipEventRecord(ax->ev[MAX_CHAINS * NR + t], ax->s3));
      // gradient marks: the weight gradients of RRDB n and of everything behind it in the parameter order
      // (RRDBs n+1.., conv_body .. conv_last, which ran on `st` before the fork) are enqueued on `sw` -> the
      // caller may start reducing that suffix of the gradient arena once this event has completed
      if (r == 0)
        for (int i = 0; i < n_marks; ++i)
          if (mark_block[i] == n) NEOSR_HIP(hipEventRecord((hipEvent_t)mark_event[i], (hipStream_t)sw));
      prev = L.gb[(gbi + 1) & 3];
      gbi = (gbi + 1) & 3;
    }
  }
  }
  if (ax) {
    for (int h = 1; h < nhalf; ++h) {
      NEOSR_HIP(hipEventRecord(ax->join[h], ax->sc[h]));
      NEOSR_HIP(hipStreamWaitEvent((hipStream_t)st, ax->join[h], 0));
    }
    NEOSR_HIP(hipEventRecord(ax->join[0], ax->s3));
    NEOSR_HIP(hipStreamWaitEvent((hipStream_t)st, ax->join[0], 0));
  }
  // skip connection feat + body_feat, then conv_first
  RUN(neosr_axpy_slice(prev, L.g_fea, L.np1, F, CC, F, 1.0f, st));
  {
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.x_nhwc; w.in_cs = L.cin_cs; w.K = L.Cin; w.g = prev; w.g_cs = CC; w.N = F;
    w.dw = Gp[0]; w.db = Gp[1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    if (gx) {
      neosr_conv_desc d = conv_base(B, H, W);
      d.mode = NEOSR_CONV_DGRAD;
      d.in = prev; d.in_cs = CC; d.K = F;
      d.w = P[0]; d.w_cout = F; d.w_cin = L.Cin;
      d.out = L.gx_nhwc; d.out_cs = L.cin_cs; d.N = L.Cin;
      RUN(neosr_conv3x3(&d, st));
      RUN(neosr_nhwc_to_nchw(L.gx_nhwc, gx, B, L.Cin, H, W, L.cin_cs, st));
    }
  }
  return 0;
}
}  // namespace

// ------------------------------------------------------------------------------ SRVGGNetCompact
namespace {

struct CompactLayout {
  int B, H, W, Cin, Cout, F, NC, r, Clast, cin_cs, clast_cs, nz;
  int has_prelu;
  int64_t np;
  float* x_nhwc;
  std::vector<float*> z;  // pre-activation outputs of conv 0..NC
  float* zl;              // last conv output (Cout*r*r)
  float* slopes_const;    // constant slope vector for relu / leakyrelu
  float *g_zl, *dA[2], *wg_ws, *ps_ws;
  // Winograd scheme (compact_w4): post-activation outputs a_i next to the pre-activations z_i, the masked gradients
  // g_z_i of every layer (the body's weight gradients run as ONE launch at the end), F(4x4,3x3) images
  int w4;
  std::vector<float*> a, gz, dAall;
  float* psm_ws;
  float *img_f, *img_d;          // NC + 1 images each: convs 1..NC then the last conv
  float* wgm_ws;                 // workspace of the body's multi weight-gradient launch
  int64_t img_body, img_last_f, img_last_d;
  int64_t total;
};

// The conv + PReLU chain as plain F(4x4,3x3) launches (round 4).  Rounds 1-3 kept only the pre-activations z_i and applied
// the PReLU (forward) / its derivative (backward) ON LOAD — which only the staged direct kernel can do: 16 + 16 launches of
// 64 workgroups at 0.14 of the matrix peak and 17 masked weight-gradient launches per step.  Here the F(4x4,3x3) kernel's
// epilogue writes a_i = PReLU(z_i) (per-channel slopes) AND z_i (second output), the backward-data epilogue writes
// g_z_{i-1} = dA_{i-1} . PReLU'(z_{i-1}) AND dA_{i-1} (for the slope gradient), so every convolution and every weight
// gradient of the body is a plain Winograd launch and the NC body weight gradients are one neosr_conv3x3_wgrad_multi
// launch.  NEOSR_AMD_COMPACT_W4=0, a Winograd mode other than 2 or num_feat % 4 != 0 keep the on-load scheme; so does a PReLU net
// whose last convolution has a channel count that is not a multiple of 4 (upscale 3 -> 27, upscale 1 -> 3): its backward-data
// launch reduces over those channels and needs the Winograd kernel's masked epilogue, which the packed image (K % 4) does
// not cover.
bool compact_w4(const neosr_compact_cfg& c) {
  static const bool on = [] { const char* e = getenv("NEOSR_AMD_COMPACT_W4"); return !(e && e[0] == '0'); }();
  const int clast = c.num_out_ch * c.upscale * c.upscale;
  return on && neosr_conv::wino_mode() == 2 && c.num_feat % 4 == 0 && c.num_conv >= 1 &&
         (clast % 4 == 0 || c.act_type != NEOSR_ACT_PRELU) &&
         (int64_t)c.B * c.H * c.W * c.num_feat * 4 < (int64_t(1) << 31);
}

CompactLayout compact_layout(const neosr_compact_cfg& c, void* ws) {
  CompactLayout L;
  L.B = c.B; L.H = c.H; L.W = c.W; L.Cin = c.num_in_ch; L.Cout = c.num_out_ch;
  L.F = c.num_feat; L.NC = c.num_conv; L.r = c.upscale;
  L.Clast = L.Cout * L.r * L.r;
  L.cin_cs = pad4(L.Cin);
  L.clast_cs = pad4(L.Clast);
  L.has_prelu = c.act_type == NEOSR_ACT_PRELU;
  L.np = (int64_t)c.B * c.H * c.W;
  Bump b(ws);
  L.x_nhwc = b.take(L.np * L.cin_cs);
  L.slopes_const = b.take(L.F);
  L.nz = c.training ? L.NC + 1 : 2;
  L.z.resize(L.nz);
  for (int i = 0; i < L.nz; ++i) L.z[i] = b.take(L.np * L.F);
  L.zl = b.take(L.np * L.clast_cs);
  if (c.training) {
    L.g_zl = L.zl;  // dead after the pixel shuffle
    L.dA[0] = b.take(L.np * L.F);
    L.dA[1] = b.take(L.np * L.F);
    int64_t w = neosr_conv3x3_wgrad_workspace_bytes(c.B, c.H, c.W, L.Cin, L.F);
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(c.B, c.H, c.W, L.F, L.F));
    w = max64(w, neosr_conv3x3_wgrad_workspace_bytes(c.B, c.H, c.W, L.F, L.Clast));
    L.wg_ws = b.take(w / 4 + 64);
    L.ps_ws = b.take(neosr_prelu_dslope_workspace_bytes(L.np, L.F) / 4 + 64);
  }
  L.w4 = compact_w4(c) ? 1 : 0;
  if (L.w4) {
    L.a.resize(c.training ? L.NC + 1 : 2);   // (inference: a ring of two)
    for (auto& p : L.a) p = b.take(L.np * L.F);
    L.img_body = neosr_pack::wino4_image_floats(L.F, L.F);
    L.img_last_f = neosr_pack::wino4_image_floats(L.Clast, L.F);
    L.img_last_d = neosr_pack::wino4_image_floats(L.F, L.Clast);
    L.img_f = b.take(L.NC * L.img_body + L.img_last_f);
    if (c.training) {
      L.img_d = b.take(L.NC * L.img_body + L.img_last_d);
      L.gz.resize(L.NC + 1);
      for (int i = 0; i <= L.NC; ++i) L.gz[i] = b.take(L.np * L.F);
      if (L.has_prelu) {   // unmasked gradients of every layer + partial rows: all slope gradients in one call at the end
        L.dAall.resize(L.NC + 1);
        for (int i = 0; i <= L.NC; ++i) L.dAall[i] = b.take(L.np * L.F);
        L.psm_ws = b.take((int64_t)(L.NC + 1) * (neosr_prelu_dslope_workspace_bytes(L.np, L.F) / 4) + 64);
      }
      std::vector<neosr_wgrad_desc> wd(L.NC < NEOSR_WGRAD_MAX ? L.NC : NEOSR_WGRAD_MAX);
      float* const dummy = (float*)16;   // (the size query validates the descriptors: any non-null, 16-byte aligned address)
      for (auto& w : wd) {
        w = wgrad_base(c.B, c.H, c.W);
        w.K = L.F; w.N = L.F; w.in_cs = L.F; w.g_cs = L.F;
        w.in = dummy; w.g = dummy; w.dw = dummy; w.db = dummy;
      }
      const int64_t wm = neosr_conv3x3_wgrad_multi_workspace_bytes(wd.data(), (int)wd.size());
      L.wgm_ws = b.take((wm > 0 ? wm : 0) / 4 + 64);
      if (wm <= 0) L.w4 = 0;   // (cannot size the body's weight-gradient launch: keep the on-load scheme)
    }
  }
  L.total = (b.off + 255) & ~(int64_t)255;
  return L;
}

int compact_check(const neosr_compact_cfg* c) {
  NEOSR_CHECK(c, "compact: null cfg");
  NEOSR_CHECK(c->B > 0 && c->H > 0 && c->W > 0 && c->num_in_ch > 0 && c->num_out_ch > 0 &&
                  c->num_feat > 0 && c->num_conv >= 0 && c->upscale > 0,
              "compact: bad cfg");
  NEOSR_CHECK(c->num_feat % 4 == 0 && c->num_feat <= 256, "compact: num_feat must be 4k <= 256");
  NEOSR_CHECK(c->num_in_ch == c->num_out_ch,
              "compact: `out += nearest(x)` needs num_in_ch == num_out_ch (compact_arch.py:83-84)");
  NEOSR_CHECK(c->act_type == NEOSR_ACT_PRELU || c->act_type == NEOSR_ACT_RELU ||
                  c->act_type == NEOSR_ACT_LRELU,
              "compact: bad act_type");
  return 0;
}

// parameter indices: conv i (0..NC) then last conv
inline int cp_stride(const CompactLayout& L) { return L.has_prelu ? 3 : 2; }
inline int cp_w(const CompactLayout& L, int i) { return i * cp_stride(L); }
inline int cp_last(const CompactLayout& L) { return (L.NC + 1) * cp_stride(L); }
inline const float* cp_slope(const CompactLayout& L, const float* const* P, int i) {
  return L.has_prelu ? P[cp_w(L, i) + 2] : L.slopes_const;
}


// ---- Winograd scheme (see compact_w4)
int compact_pack_w4(const CompactLayout& L, const float* const* P, int mode, void* st) {
  std::vector<neosr_pack::Image> imgs(L.NC + 1);
  memset(imgs.data(), 0, imgs.size() * sizeof(neosr_pack::Image));
  float* base = mode == NEOSR_CONV_FWD ? L.img_f : L.img_d;
  for (int i = 1; i <= L.NC; ++i) {
    neosr_pack::Image& im = imgs[i - 1];
    im.dst = base + (int64_t)(i - 1) * L.img_body;
    im.N = L.F; im.K = L.F; im.mode = mode; im.nseg = 1;
    im.seg[0].w = P[cp_w(L, i)]; im.seg[0].w_cin = L.F; im.seg[0].k_cnt = L.F;
  }
  neosr_pack::Image& il = imgs[L.NC];
  il.dst = base + (int64_t)L.NC * L.img_body;
  il.mode = mode; il.nseg = 1;
  il.seg[0].w = P[cp_last(L)]; il.seg[0].w_cin = L.F;
  if (mode == NEOSR_CONV_FWD) { il.N = L.Clast; il.K = L.F; il.seg[0].k_cnt = L.F; }
  else { il.N = L.F; il.K = L.Clast; il.seg[0].k_cnt = L.Clast; }
  return neosr_pack::launch_wino4(imgs.data(), (int)imgs.size(), st);
}

int compact_forward_w4(const neosr_compact_cfg* c, const CompactLayout& L, const float* const* P, const float* x, float* y,
                       void* st) {
  const int B = L.B, H = L.H, W = L.W, F = L.F;
  const bool train = c->training != 0;
  RUN(neosr_nchw_to_nhwc(x, L.x_nhwc, B, L.Cin, H, W, L.cin_cs, st));
  RUN(compact_pack_w4(L, P, NEOSR_CONV_FWD, st));
  auto set_act = [&](neosr_conv_desc& d, int i) {
    d.act = c->act_type;
    if (L.has_prelu) d.prelu = P[cp_w(L, i) + 2];
    else d.slope = c->act_type == NEOSR_ACT_RELU ? 0.f : 0.1f;
  };
  {  // conv 0 (3 input channels: thin kernel): a_0, and z_0 for the backward pass
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.x_nhwc; d.in_cs = L.cin_cs; d.K = L.Cin; d.w_cin = L.Cin;
    d.w = P[cp_w(L, 0)]; d.bias = P[cp_w(L, 0) + 1]; d.w_cout = F;
    d.out_cs = F; d.N = F;
    if (train) {
      d.out = L.z[0];
      RUN(neosr_conv3x3(&d, st));
    }
    d.out = L.a[0];
    set_act(d, 0);
    RUN(neosr_conv3x3(&d, st));
  }
  const int na = (int)L.a.size();
  for (int i = 1; i <= L.NC; ++i) {
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.a[(i - 1) % na]; d.in_cs = F; d.K = F; d.w_cin = F;
    d.w = P[cp_w(L, i)]; d.bias = P[cp_w(L, i) + 1]; d.w_cout = F;
    d.w_wino4 = L.img_f + (int64_t)(i - 1) * L.img_body;
    d.out = L.a[i % na]; d.out_cs = F; d.N = F;
    set_act(d, i);
    if (train) { d.out2 = L.z[i]; d.out2_cs = F; }
    RUN(neosr_conv3x3(&d, st));
  }
  {
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.a[L.NC % na]; d.in_cs = F; d.K = F; d.w_cin = F;
    d.w = P[cp_last(L)]; d.bias = P[cp_last(L) + 1]; d.w_cout = L.Clast;
    d.w_wino4 = L.img_f + (int64_t)L.NC * L.img_body;
    d.out = L.zl; d.out_cs = L.clast_cs; d.N = L.Clast;
    RUN(neosr_conv3x3(&d, st));
  }
  return neosr_pixel_shuffle_nhwc_to_nchw(L.zl, x, y, B, L.Cout, H, W, L.r, L.clast_cs, st);
}

int compact_backward_w4(const neosr_compact_cfg* c, const CompactLayout& L, const float* const* P, float* const* Gp,
                        const float* gy, void* st) {
  const int B = L.B, H = L.H, W = L.W, F = L.F;
  RUN(neosr_pixel_unshuffle_nchw_to_nhwc(gy, L.g_zl, B, L.Cout, H, W, L.r, L.clast_cs, st));
  RUN(compact_pack_w4(L, P, NEOSR_CONV_DGRAD, st));
  // gradient wrt z_j from the data gradient of the layer above: masked by PReLU'(z_j) in the epilogue, dA_j beside it
  auto set_mask = [&](neosr_conv_desc& d, int j) {
    d.out = L.gz[j]; d.out_cs = F; d.N = F;
    d.out_mask = L.z[j]; d.out_mask_cs = F;
    if (L.has_prelu) {
      d.out_mask_slopes = P[cp_w(L, j) + 2];
      d.out2 = L.dAall[j]; d.out2_cs = F;
    } else {
      d.out_mask_slope = c->act_type == NEOSR_ACT_RELU ? 0.f : 0.1f;
    }
  };
  {  // last conv: input a_NC
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.a[L.NC]; w.in_cs = F; w.K = F;
    w.g = L.g_zl; w.g_cs = L.clast_cs; w.N = L.Clast;
    w.dw = Gp[cp_last(L)]; w.db = Gp[cp_last(L) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H, W);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_zl; d.in_cs = L.clast_cs; d.K = L.Clast;
    d.w = P[cp_last(L)]; d.w_cout = L.Clast; d.w_cin = F;
    d.w_wino4 = L.img_d + (int64_t)L.NC * L.img_body;
    set_mask(d, L.NC);
    RUN(neosr_conv3x3(&d, st));
  }
  for (int i = L.NC; i >= 1; --i) {
    neosr_conv_desc d = conv_base(B, H, W);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.gz[i]; d.in_cs = F; d.K = F;
    d.w = P[cp_w(L, i)]; d.w_cout = F; d.w_cin = F;
    d.w_wino4 = L.img_d + (int64_t)(i - 1) * L.img_body;
    set_mask(d, i - 1);
    RUN(neosr_conv3x3(&d, st));
  }
  if (L.has_prelu) {   // all NC + 1 slope gradients: two launches
    std::vector<neosr_dslope_item> it(L.NC + 1);
    for (int i = 0; i <= L.NC; ++i) {
      it[i].dA = L.dAall[i]; it[i].z = L.z[i]; it[i].dslope = Gp[cp_w(L, i) + 2];
    }
    RUN(neosr_prelu_dslope_many(it.data(), L.NC + 1, L.psm_ws, L.np, F, F, F, st));
  }
  // the body's weight (+ bias) gradients: plain operands (a_{i-1}, g_z_i), up to NEOSR_WGRAD_MAX convolutions per launch
  for (int i0 = 1; i0 <= L.NC; i0 += NEOSR_WGRAD_MAX) {
    const int n = L.NC - i0 + 1 < NEOSR_WGRAD_MAX ? L.NC - i0 + 1 : NEOSR_WGRAD_MAX;
    neosr_wgrad_desc wd[NEOSR_WGRAD_MAX];
    for (int k = 0; k < n; ++k) {
      const int i = i0 + k;
      wd[k] = wgrad_base(B, H, W);
      wd[k].in = L.a[i - 1]; wd[k].in_cs = F; wd[k].K = F;
      wd[k].g = L.gz[i]; wd[k].g_cs = F; wd[k].N = F;
      wd[k].dw = Gp[cp_w(L, i)]; wd[k].db = Gp[cp_w(L, i) + 1];
    }
    RUN(neosr_conv3x3_wgrad_multi(wd, n, L.wgm_ws, st));
  }
  {  // conv 0
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.x_nhwc; w.in_cs = L.cin_cs; w.K = L.Cin;
    w.g = L.gz[0]; w.g_cs = F; w.N = F;
    w.dw = Gp[cp_w(L, 0)]; w.db = Gp[cp_w(L, 0) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
  }
  return 0;
}

}  // namespace

extern "C" int32_t neosr_compact_num_params(const neosr_compact_cfg* c) {
  return (c->num_conv + 1) * (c->act_type == NEOSR_ACT_PRELU ? 3 : 2) + 2;
}

extern "C" int64_t neosr_compact_workspace_bytes(const neosr_compact_cfg* c) {
  if (compact_check(c)) return -1;
  return compact_layout(*c, nullptr).total;
}

extern "C" int neosr_compact_forward(const neosr_compact_cfg* c, const float* const* P,
                                     const float* x, float* y, void* ws, void* st) {
  RUN(compact_check(c));
  NEOSR_CHECK(P && x && y && ws, "compact_forward: null pointer");
  const CompactLayout L = compact_layout(*c, ws);
  if (L.w4) return compact_forward_w4(c, L, P, x, y, st);
  const int B = L.B, H = L.H, W = L.W, F = L.F;
  if (!L.has_prelu)
    RUN(neosr_fill(L.slopes_const, F, c->act_type == NEOSR_ACT_RELU ? 0.f : 0.1f, st));
  RUN(neosr_nchw_to_nhwc(x, L.x_nhwc, B, L.Cin, H, W, L.cin_cs, st));
  for (int i = 0; i <= L.NC; ++i) {
    neosr_conv_desc d = conv_base(B, H, W);
    if (i == 0) {
      d.in = L.x_nhwc; d.in_cs = L.cin_cs; d.K = L.Cin; d.w_cin = L.Cin;
    } else {
      d.in = L.z[(i - 1) % L.nz]; d.in_cs = F; d.K = F; d.w_cin = F;
      d.in_prelu = cp_slope(L, P, i - 1);
    }
    d.w = P[cp_w(L, i)]; d.bias = P[cp_w(L, i) + 1]; d.w_cout = F;
    d.out = L.z[i % L.nz]; d.out_cs = F; d.N = F;
    RUN(neosr_conv3x3(&d, st));
  }
  {
    neosr_conv_desc d = conv_base(B, H, W);
    d.in = L.z[L.NC % L.nz]; d.in_cs = F; d.K = F; d.w_cin = F;
    d.in_prelu = cp_slope(L, P, L.NC);
    d.w = P[cp_last(L)]; d.bias = P[cp_last(L) + 1]; d.w_cout = L.Clast;
    d.out = L.zl; d.out_cs = L.clast_cs; d.N = L.Clast;
    RUN(neosr_conv3x3(&d, st));
  }
  RUN(neosr_pixel_shuffle_nhwc_to_nchw(L.zl, x, y, B, L.Cout, H, W, L.r, L.clast_cs, st));
  return 0;
}

extern "C" int neosr_compact_backward(const neosr_compact_cfg* c, const float* const* P,
                                      float* const* Gp, const float* gy, float* gx, void* ws,
                                      void* st) {
  RUN(compact_check(c));
  NEOSR_CHECK(P && Gp && gy && ws, "compact_backward: null pointer");
  NEOSR_CHECK(c->training, "compact_backward: cfg.training must be set");
  NEOSR_CHECK(gx == nullptr, "compact_backward: input gradient is not implemented");
  const CompactLayout L = compact_layout(*c, ws);
  if (L.w4) return compact_backward_w4(c, L, P, Gp, gy, st);
  const int B = L.B, H = L.H, W = L.W, F = L.F;
  RUN(neosr_pixel_unshuffle_nchw_to_nhwc(gy, L.g_zl, B, L.Cout, H, W, L.r, L.clast_cs, st));
  {  // last conv: input prelu(z_NC)
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    w.in = L.z[L.NC]; w.in_cs = F; w.K = F; w.in_prelu = cp_slope(L, P, L.NC);
    w.g = L.g_zl; w.g_cs = L.clast_cs; w.N = L.Clast;
    w.dw = Gp[cp_last(L)]; w.db = Gp[cp_last(L) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    neosr_conv_desc d = conv_base(B, H, W);
    d.mode = NEOSR_CONV_DGRAD;
    d.in = L.g_zl; d.in_cs = L.clast_cs; d.K = L.Clast;
    d.w = P[cp_last(L)]; d.w_cout = L.Clast; d.w_cin = F;
    d.out = L.dA[L.NC & 1]; d.out_cs = F; d.N = F;
    RUN(neosr_conv3x3(&d, st));
  }
  for (int i = L.NC; i >= 0; --i) {
    const float* dAi = L.dA[i & 1];  // gradient wrt a_i = act(z_i)
    const float* si = cp_slope(L, P, i);
    if (L.has_prelu)
      RUN(neosr_prelu_dslope(dAi, L.z[i], Gp[cp_w(L, i) + 2], L.ps_ws, L.np, F, F, F, 0, st));
    neosr_wgrad_desc w = wgrad_base(B, H, W);
    if (i == 0) {
      w.in = L.x_nhwc; w.in_cs = L.cin_cs; w.K = L.Cin;
    } else {
      w.in = L.z[i - 1]; w.in_cs = F; w.K = F; w.in_prelu = cp_slope(L, P, i - 1);
    }
    w.g = dAi; w.g_cs = F; w.N = F; w.g_mask = L.z[i]; w.mask_cs = F; w.mask_slopes = si;
    w.dw = Gp[cp_w(L, i)]; w.db = Gp[cp_w(L, i) + 1]; w.workspace = L.wg_ws;
    RUN(neosr_conv3x3_wgrad(&w, st));
    if (i > 0) {
      neosr_conv_desc d = conv_base(B, H, W);
      d.mode = NEOSR_CONV_DGRAD;
      d.in = dAi; d.in_cs = F; d.K = F; d.in_mask = L.z[i]; d.mask_cs = F; d.mask_slopes = si;
      d.w = P[cp_w(L, i)]; d.w_cout = F; d.w_cin = F;
      d.out = L.dA[(i - 1) & 1]; d.out_cs = F; d.N = F;
      RUN(neosr_conv3x3(&d, st));
    }
  }
  return 0;
}

// 1 = the RRDB trunk runs on the caller's stream only, n = 2 (default) .. 4 = the batch is cut into n groups of samples
// that run as independent launch chains.  Returns the previous setting.  Env NEOSR_AMD_STREAMS=n selects it at start-up.
// Behind a chain launch (neosr_set_conv_chain) the fifteen weight gradients of an RRDB run as ONE launch (1, default) or
// as one launch per RDB (0).  Same products, another split of the pixel range: results differ by summation order
// (~1e-7 relative).  Returns the previous setting.
extern "C" int neosr_set_wgrad_rrdb(int on) {
  const int prev = g_wgrad_rrdb < 0 ? 1 : g_wgrad_rrdb;
  g_wgrad_rrdb = on ? 1 : 0;
  return prev;
}

extern "C" int neosr_set_num_streams(int n) {
  aux_get(0);  // resolve the default
  const int prev = g_num_streams;
  g_num_streams = n < 1 ? 1 : (n > MAX_CHAINS ? MAX_CHAINS : n);
  return prev;
}
