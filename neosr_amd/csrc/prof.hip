// prof.hip — opt-in per-kernel-class timing with HIP events on the launch stream.
// bench.py enables it for a dedicated pass (never during the pass that produces `value`) to get
// the average duration and algorithmic FLOPs of the conv kernels for the roofline line.
#include "common.h"
#include "prof.h"
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int cls; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
double g_flops[NEOSR_PROF_NCLASS];
double g_bytes[NEOSR_PROF_NCLASS];
long long g_launch[NEOSR_PROF_NCLASS];
double g_exec[NEOSR_PROF_NCLASS];            // multiplications the launches really executed (Winograd forms: fewer)
long long g_algo[NEOSR_PROF_NCLASS][3];      // launches by algorithm: direct, Winograd F(2x2,3x3), Winograd F(4x4,3x3)
double g_last_flops = 0.0;
int g_last_cls = 0;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

bool neosr_prof_on() { return g_on; }

void neosr_prof_begin(int cls, void* stream, double flops, double bytes) {
  Rec r;
  r.a = get_event();
  r.b = get_event();
  r.cls = cls;
  (void)hipEventRecord(r.a, (hipStream_t)stream);
  g_recs.push_back(r);
  g_flops[cls] += flops;
  g_bytes[cls] += bytes;
  g_launch[cls] += 1;
  g_exec[cls] += flops;
  g_algo[cls][0] += 1;
  g_last_flops = flops;
  g_last_cls = cls;
}

// the launch just begun runs in a Winograd form: algo 1 = F(2x2,3x3) (16 of the direct form's 36 multiplications),
// 2 = F(4x4,3x3) (36 of 144)
void neosr_prof_algo(int algo) {
  if (algo != 1 && algo != 2) return;
  g_exec[g_last_cls] -= g_last_flops * (algo == 1 ? 20.0 / 36.0 : 0.75);
  g_algo[g_last_cls][0] -= 1;
  g_algo[g_last_cls][algo] += 1;
}

// the launch just begun is a chain of n layers: they count as n launches of the class (per-layer averages stay comparable)
long long g_chain_launch[NEOSR_PROF_NCLASS], g_chain_layers[NEOSR_PROF_NCLASS];
void neosr_prof_layers(int n) {
  if (n > 1) g_launch[g_last_cls] += n - 1, g_algo[g_last_cls][2] += n - 1;
  g_chain_launch[g_last_cls] += 1;
  g_chain_layers[g_last_cls] += n;
}

void neosr_prof_end(void* stream) { (void)hipEventRecord(g_recs.back().b, (hipStream_t)stream); }

extern "C" int neosr_prof_enable(int on) {
  g_on = on != 0;
  if (g_on) {
    for (int i = 0; i < NEOSR_PROF_NCLASS; ++i) {
      g_flops[i] = 0; g_bytes[i] = 0; g_launch[i] = 0; g_exec[i] = 0;
      g_algo[i][0] = g_algo[i][1] = g_algo[i][2] = 0;
      g_chain_launch[i] = g_chain_layers[i] = 0;
    }
  }
  return 0;
}

extern "C" int neosr_prof_num_classes() { return NEOSR_PROF_NCLASS; }

// ms[c], launches[c], flops[c], bytes[c] for c < NEOSR_PROF_NCLASS; synchronises the device.
extern "C" int neosr_prof_collect(double* ms, long long* launches, double* flops, double* bytes) {
  NEOSR_HIP(hipDeviceSynchronize());
  for (int i = 0; i < NEOSR_PROF_NCLASS; ++i) { ms[i] = 0; launches[i] = g_launch[i]; flops[i] = g_flops[i]; bytes[i] = g_bytes[i]; }
  for (Rec& r : g_recs) {
    float t = 0.f;
    NEOSR_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += t;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}

// executed[c] = FLOPs of the multiplications the class's launches really ran (flops[c] of neosr_prof_collect counts the
// DIRECT form, SURVEY §8d's algorithmic figure); by_algo[3 c + a] = launches in the direct / F(2x2,3x3) / F(4x4,3x3) form.
// Call before neosr_prof_collect (which recycles the events, not these sums).
// chain_launches[c] / chain_layers[c]: launches of conv3x3_wino4_chain_kernel in class c and the layers they ran (each layer
// counts as one launch in neosr_prof_collect's launches[c], so that per-layer averages stay comparable with the
// one-layer kernels; kernel launches = launches[c] - chain_layers[c] + chain_launches[c]).  Call before neosr_prof_collect.
extern "C" int neosr_prof_collect_chain(long long* chain_launches, long long* chain_layers) {
  for (int i = 0; i < NEOSR_PROF_NCLASS; ++i) {
    chain_launches[i] = g_chain_launch[i];
    chain_layers[i] = g_chain_layers[i];
  }
  return 0;
}

extern "C" int neosr_prof_collect_exec(double* executed, long long* by_algo) {
  for (int i = 0; i < NEOSR_PROF_NCLASS; ++i) {
    executed[i] = g_exec[i];
    for (int a = 0; a < 3; ++a) by_algo[3 * i + a] = g_algo[i][a];
  }
  return 0;
}
