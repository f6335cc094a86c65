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

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}
}  // namespace

bool neosr_prof_on() { return g_on; }

void neosr_prof_begin(int cls, void* stream, double flops, double bytes) {
  Rec r;
  r.a = get_event();
  r.b = get_event();
  r.cls = cls;
  hipEventRecord(r.a, (hipStream_t)stream);
  g_recs.push_back(r);
  g_flops[cls] += flops;
  g_bytes[cls] += bytes;
  g_launch[cls] += 1;
}

void neosr_prof_end(void* stream) { hipEventRecord(g_recs.back().b, (hipStream_t)stream); }

extern "C" int neosr_prof_enable(int on) {
  g_on = on != 0;
  if (g_on) {
    for (int i = 0; i < NEOSR_PROF_NCLASS; ++i) { g_flops[i] = 0; g_bytes[i] = 0; g_launch[i] = 0; }
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
