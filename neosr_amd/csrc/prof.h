// prof.h — kernel classes for the opt-in HIP-event profiler (see prof.hip)
#pragma once
enum : int {
  NEOSR_PROF_CONV_FWD = 0,
  NEOSR_PROF_CONV_DGRAD = 1,
  NEOSR_PROF_CONV_WGRAD = 2,
  NEOSR_PROF_WGRAD_REDUCE = 3,
  NEOSR_PROF_NCLASS = 4
};
bool neosr_prof_on();
void neosr_prof_begin(int cls, void* stream, double flops, double bytes);
void neosr_prof_end(void* stream);
