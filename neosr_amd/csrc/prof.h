// prof.h — kernel classes for the opt-in HIP-event profiler (see prof.hip)
#pragma once
enum : int {
  NEOSR_PROF_CONV_FWD = 0,        // conv3x3_glds_kernel, forward launches (packed weights)
  NEOSR_PROF_CONV_DGRAD = 1,      // conv3x3_glds_kernel, backward-data launches
  NEOSR_PROF_CONV_WGRAD = 2,      // conv3x3_wgrad_multi_kernel (+ thin weight gradients)
  NEOSR_PROF_WGRAD_REDUCE = 3,
  NEOSR_PROF_CONV_FWD_OTHER = 4,  // staged / thin kernels, forward
  NEOSR_PROF_CONV_DGRAD_OTHER = 5,
  NEOSR_PROF_GEMM_NT = 6,         // nn.Linear forward (gemm_nt_glds_kernel / gemm_mfma_kernel<0>)
  NEOSR_PROF_GEMM_NN = 7,         // nn.Linear backward-data
  NEOSR_PROF_GEMM_TN = 8,         // nn.Linear backward-weight (split-K kernel only, not its column-sum pass)
  NEOSR_PROF_ATTN_FWD = 9,        // window_attention_fwd_kernel / flash_wattn_fwd_kernel
  NEOSR_PROF_ATTN_BWD = 10,       // window_attention_bwd_kernel / flash_wattn_bwd_dq + bwd_dkv kernels
  NEOSR_PROF_NCLASS = 11
};
bool neosr_prof_on();
void neosr_prof_begin(int cls, void* stream, double flops, double bytes);
void neosr_prof_algo(int algo);  // the launch just begun: 1 = Winograd F(2x2,3x3), 2 = F(4x4,3x3) (executed FLOPs)
void neosr_prof_layers(int n);  // the launch just begun runs n layers (conv3x3_wino4_chain_kernel)
void neosr_prof_end(void* stream);
