// conv_wino4_chain.h — layer table and launcher of conv3x3_wino4_chain_kernel (conv_wino4_chain.hip).  Internal to
// libneosr_amd: nets.hip hands the launcher the SAME neosr_conv_desc records it would pass to neosr_conv3x3 one by one.
#pragma once
#include <cstdint>
#include "../../include/neosr_amd.h"

namespace neosr_conv {

struct W4Layer {  // device-side record of one layer, 128 bytes
  const float* in;        // (B, H, W, in_cs): the first K channels are reduced
  const float* u;         // Winograd F(4x4,3x3) weight image (neosr_conv3x3_pack_wino4)
  const float* bias;
  const float* res1;
  const float* res2;
  const float* out_mask;
  float* out;
  int32_t K, N, out_cs, res1_cs, res1_nch, res2_cs, res2_nch, out_mask_cs, act;
  int32_t n64;            // 1: the 64-output-channel wave mapping of conv3x3_wino4_kernel<true>
  int32_t dep;            // first 32-channel chunk of `in` that the PREVIOUS layer of the table wrote (-1: none)
  int32_t pad0;
  float slope, alpha, alpha2, out_mask_slope;
  int32_t pad1[2];
};
static_assert(sizeof(W4Layer) == 128, "W4Layer: 32 dwords (the kernel reads it as eight 16-byte LDS reads)");

constexpr int W4_MAX_LAYERS = 15;   // (15 x 128 bytes + the header below stay under the 4 KB kernel-argument segment)

struct W4ChainArgs {
  W4Layer layers[W4_MAX_LAYERS];   // the layer table travels in the kernel arguments: constant address space, no upload
  unsigned* flags;    // one word per pixel tile: layers finished by the tile's workgroup; zeroed before the launch
  unsigned* status;   // [0] != 0: a flag wait ran into its spin bound (value = 1 + the epoch it waited for); sticky
  int32_t nlayers;
  int32_t B, H, W, in_cs;
  int32_t tiles_x, tiles_y, tx_shift, ty_shift, xcd;
  int32_t sync;       // 0: flag waits skipped (timing experiments only: results are then racy)
  int32_t indep;      // 1: the layers are independent of each other (sample strips of one convolution): no drain, no flags
  unsigned long long* timeline;  // debug only (NEOSR_TIMELINE builds): [wave 12][layer 16][mark 16] clocks of workgroup 0
};

int chain_max_tiles();  // pixel tiles (= workgroups) one chain launch may have: the CU count of the device
bool chain_enabled();  // NEOSR_AMD_CHAIN=0 / neosr_set_conv_chain(0): one launch per layer
// `dep[i]`: see W4Layer::dep; dep == nullptr: independent layers, `flags` may be nullptr (dep[0] is ignored: layer 0 only reads what earlier launches wrote).  All layers share B, H,
// W and the input channel stride; n <= W4_MAX_LAYERS.  Returns 0 = launched, 1 = error (neosr_last_error), -1 = the table does not qualify
// (geometry, options, more tiles than CUs, chain switched off): the caller launches the layers one by one.
int launch_wino4_chain(const neosr_conv_desc* d, const int* dep, int n, unsigned* flags, void* stream, bool profile = true);
// one convolution with more pixel tiles than CUs as chain launches over strips of samples (dep == nullptr form of the above)
int launch_wino4_strips(const neosr_conv_desc& d, void* stream);

}  // namespace neosr_conv
