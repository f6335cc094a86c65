// attn_wave.h — wave-per-(window, head) 8x8 window attention kernels (attn_wave.hip), launched from attn.hip.
#pragma once
#include "../../include/neosr_amd.h"

namespace neosr_wattn {
// head_dim <= 30: the 16 k-slots of a lane hold half a head row plus (backward) one slot for the folded lse / delta
bool wave_ok(const neosr_wattn_desc& d);
void launch_fwd(const neosr_wattn_desc& d, void* stream);
void launch_bwd(const neosr_wattn_desc& d, void* stream);  // needs d.out (the forward output) besides d.dout
}  // namespace neosr_wattn
