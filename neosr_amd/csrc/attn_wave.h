// attn_wave.h — wave-per-(window, head, query tile) 8x8 window attention forward (attn_wave.hip), launched from attn.hip.
#pragma once
#include "../../include/neosr_amd.h"

namespace neosr_wattn {
// head_dim <= 30: the 16 k-slots of a lane hold half a head row
bool wave_ok(const neosr_wattn_desc& d);
void launch_fwd(const neosr_wattn_desc& d, void* stream);
// HAT 16 x 16 self-attention forward in the same style (ks == ws == 16, head_dim <= 30)
bool wave16_ok(const neosr_fattn_desc& d);
void launch16_fwd(const neosr_fattn_desc& d, void* stream);
}  // namespace neosr_wattn
