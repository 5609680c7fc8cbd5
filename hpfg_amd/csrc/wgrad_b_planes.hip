#include "wgrad_bf16_kernel.h"

// both operands materialised in split-bf16 form (HPFG_ACT_PLANES): the channel-rich layers, whose (input tile, dZ tile) pair is staged once per
// 32 x 32 channel block -- Cout/32 + Cin/32 times per tile
int hpfg_wgrad16_launch_planes(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg16;
  if (akind == HPFG_KIND_PLANES) return launch_wgrad16<HPFG_KIND_PLANES, HPFG_KIND_PLANES>(a, st);
  hpfg_set_error("wgrad(bf16x3): a PLANES gradient source needs a PLANES input source (got kind %d)", akind);
  return -1;
}
