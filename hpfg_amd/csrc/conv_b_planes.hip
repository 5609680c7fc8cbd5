#include "conv_bf16_kernel.h"

// PLANES source (a materialised split-bf16 activation or dZ, include/hpfg_hip.h: HPFG_ACT_PLANES): forward and dgrad of the channel-rich layers
int hpfg_conv16_launch_planes(const HpfgConvArgs& a, hipStream_t st, int* rows_only, const HpfgConvArgs* b) {
  if (a.taps == 9) return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_PLANES, 9>(a, st, rows_only, b);
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_PLANES, 1>(a, st, rows_only, b);
}
