#include "conv_kernel.h"

int hpfg_conv_launch_bnact(const HpfgConvArgs& a, hipStream_t st) {
  if (a.taps == 1) return hpfg_conv::conv_dispatch_kind<HPFG_KIND_BNACT, 1>(a, st);
  return hpfg_conv::conv_dispatch_kind<HPFG_KIND_BNACT, 9>(a, st);
}
