#include "conv_bf16_kernel.h"

int hpfg_conv16_launch_bnact(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  if (a.taps == 1) return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_BNACT, 1>(a, st, rows_only);
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_BNACT, 9>(a, st, rows_only);
}
