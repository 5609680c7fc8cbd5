// Multi-network launches (hpfg_conv_fwd_multi) of the BNACT-loader layers: 3x3 and 1x1.
#include "conv_bf16_kernel.h"

int hpfg_conv16_multi_bnact(const HpfgConvArgs& a, int nnets, hipStream_t st) {
  if (a.taps == 1) return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_BNACT, 1, true>(a, st, nullptr, nnets);
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_BNACT, 9, true>(a, st, nullptr, nnets);
}
