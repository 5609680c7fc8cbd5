// Multi-network launches (hpfg_conv_fwd_multi) of the concat-loader layers (first conv of an UpBlock, model/unet.py:56-58).
#include "conv_bf16_kernel.h"

int hpfg_conv16_multi_cat(const HpfgConvArgs& a, int nnets, hipStream_t st) {
  if (a.taps != 9) {
    hpfg_set_error("conv_fwd_multi: 1x1 convolution with a cat loader is not instantiated");
    return -1;
  }
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_CAT, 9, true>(a, st, nullptr, nnets);
}
