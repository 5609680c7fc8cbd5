// Multi-network launches (hpfg_conv_fwd_multi) of the max-pool-loader layers (first conv of a DownBlock, model/unet.py:37-38).
#include "conv_bf16_kernel.h"

int hpfg_conv16_multi_pool(const HpfgConvArgs& a, int nnets, hipStream_t st) {
  if (a.taps != 9) {
    hpfg_set_error("conv_fwd_multi: 1x1 convolution with a pool loader is not instantiated");
    return -1;
  }
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_POOL, 9, true>(a, st, nullptr, nnets);
}
