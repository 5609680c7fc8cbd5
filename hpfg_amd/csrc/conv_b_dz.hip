#include "conv_bf16_kernel.h"

int hpfg_conv16_launch_dz(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  if (a.taps != 9) {
    hpfg_set_error("conv_fwd(bf16x3): 1x1 convolution with a dz loader is not instantiated");
    return -1;
  }
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_DZ, 9>(a, st, rows_only);
}
