#include "conv_kernel.h"

int hpfg_conv_launch_pool(const HpfgConvArgs& a, hipStream_t st) {
  if (a.taps != 9) {
    hpfg_set_error("conv_fwd: 1x1 convolution with a pool loader is not instantiated");
    return -1;
  }
  return hpfg_conv::conv_dispatch_kind<HPFG_KIND_POOL, 9>(a, st);
}
