#include "conv_bf16_kernel.h"

// HPFG_ACT_UPBWD source: the dgrad of a decoder block's 1x1 conv with the upsample backward made on load (conv1x1_bf16x3_kernel, UPB)
int hpfg_conv16_launch_upb(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  if (a.taps != 1) {
    hpfg_set_error("conv_fwd(bf16x3): an UPBWD source is instantiated for the 1x1 convolution only");
    return -1;
  }
  return hpfg_conv16::conv_dispatch_kind<HPFG_KIND_UPB, 1>(a, st, rows_only);
}
