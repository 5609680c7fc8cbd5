#include "wgrad_bf16_kernel.h"

// 1x1 convolutions (decoder conv1x1 of every up block, out_conv): dZ = the plain gradient tensor, A = the activated block output
int hpfg_wgrad16_launch_1x1(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg16;
  switch (akind) {
    case HPFG_KIND_PLAIN: return launch_wgrad16_taps<HPFG_KIND_PLAIN, HPFG_KIND_PLAIN, 1>(a, st);
    case HPFG_KIND_BNACT: return launch_wgrad16_taps<HPFG_KIND_BNACT, HPFG_KIND_PLAIN, 1>(a, st);
    default: break;
  }
  return 1;   // not handled here: the caller falls back to the exact-fp32 kernel
}
