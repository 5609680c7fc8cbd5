#include "wgrad_kernel.h"

int hpfg_wgrad_launch_dz(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg;
  if (a.taps != 9) {
    hpfg_set_error("wgrad: a BatchNorm-backward gradient source is only instantiated for 3x3 convolutions");
    return -1;
  }
  switch (akind) {
    case HPFG_KIND_PLAIN: return launch_wgrad<9, HPFG_KIND_PLAIN, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_BNACT: return launch_wgrad<9, HPFG_KIND_BNACT, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_POOL: return launch_wgrad<9, HPFG_KIND_POOL, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_CAT: return launch_wgrad<9, HPFG_KIND_CAT, HPFG_KIND_DZ>(a, st);
    default: break;
  }
  hpfg_set_error("wgrad: unsupported input source kind %d", akind);
  return -1;
}
