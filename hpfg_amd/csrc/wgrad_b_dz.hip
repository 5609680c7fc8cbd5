#include "wgrad_bf16_kernel.h"

int hpfg_wgrad16_launch_dz(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg16;
  switch (akind) {
    case HPFG_KIND_PLAIN: return launch_wgrad16<HPFG_KIND_PLAIN, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_BNACT: return launch_wgrad16<HPFG_KIND_BNACT, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_POOL: return launch_wgrad16<HPFG_KIND_POOL, HPFG_KIND_DZ>(a, st);
    case HPFG_KIND_CAT: return launch_wgrad16<HPFG_KIND_CAT, HPFG_KIND_DZ>(a, st);
    default: break;
  }
  hpfg_set_error("wgrad(bf16x3): unsupported input source kind %d", akind);
  return -1;
}
