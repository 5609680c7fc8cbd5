#include "wgrad_bf16_kernel.h"

int hpfg_wgrad16_launch_plain(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg16;
  if (akind == HPFG_KIND_BNACT) return launch_wgrad16<HPFG_KIND_BNACT, HPFG_KIND_PLAIN>(a, st);
  if (akind == HPFG_KIND_PLAIN) return launch_wgrad16<HPFG_KIND_PLAIN, HPFG_KIND_PLAIN>(a, st);
  if (akind == HPFG_KIND_POOL) return launch_wgrad16<HPFG_KIND_POOL, HPFG_KIND_PLAIN>(a, st);      // (dZ stored by the dgrad: HpfgConvArgs.dz_out)
  if (akind == HPFG_KIND_CAT) return launch_wgrad16<HPFG_KIND_CAT, HPFG_KIND_PLAIN>(a, st);
  hpfg_set_error("wgrad(bf16x3): unsupported input source kind %d for a plain gradient source", akind);
  return -1;
}
