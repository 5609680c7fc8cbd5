#include "wgrad_kernel.h"

int hpfg_wgrad_launch_plain(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg;
  if (a.taps == 1) {
    if (akind == HPFG_KIND_BNACT) return launch_wgrad<1, HPFG_KIND_BNACT, HPFG_KIND_PLAIN>(a, st);
    if (akind == HPFG_KIND_PLAIN) return launch_wgrad<1, HPFG_KIND_PLAIN, HPFG_KIND_PLAIN>(a, st);
  } else {
    if (akind == HPFG_KIND_BNACT) return launch_wgrad<9, HPFG_KIND_BNACT, HPFG_KIND_PLAIN>(a, st);
    if (akind == HPFG_KIND_PLAIN) return launch_wgrad<9, HPFG_KIND_PLAIN, HPFG_KIND_PLAIN>(a, st);
  }
  hpfg_set_error("wgrad: unsupported input source kind %d for a plain gradient source", akind);
  return -1;
}
