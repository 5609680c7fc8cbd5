#include "wgrad_bf16_kernel.h"

// Weight gradient with at least one operand read from a side tensor stored already split (HPFG_ACT_SPLIT16, HpfgConvArgs.stage_out): that
// operand's loader is a copy into LDS.  <SPLIT, SPLIT> is the engine's default for the channel-rich 56 / 28 / 14-pixel layers; the mixed
// forms serve runs that switch one of the two side tensors off (tests, A/B tools).
int hpfg_wgrad16_launch_split(const HpfgWgradArgs& a, int akind, hipStream_t st) {
  using namespace hpfg_wg16;
  for (const HpfgAct* s : {&a.a0, &a.g}) {
    if (s->mode != HPFG_ACT_SPLIT16) continue;
    if (!s->z || s->C % 8 || s->pstride % 8) {
      hpfg_set_error("wgrad(bf16x3): a SPLIT16 source needs channels / pixel stride multiples of 8 (C=%d, pstride=%d)", s->C, s->pstride);
      return -1;
    }
  }
  if (a.g.mode == HPFG_ACT_SPLIT16) {
    switch (akind) {
      case HPFG_KIND_SPLIT: return launch_wgrad16<HPFG_KIND_SPLIT, HPFG_KIND_SPLIT>(a, st);
      case HPFG_KIND_PLAIN: return launch_wgrad16<HPFG_KIND_PLAIN, HPFG_KIND_SPLIT>(a, st);
      case HPFG_KIND_BNACT: return launch_wgrad16<HPFG_KIND_BNACT, HPFG_KIND_SPLIT>(a, st);
      case HPFG_KIND_POOL: return launch_wgrad16<HPFG_KIND_POOL, HPFG_KIND_SPLIT>(a, st);
      case HPFG_KIND_CAT: return launch_wgrad16<HPFG_KIND_CAT, HPFG_KIND_SPLIT>(a, st);
      default: break;
    }
  } else if (akind == HPFG_KIND_SPLIT) {
    if (a.g.mode == HPFG_ACT_DZ) return launch_wgrad16<HPFG_KIND_SPLIT, HPFG_KIND_DZ>(a, st);
    if (a.g.mode == HPFG_ACT_PLAIN) return launch_wgrad16<HPFG_KIND_SPLIT, HPFG_KIND_PLAIN>(a, st);
  }
  hpfg_set_error("wgrad(bf16x3): unsupported combination of sources with a SPLIT16 operand (input kind %d, gradient mode %d)", akind, a.g.mode);
  return -1;
}
