// Host side of conv_thin_kernel.h: the thin forward 3x3 layers of the U-Net at 16-pixel-aligned resolutions.
#include "conv_thin_kernel.h"

namespace {

using namespace hpfg_thin;

template <int CI, int CO, int AK, int NW, int WGS>
int launch(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  const int grid = thin_grid<CI, CO, AK, NW, WGS>(a);
  if (rows_only) {
    *rows_only = grid;
    return 0;
  }
  if (a.stat_acc) HPFG_ACC_CHECK(grid, a.stat_shards, "conv_fwd(thin)");
  hipLaunchKernelGGL((conv_thin_kernel<CI, CO, AK, NW, WGS>), dim3(grid), dim3(64 * NW), 0, st, a, a.W / T, a.H / T);
  return hpfg_launch_status("conv_thin_kernel");
}

bool enabled() { return hpfg_opt(HPFG_OPT_CONV_THIN) != 0; }      // (tests switch it inside one process: hpfg_set_option)

}  // namespace

// HPFG_THIN_NONE: this layer is not one of the kernel's shapes (the caller falls through to conv_bf16x3_kernel)
int hpfg_conv_thin_try(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  if (!enabled() || a.taps != 9 || a.H % T || a.W % T || a.bwd_stats || a.out_split || a.stage_out || (a.math & ~0xff)) return HPFG_THIN_NONE;
  if (a.Cout % 4 || a.out_pstride % 4 || a.CoutPad > 64) return HPFG_THIN_NONE;
  const int cin = a.a0.C + a.a1.C, ci = cin / 16, co = a.CoutPad / 16;
  if (cin % 16 || a.a0.pstride % 4) return HPFG_THIN_NONE;
  const int ak = hpfg_kind_of(a.a0, a.a1);
  if (ak == HPFG_KIND_CAT && (a.a0.C != a.a1.C || a.a0.C % 16 || a.a1.pstride % 4)) return HPFG_THIN_NONE;
#define HPFG_THIN_CASE(CI, CO, AK, NW, WGS) \
  if (ci == CI && co == CO && ak == AK) return launch<CI, CO, AK, NW, WGS>(a, st, rows_only);
  if (ak == HPFG_KIND_DZ && a.a0.aux_pstride % 4) return HPFG_THIN_NONE;
  HPFG_THIN_CASE(2, 4, HPFG_KIND_DZ, 8, 1)         // dgrad of up3.c1 (dZ of 32 channels -> the 64-channel concat gradient)
  if (co > 2) return HPFG_THIN_NONE;
  HPFG_THIN_CASE(1, 1, HPFG_KIND_BNACT, 4, 3)      // in_conv.c2, up4.c2, out_conv
  // (32 -> 32 @112^2 is not here: with its weight fragments in LDS only one 8-wave workgroup fits a CU: 29.9 us against 28.3 us)
  HPFG_THIN_CASE(1, 2, HPFG_KIND_POOL, 8, 1)       // down1.c1
  HPFG_THIN_CASE(2, 1, HPFG_KIND_CAT, 4, 2)        // up4.c1
#undef HPFG_THIN_CASE
  return HPFG_THIN_NONE;
}
