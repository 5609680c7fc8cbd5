// hpfg_fused_bwd: host side of the fused dgrad + wgrad kernel of the thin 3x3 layers (fused_bwd_kernel.h).
#include "fused_bwd_kernel.h"

int hpfg_first_wgrad_grid(int N, int H, int W);      // first_wgrad.hip
int hpfg_first_wgrad_launch(const HpfgAct& g, const HpfgAct& x, float* slab, int Cin, int N, int H, int W, hipStream_t st);

namespace {

using namespace hpfg_fused;

template <int CI, int CO, int AK, int GK, int NW, int WGS, int PFA, int PFPOS, int BMAX = 48, bool NODG = false>
int launch(const HpfgFusedBwdArgs& a, hipStream_t st, bool grid_only) {
  const int grid = fused_grid<CI, CO, AK, GK, NW, WGS, BMAX>(a);
  if (grid_only) return grid;
  const int tx = a.d.W / T, ty = a.d.H / T;
  if (a.d.stat_acc) HPFG_ACC_CHECK(grid, a.d.stat_shards, "fused_bwd");
  if constexpr (NODG) {
    hipLaunchKernelGGL((fused_bwd_kernel<CI, CO, AK, GK, false, NW, WGS, PFA, PFPOS, BMAX, true>), dim3(grid), dim3(64 * NW), 0, st, a, tx, ty);
  } else if (a.d.bwd_stats) {
    if constexpr (AK == HPFG_KIND_BNACT)
      hipLaunchKernelGGL((fused_bwd_kernel<CI, CO, AK, GK, true, NW, WGS, PFA, PFPOS, BMAX>), dim3(grid), dim3(64 * NW), 0, st, a, tx, ty);
    else
      return -3;
  } else {
    hipLaunchKernelGGL((fused_bwd_kernel<CI, CO, AK, GK, false, NW, WGS, PFA, PFPOS, BMAX>), dim3(grid), dim3(64 * NW), 0, st, a, tx, ty);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the layer shapes of the U-Net at 16-pixel-aligned resolutions: (CinPad/16, CoutPad/16, input kind, dZ kind, waves per workgroup, workgroups per CU,
// input chunks prefetched a tile ahead, position of that prefetch)
int dispatch(const HpfgFusedBwdArgs& a, hipStream_t st, bool grid_only) {
  const int ci = a.CinPad / 16, co = a.CoutPad / 16;
  const int ak = hpfg_kind_of(a.xa0, a.xa1), gk = hpfg_kind_of(a.d.a0, a.d.a1);
#define HPFG_FUSED_CASE(CI, CO, AK, GK, NW, WGS, PFA, PFPOS) \
  if (ci == CI && co == CO && ak == AK && gk == GK) return launch<CI, CO, AK, GK, NW, WGS, PFA, PFPOS>(a, st, grid_only);
  if (!a.d.out) {      // weight gradient only: the first layer (network input, <= 4 channels, read through its strides)
    if (ci == 1 && co == 1 && ak == HPFG_KIND_PLAIN && gk == HPFG_KIND_DZ) {
      // 1-channel inputs of 64 .. 512 pixels a row: the streaming kernel (first_wgrad.hip); option HPFG_OPT_FIRST_WGRAD = 0 keeps the tile kernel
      // (tests, A/B runs).  RGB inputs (27 taps: 108 accumulators) measured slower there than on the tile kernel and stay on it.
      if (a.Cin == 1 && a.d.W >= 64 && a.d.W <= 512 && a.Cout == 16 && a.xa0.mode == HPFG_ACT_STRIDED && a.d.a0.pstride % 4 == 0 && a.d.a0.aux_pstride % 4 == 0 &&
          hpfg_opt(HPFG_OPT_FIRST_WGRAD) != 0) {
        if (grid_only) return hpfg_first_wgrad_grid(a.d.N, a.d.H, a.d.W);
        return hpfg_first_wgrad_launch(a.d.a0, a.xa0, a.slab, a.Cin, a.d.N, a.d.H, a.d.W, st);
      }
      return launch<1, 1, HPFG_KIND_PLAIN, HPFG_KIND_DZ, 4, 2, 1, 1, 48, true>(a, st, grid_only);
    }
    return grid_only ? 0 : -2;
  }
  HPFG_FUSED_CASE(1, 1, HPFG_KIND_BNACT, HPFG_KIND_DZ, 4, 2, 1, 1)      // in_conv.c2, up4.c2
  HPFG_FUSED_CASE(1, 1, HPFG_KIND_BNACT, HPFG_KIND_PLAIN, 4, 2, 1, 1)   // out_conv
  HPFG_FUSED_CASE(1, 2, HPFG_KIND_POOL, HPFG_KIND_DZ, 8, 1, 1, 0)       // down1.c1
  HPFG_FUSED_CASE(2, 2, HPFG_KIND_BNACT, HPFG_KIND_DZ, 8, 1, 2, 1)      // down1.c2, up3.c2
  HPFG_FUSED_CASE(2, 1, HPFG_KIND_CAT, HPFG_KIND_DZ, 8, 1, 2, 1)        // up4.c1
#undef HPFG_FUSED_CASE
  return grid_only ? 0 : -2;
}

int check(const HpfgFusedBwdArgs* a) {
  HPFG_ARG_CHECK(a, "fused_bwd: null args");
  const HpfgConvArgs& d = a->d;
  if (d.taps != 9 || d.H % T || d.W % T || (d.math & 0xff) != HPFG_MATH_BF16X3) return 1;
  if (a->CinPad % 16 || a->CoutPad % 16 || a->CinPad > 64 || a->CoutPad > 32) return 1;
  return 0;
}

}  // namespace

extern "C" int hpfg_fused_bwd_grid(const HpfgFusedBwdArgs* a) {
  const int c = check(a);
  if (c) return c < 0 ? -1 : 0;
  return dispatch(*a, nullptr, true);
}

extern "C" int hpfg_fused_bwd(const HpfgFusedBwdArgs* a, void* stream) {
  const int c = check(a);
  if (c < 0) return -1;
  HPFG_ARG_CHECK(c == 0, "fused_bwd: 3x3 bf16x3 layers with H, W multiples of 16, CinPad <= 64 and CoutPad <= 32 only");
  const HpfgConvArgs& d = a->d;
  const bool wonly = !d.out;      // no dX wanted: weight gradient only (first layer)
  HPFG_ARG_CHECK(a->slab && d.a0.z && (wonly ? (!d.wpk && !d.bwd_stats && !d.out2) : d.wpk != nullptr), "fused_bwd: slab and the dZ source are required; wpk with out");
  HPFG_ARG_CHECK(d.Cout == a->Cin && d.CoutPad == a->CinPad && d.a0.C == a->Cout && a->xa0.C + a->xa1.C == a->Cin,
                 "fused_bwd: the dgrad side must describe the same layer (d.Cout == Cin, d.a0.C == Cout, input channels == Cin)");
  HPFG_ARG_CHECK((wonly || a->Cin % 8 == 0) && (a->Cout % 8 == 0 || d.a0.mode == HPFG_ACT_PLAIN || d.a0.mode == HPFG_ACT_STRIDED), "fused_bwd: Cin (and Cout behind a DZ source) must be multiples of 8");
#ifndef HPFG_TRACE      // (the diagnostics build takes its stamp buffer through d.bias)
  HPFG_ARG_CHECK(!d.bias, "fused_bwd: the dgrad side takes no bias");
#endif
  HPFG_ARG_CHECK(d.a1.mode == 0 && !d.a1.z, "fused_bwd: the dgrad side takes no second source");
  HPFG_ARG_CHECK(wonly || (d.Cout == d.CoutPad && d.out_pstride % 4 == 0 && d.out2_pstride % 4 == 0), "fused_bwd: Cin must be a multiple of 16 and the dX pixel strides of 4");
  if (hpfg_kind_of(a->xa0, a->xa1) == HPFG_KIND_CAT)
    HPFG_ARG_CHECK(a->xa0.C == a->xa1.C && a->xa0.C % 16 == 0, "fused_bwd: a concatenated input must be two halves of a multiple of 16 channels");
  HPFG_ARG_CHECK(!d.out_split || (d.out2 && d.out_split % 16 == 0 && d.out_split < d.Cout && !d.bwd_stats),
                 "fused_bwd: out_split needs out2, a multiple of 16 below Cin, and no bwd_stats");
  if (d.bwd_stats) {
    HPFG_ARG_CHECK((d.stat_partials || d.stat_acc) && d.bwd_of.z && d.bwd_of.bn, "fused_bwd: bwd_stats needs stat_partials or stat_acc, and bwd_of.z / .bn");
    HPFG_ARG_CHECK(!d.stat_acc || (d.stat_shards >= 1 && d.stat_shards <= HPFG_ACC_MAX_SHARDS && (d.stat_shards & (d.stat_shards - 1)) == 0),
                   "fused_bwd: bad stat_shards %d", d.stat_shards);
    HPFG_ARG_CHECK(d.bwd_of.C == d.Cout && d.Cout == d.CoutPad && d.bwd_of.Hs == d.H && d.bwd_of.Ws == d.W && d.bwd_of.pstride % 4 == 0,
                   "fused_bwd: bwd_of must describe a layer with C == Cin == CinPad at the layer's size");
    // the epilogue takes z and the scale / shift rows of that layer from the staged input tile
    HPFG_ARG_CHECK(a->xa0.mode == HPFG_ACT_BNACT && a->xa1.mode == 0 && d.bwd_of.z == a->xa0.z && d.bwd_of.bn == a->xa0.bn &&
                       d.bwd_of.bn_coff == a->xa0.bn_coff && d.bwd_of.pstride == a->xa0.pstride,
                   "fused_bwd: bwd_of must be the producer of this layer's (BNACT) input");
  } else {
    HPFG_ARG_CHECK(!d.stat_partials && !d.stat_acc, "fused_bwd: stat_partials / stat_acc without bwd_stats");
  }
  const int r = dispatch(*a, (hipStream_t)stream, false);
  HPFG_ARG_CHECK(r != -2, "fused_bwd: no instantiation for CinPad %d, CoutPad %d, input kind %d, dZ kind %d", a->CinPad, a->CoutPad,
                 hpfg_kind_of(a->xa0, a->xa1), hpfg_kind_of(a->d.a0, a->d.a1));
  HPFG_ARG_CHECK(r != -3, "fused_bwd: bwd_stats needs a BNACT input");
  HPFG_ARG_CHECK(r == 0, "fused_bwd: launch failed");
  return 0;
}
