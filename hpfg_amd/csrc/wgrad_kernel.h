// Weight-gradient kernel template (see wgrad.hip); instantiated per (input loader kind, gradient loader kind) in wgrad_k*.hip.
#pragma once
#include "common.h"

namespace hpfg_wg {

constexpr int TH = 8, TW = 16;  // pixel tile (one work item)
constexpr int PSA = 17;         // LDS pixel stride of the A tile (16 channels + 1)

template <int TAPS, int NJ>
struct WCfg {
  static constexpr int NW = TAPS == 9 ? 3 : NJ;           // waves per workgroup
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HP = TH + 2 * HALO, WP = TW + 2 * HALO;
  static constexpr int PSB = 16 * NJ + 1;                  // LDS pixel stride of the dZ tile
  static constexpr int LDS_A = HP * WP * PSA;
  static constexpr int LDS_B = TH * TW * PSB;
  static constexpr int NT = TAPS == 9 ? 3 : 1;             // taps per wave
  static constexpr int NA = TAPS == 9 ? NJ : 1;            // n-tiles per wave
};

// AK: loader kind of the conv input (HPFG_KIND_*), GK: loader kind of dZ (HPFG_KIND_DZ or HPFG_KIND_PLAIN)
template <int TAPS, int NJ, int AK, int GK>
__global__ __launch_bounds__((TAPS == 9 ? 192 : 64 * NJ)) void wgrad_kernel(HpfgWgradArgs p, int tiles_x, int tiles_y) {
  using C = WCfg<TAPS, NJ>;
  __shared__ float ldsA[C::LDS_A];
  __shared__ float ldsB[C::LDS_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = 64 * C::NW;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16 * NJ;
  const int H = p.H, W = p.W;
  const ActCtx cxa0 = make_ctx(p.a0), cxa1 = make_ctx(p.a1), cxg = make_ctx(p.g);
  const HpfgAct none = {};

  f32x4 acc[C::NT][C::NA];
#pragma unroll
  for (int t = 0; t < C::NT; ++t)
#pragma unroll
    for (int j = 0; j < C::NA; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = tiles_x * tiles_y;
  const int nwork = p.N * ntiles;
  for (int wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    __syncthreads();
#pragma unroll 1
    for (int idx = tid; idx < C::HP * C::WP * 4; idx += nthr) {
      int pix = idx >> 2, cq = idx & 3;
      int gy = ty0 + pix / C::WP - C::HALO, gx = tx0 + pix % C::WP - C::HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = kind_load4<AK>(p.a0, cxa0, p.a1, cxa1, n, gy, gx, ci0 + cq * 4);
      float* d = ldsA + pix * PSA + cq * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
#pragma unroll 1
    for (int idx = tid; idx < TH * TW * 4 * NJ; idx += nthr) {
      int pix = idx / (4 * NJ), cq = idx % (4 * NJ);
      int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < H && gx < W) v = kind_load4<GK>(p.g, cxg, none, cxg, n, gy, gx, co0 + cq * 4);
      float* d = ldsB + pix * C::PSB + cq * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < TH * TW / 4; ++ks) {
      const int pix = ks * 4 + (lane >> 4);            // k index = pixel
      const int r = pix / TW, c = pix % TW;
      if (TAPS == 9) {
        float b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = ldsB[pix * C::PSB + j * 16 + (lane & 15)];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float a = ldsA[((r + wave) * C::WP + c + kx) * PSA + (lane & 15)];
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[kx][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[kx][j], 0, 0, 0);
        }
      } else {
        float b = ldsB[pix * C::PSB + wave * 16 + (lane & 15)];
        float a = ldsA[pix * PSA + (lane & 15)];
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0][0], 0, 0, 0);
      }
    }
  }
  // slab[s][tap][ci][co]; C/D layout: row (ci) = (lane>>4)*4 + r, col (co) = lane & 15
  float* slab = p.slab + (long)blockIdx.x * p.taps * p.CinPad * p.CoutPad;
#pragma unroll
  for (int t = 0; t < C::NT; ++t)
#pragma unroll
    for (int j = 0; j < C::NA; ++j) {
      const int tap = TAPS == 9 ? wave * 3 + t : 0;
      const int nt = TAPS == 9 ? j : wave;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = ci0 + (lane >> 4) * 4 + r, co = co0 + nt * 16 + (lane & 15);
        slab[((long)tap * p.CinPad + ci) * p.CoutPad + co] = acc[t][j][r];
      }
    }
}

inline int pick_nj(int CoutPad) { return CoutPad % 64 == 0 ? 4 : (CoutPad % 32 == 0 ? 2 : 1); }

template <int TAPS, int AK, int GK>
int launch_wgrad(const HpfgWgradArgs& a, hipStream_t st) {
  const int nj = pick_nj(a.CoutPad);
  const int tx = (a.W + TW - 1) / TW, ty = (a.H + TH - 1) / TH;
  dim3 grid(a.S, a.CinPad / 16, a.CoutPad / (16 * nj));
  if (nj == 4) hipLaunchKernelGGL((wgrad_kernel<TAPS, 4, AK, GK>), grid, dim3(64 * WCfg<TAPS, 4>::NW), 0, st, a, tx, ty);
  else if (nj == 2) hipLaunchKernelGGL((wgrad_kernel<TAPS, 2, AK, GK>), grid, dim3(64 * WCfg<TAPS, 2>::NW), 0, st, a, tx, ty);
  else hipLaunchKernelGGL((wgrad_kernel<TAPS, 1, AK, GK>), grid, dim3(64 * WCfg<TAPS, 1>::NW), 0, st, a, tx, ty);
  return hpfg_launch_status("wgrad_kernel");
}

}  // namespace hpfg_wg

int hpfg_wgrad_launch_dz(const HpfgWgradArgs& a, int akind, hipStream_t st);      // dZ from BN backward (3x3 only)
int hpfg_wgrad_launch_plain(const HpfgWgradArgs& a, int akind, hipStream_t st);   // dZ given directly (1x1, out_conv)
