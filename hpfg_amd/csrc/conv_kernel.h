// Implicit-GEMM conv kernel template (see conv.hip for the description); instantiated per loader kind in conv_k*.hip.
#pragma once
#include "common.h"

namespace hpfg_conv {

constexpr int KC = 16;      // input channels staged per chunk
constexpr int PS = KC + 1;  // LDS pixel stride (floats)

template <int TH_, int TW_, int WM_, int WN_, int NI_, int TAPS_>
struct Cfg {
  static constexpr int TH = TH_, TW = TW_, WM = WM_, WN = WN_, NI = NI_, TAPS = TAPS_;
  static constexpr int MI = TH * TW / 16 / WM;
  static constexpr int BN = 16 * NI * WN;
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HP = TH + 2 * HALO, WP = TW + 2 * HALO;
  static constexpr int LDS_FLOATS = HP * WP * PS;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(MI >= 1 && MI * WM * 16 == TH * TW, "tile must split into 16-pixel MFMA rows");
};

// Stage one float4 (4 input channels of one tile pixel) of chunk CH: piece index I covers tile slots [I*256, I*256+256).
#define HPFG_STAGE_LOAD(I, CH, DST)                                                                                     \
  {                                                                                                                      \
    const int idx_ = tid + (I) * 256;                                                                                    \
    f32x4 v_ = {0.f, 0.f, 0.f, 0.f};                                                                                     \
    if (idx_ < C::HP * C::WP * 4) {                                                                                      \
      const int pix_ = idx_ >> 2, cq_ = idx_ & 3;                                                                        \
      const int gy_ = ty0 + pix_ / C::WP - C::HALO, gx_ = tx0 + pix_ % C::WP - C::HALO;                                  \
      if (gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W) v_ = kind_load4<KIND>(p.a0, cx0, p.a1, cx1, n, gy_, gx_, (CH) * KC + cq_ * 4); \
    }                                                                                                                    \
    DST = v_;                                                                                                            \
  }
#define HPFG_STAGE_STORE(I, BUF, SRC)                                                                                    \
  {                                                                                                                      \
    const int idx_ = tid + (I) * 256;                                                                                    \
    if (idx_ < C::HP * C::WP * 4) {                                                                                      \
      float* d_ = (BUF) + (idx_ >> 2) * PS + (idx_ & 3) * 4;                                                             \
      d_[0] = SRC[0]; d_[1] = SRC[1]; d_[2] = SRC[2]; d_[3] = SRC[3];                                                    \
    }                                                                                                                    \
  }
#define HPFG_LOAD_B(CH, TAP, BF)                                                                                         \
  _Pragma("unroll") for (int j = 0; j < C::NI; ++j) BF[j] = wpk[((long)((TAP) * nchunks + (CH)) * ntn + nt0 + j) * 64 + lane];

template <class C, int KIND>
__global__ __launch_bounds__(256) void conv_mfma_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  constexpr int NBUF = C::TAPS == 9 ? 2 : 1;                      // 3x3: LDS tile double-buffered across input-channel chunks
  constexpr int NLD = (C::HP * C::WP * 4 + 255) / 256;            // float4 pieces per thread per chunk
  static_assert(C::TAPS == 1 || NLD <= 7, "stage pipeline assumes <= 7 pieces per chunk");
  __shared__ float lds[NBUF * C::LDS_FLOATS > 2 * 4 * C::BN ? NBUF * C::LDS_FLOATS : 2 * 4 * C::BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % C::WM, wn = wave / C::WM;
  const int tile = blockIdx.x, n = blockIdx.y, cb = blockIdx.z;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const int H = p.H, W = p.W;
  const ActCtx cx0 = make_ctx(p.a0), cx1 = make_ctx(p.a1);

  f32x4 acc[C::MI][C::NI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = ((pxl / C::TW) * C::WP + (pxl % C::TW)) * PS + (lane >> 4);
  }

  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + KC - 1) / KC;
  const int ntn = p.CoutPad / 16;                       // n-tiles in the packed weights
  const int nt0 = (cb * C::WN + wn) * C::NI;            // first n-tile of this wave
  const f32x4* wpk = reinterpret_cast<const f32x4*>(p.wpk);

  if (C::TAPS == 9) {
    // Software pipeline over input-channel chunks: while the MFMAs of chunk ch read LDS buffer ch&1, the activated tile of
    // chunk ch+1 is fetched piecewise (one float4 per thread per tap step, consumed two tap steps later) into the other
    // buffer; B fragments are fetched one (chunk, tap) ahead.  One barrier per chunk.
    for (int i = 0; i < NLD; ++i) {
      f32x4 t;
      HPFG_STAGE_LOAD(i, 0, t)
      HPFG_STAGE_STORE(i, lds, t)
    }
    f32x4 bcur[C::NI], bnxt[C::NI];
    HPFG_LOAD_B(0, 0, bcur)
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
      const float* cur = lds + (ch & 1) * C::LDS_FLOATS;
      float* nxt = lds + ((ch + 1) & 1) * C::LDS_FLOATS;
      const bool more = ch + 1 < nchunks;
      f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        if (more) {
          if (tap >= 2 && tap - 2 < NLD) {
            if ((tap & 1) == 0) HPFG_STAGE_STORE(tap - 2, nxt, st0) else HPFG_STAGE_STORE(tap - 2, nxt, st1)
          }
          if (tap < NLD) {
            if ((tap & 1) == 0) HPFG_STAGE_LOAD(tap, ch + 1, st0) else HPFG_STAGE_LOAD(tap, ch + 1, st1)
          }
        }
        if (tap + 1 < 9) {
          HPFG_LOAD_B(ch, tap + 1, bnxt)
        } else if (more) {
          HPFG_LOAD_B(ch + 1, 0, bnxt)
        }
        const int toff = ((tap / 3) * C::WP + (tap % 3)) * PS;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          float a[C::MI];
#pragma unroll
          for (int m = 0; m < C::MI; ++m) a[m] = cur[aoff[m] + toff + ks * 4];
#pragma unroll
          for (int m = 0; m < C::MI; ++m)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bcur[j][ks], acc[m][j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < C::NI; ++j) bcur[j] = bnxt[j];
      }
      __syncthreads();
    }
  } else {
    for (int ch = 0; ch < nchunks; ++ch) {
      __syncthreads();
      for (int i = 0; i < NLD; ++i) {
        f32x4 t;
        HPFG_STAGE_LOAD(i, ch, t)
        HPFG_STAGE_STORE(i, lds, t)
      }
      f32x4 bf[C::NI];
      HPFG_LOAD_B(ch, 0, bf)
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float a[C::MI];
#pragma unroll
        for (int m = 0; m < C::MI; ++m) a[m] = lds[aoff[m] + ks * 4];
#pragma unroll
        for (int m = 0; m < C::MI; ++m)
#pragma unroll
          for (int j = 0; j < C::NI; ++j) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bf[j][ks], acc[m][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: + bias, store raw output, per-channel partial sums for BatchNorm ----
  float s1[C::NI], s2[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
    const int co = (nt0 + j) * 16 + (lane & 15);
    const float b = (p.bias && co < p.CoutPad) ? p.bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < C::MI; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int pxl = (wm * C::MI + m) * 16 + (lane >> 4) * 4 + r;
        int gy = ty0 + pxl / C::TW, gx = tx0 + pxl % C::TW;
        float v = acc[m][j][r] + b;
        if (gy < H && gx < W && co < p.Cout) {
          if (p.out_split && co >= p.out_split) p.out2[((long)(n * H + gy) * W + gx) * p.out2_pstride + (co - p.out_split)] = v;
          else p.out[((long)(n * H + gy) * W + gx) * p.out_pstride + co] = v;
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
  }
  if (p.stat_partials) {
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      s1[j] += __shfl_xor(s1[j], 16);
      s2[j] += __shfl_xor(s2[j], 16);
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();   // all waves are done reading the tile; reuse LDS as [2][WM][BN]
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        int cl = (wn * C::NI + j) * 16 + lane;
        lds[(0 * C::WM + wm) * C::BN + cl] = s1[j];
        lds[(1 * C::WM + wm) * C::BN + cl] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * C::BN) {
      int which = tid / C::BN, cl = tid % C::BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < C::WM; ++w) t += lds[(which * C::WM + w) * C::BN + cl];
      int co = cb * C::BN + cl;
      long blk = (long)n * (tiles_x * tiles_y) + tile;
      if (co < p.CoutPad) p.stat_partials[(blk * 2 + which) * p.CoutPad + co] = t;
    }
  }
}

#undef HPFG_STAGE_LOAD
#undef HPFG_STAGE_STORE
#undef HPFG_LOAD_B

template <class C, int KIND>
int launch_cfg(const HpfgConvArgs& a, hipStream_t st) {
  int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  dim3 grid(tx * ty, a.N, a.CoutPad / C::BN);
  hipLaunchKernelGGL((conv_mfma_kernel<C, KIND>), grid, dim3(256), 0, st, a, tx, ty);
  return hpfg_launch_status("conv_mfma_kernel");
}

// one translation unit per loader kind instantiates these
template <int KIND, int TAPS>
int conv_dispatch_kind(const HpfgConvArgs& a, hipStream_t st) {
  const bool big = (a.H % 16 == 0) && (a.W % 16 == 0);
  const int cp = a.CoutPad;
  if (big) {
    if (cp % 64 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 4, TAPS>, KIND>(a, st);
    if (cp % 32 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 2, TAPS>, KIND>(a, st);
    return launch_cfg<Cfg<16, 16, 4, 1, 1, TAPS>, KIND>(a, st);
  }
  if (cp % 128 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 2, TAPS>, KIND>(a, st);
  if (cp % 64 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 1, TAPS>, KIND>(a, st);
  if (cp % 32 == 0) return launch_cfg<Cfg<8, 8, 2, 2, 1, TAPS>, KIND>(a, st);
  return launch_cfg<Cfg<8, 8, 4, 1, 1, TAPS>, KIND>(a, st);
}


}  // namespace hpfg_conv

// entry points defined by the per-kind translation units
int hpfg_conv_launch_plain(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv_launch_bnact(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv_launch_pool(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv_launch_cat(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv_launch_dz(const HpfgConvArgs& a, hipStream_t st);
