// Implicit-GEMM 3x3 / 1x1 convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3" split precision).
//
// Every fp32 operand is split as x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a product is evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation inside v_mfma_f32_16x16x32_bf16 (the dropped lo*lo term is 2^-16 relative).
// Three bf16 MFMAs replace sixteen fp32-input MFMAs' worth of cycles (MI355X_MICROARCH.md: f32-in MFMA = 1/16 of the bf16 rate),
// a 5.3x higher contraction roof at ~1e-5 relative error per layer -- inside the 1e-3 parity budget (tests/test_gpu_bf16x3.py).
//
// Same mapping, loaders, epilogue and BN partial sums as conv_kernel.h; what differs:
//   * LDS tile: bf16 planes [channel group of 8][hi|lo][pixel slot][8 ch], 16 B per (pixel, group) so an A fragment is one
//     ds_read_b128; the plane size is a multiple of 256 B and the 16 pixels of an MFMA row tile map to distinct 16-B slots
//     mod 16, so every ds_read_b128 lane group is bank-conflict free;
//   * K packing of one MFMA (K = 32): KC = 32 -> one tap x 32 input channels (small tiles, channel-rich layers);
//     KC = 16 -> two taps x 16 input channels (16x16 tiles of the 16/32-channel layers; taps padded 9 -> 10 with zero weights);
//   * B fragments come pre-split (hi, lo) in fragment order from hpfg_pack_weights.
#pragma once
#include "common.h"

namespace hpfg_conv16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TH_, int TW_, int WM_, int WN_, int NI_, int TAPS_, int KC_>
struct Cfg {
  static constexpr int TH = TH_, TW = TW_, WM = WM_, WN = WN_, NI = NI_, TAPS = TAPS_, KC = KC_;
  static constexpr int MI = TH * TW / 16 / WM;
  static constexpr int BN = 16 * NI * WN;
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HP = TH + 2 * HALO, WP = TW + 2 * HALO;
  // row stride in 16-B slots: an MFMA row tile's 16 pixels must hit 16 distinct slots mod 16
  static constexpr int RS = TW == 16 ? WP : (TAPS == 9 ? 24 : 8);
  static constexpr int NSLOT = (HP * RS + 15) / 16 * 16;
  static constexpr int NG = KC / 8;                      // channel groups of 8
  static constexpr int PLANE = NSLOT * 16;               // bytes
  static constexpr int BUF_BYTES = NG * 2 * PLANE;
  static constexpr int KSTEPS = TAPS == 1 ? 1 : (KC == 32 ? 9 : 5);   // MFMA k-steps per input-channel chunk
  static constexpr int NPIECE = HP * WP * NG;            // (pixel, group) staging pieces per chunk
  static constexpr int NLD = (NPIECE + 255) / 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(KC == 32 || (KC == 16 && TAPS == 9), "two-tap K packing only for 3x3");
  static_assert(TAPS == 1 || NLD + 2 <= KSTEPS, "stage pipeline must fit into the k-steps of a chunk");
};

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h0 = (__bf16)a[j], h1 = (__bf16)b[j];
    hi[j] = h0;
    hi[4 + j] = h1;
    lo[j] = (__bf16)(a[j] - (float)h0);
    lo[4 + j] = (__bf16)(b[j] - (float)h1);
  }
}

#define HPFG16_STAGE_LOAD(I, CH, NN, TY, TX, V0, V1)                                                                     \
  {                                                                                                                      \
    const int idx_ = tid + (I) * 256;                                                                                    \
    V0 = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                      \
    V1 = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                      \
    if (idx_ < C::NPIECE) {                                                                                              \
      const int pix_ = idx_ / C::NG, g_ = idx_ % C::NG;                                                                  \
      const int gy_ = (TY) + pix_ / C::WP - C::HALO, gx_ = (TX) + pix_ % C::WP - C::HALO;                                \
      if (gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W && !(p.math & 0x400)) {                                             \
        V0 = kind_load4<KIND>(p.a0, cx0, p.a1, cx1, NN, gy_, gx_, (CH) * C::KC + g_ * 8);                                \
        V1 = kind_load4<KIND>(p.a0, cx0, p.a1, cx1, NN, gy_, gx_, (CH) * C::KC + g_ * 8 + 4);                            \
      }                                                                                                                  \
    }                                                                                                                    \
  }
#define HPFG16_STAGE_STORE(I, BUF, V0, V1)                                                                               \
  {                                                                                                                      \
    const int idx_ = tid + (I) * 256;                                                                                    \
    if (idx_ < C::NPIECE) {                                                                                              \
      const int pix_ = idx_ / C::NG, g_ = idx_ % C::NG;                                                                  \
      const int slot_ = (pix_ / C::WP) * C::RS + pix_ % C::WP;                                                           \
      bf16x8 hi_, lo_;                                                                                                   \
      split8(V0, V1, hi_, lo_);                                                                                          \
      *reinterpret_cast<bf16x8*>((BUF) + (g_ * 2 + 0) * C::PLANE + slot_ * 16) = hi_;                                    \
      *reinterpret_cast<bf16x8*>((BUF) + (g_ * 2 + 1) * C::PLANE + slot_ * 16) = lo_;                                    \
    }                                                                                                                    \
  }
// B fragments of global k-step KS (= chunk * KSTEPS + step): [ks][ntile][hi|lo][64 lanes] x 16 B
#define HPFG16_LOAD_B(KS, BH, BL)                                                                                        \
  _Pragma("unroll") for (int j = 0; j < C::NI; ++j) {                                                                    \
    const bf16x8* q_ = wpk + (((long)(KS) * ntn + nt0 + j) * 2) * 64 + lane;                                             \
    BH[j] = q_[0];                                                                                                       \
    BL[j] = q_[64];                                                                                                      \
  }

// Epilogue shared by both kernels: + bias, store the raw output tile, per-channel partial sums for BatchNorm.
template <class C>
__device__ __forceinline__ void conv16_epilogue(const HpfgConvArgs& p, f32x4 (&acc)[C::MI][C::NI], float* ldsf, int tid, int lane, int wm, int wn,
                                                int nt0, int cb, int n, int ty0, int tx0, long blk) {
  const int H = p.H, W = p.W;
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
          if (!(p.math & 0x100)) p.out[((long)(n * H + gy) * W + gx) * p.out_pstride + co] = v;
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
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        int cl = (wn * C::NI + j) * 16 + lane;
        ldsf[(0 * C::WM + wm) * C::BN + cl] = s1[j];
        ldsf[(1 * C::WM + wm) * C::BN + cl] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * C::BN) {
      int which = tid / C::BN, cl = tid % C::BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < C::WM; ++w) t += ldsf[(which * C::WM + w) * C::BN + cl];
      int co = cb * C::BN + cl;
      if (co < p.CoutPad) p.stat_partials[(blk * 2 + which) * p.CoutPad + co] = t;
    }
    __syncthreads();
  }
}

// 3x3: persistent workgroups.  A workgroup walks tiles w = blockIdx.x, blockIdx.x + gridDim.x, ... (tile-major inside an image,
// so consecutive tiles share halos in L2) and streams (tile, input-channel chunk) items through a double-buffered LDS tile:
// while the MFMAs of item i run, the activated tile of item i+1 -- the next chunk, or chunk 0 of the workgroup's NEXT tile --
// is fetched piecewise into the other buffer, so the load latency of a tile is hidden behind the previous tile's MFMAs and
// output stores even for single-chunk (16-channel) layers.  One barrier per item.
template <class C, int KIND>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 9, "persistent kernel is the 3x3 path");
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * C::BUF_BYTES + STAT_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds + 2 * C::BUF_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % C::WM, wn = wave / C::WM;
  const int cb = blockIdx.y;
  const int H = p.H, W = p.W;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * p.N;
  const ActCtx cx0 = make_ctx(p.a0), cx1 = make_ctx(p.a1);

  f32x4 acc[C::MI][C::NI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kg = lane >> 4;
  const int gl = C::KC == 32 ? kg : (kg & 1);
  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (gl * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }

  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + C::KC - 1) / C::KC;
  const int ntn = p.CoutPad / 16;
  const int nt0 = (cb * C::WN + wn) * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);

  int w = blockIdx.x;
  if (w >= nwork) return;
  int n = w / ntiles, tl = w % ntiles;
  int ty0 = (tl / tiles_x) * C::TH, tx0 = (tl % tiles_x) * C::TW;
  for (int i = 0; i < C::NLD; ++i) {
    f32x4 v0, v1;
    HPFG16_STAGE_LOAD(i, 0, n, ty0, tx0, v0, v1)
    HPFG16_STAGE_STORE(i, lds, v0, v1)
  }
  bf16x8 bh[C::NI], bl[C::NI], nh[C::NI], nl[C::NI];
  HPFG16_LOAD_B(0, bh, bl)
  __syncthreads();
  int item = 0;
  while (true) {
    for (int ch = 0; ch < nchunks; ++ch, ++item) {
      const unsigned char* cur = lds + (item & 1) * C::BUF_BYTES;
      unsigned char* nxt = lds + ((item + 1) & 1) * C::BUF_BYTES;
      // the item after this one: next chunk of this tile, or chunk 0 of this workgroup's next tile
      const bool last_chunk = ch + 1 == nchunks;
      const int wn_ = w + (int)gridDim.x;
      const bool more = !last_chunk || wn_ < nwork;
      const int nch = last_chunk ? 0 : ch + 1;
      int nn = n, nty = ty0, ntx = tx0;
      if (last_chunk && more) {
        nn = wn_ / ntiles;
        const int t2 = wn_ % ntiles;
        nty = (t2 / tiles_x) * C::TH;
        ntx = (t2 % tiles_x) * C::TW;
      }
      f32x4 s0a = {0.f, 0.f, 0.f, 0.f}, s0b = s0a, s1a = s0a, s1b = s0a;
#pragma unroll 1
      for (int s = 0; s < C::KSTEPS; ++s) {
        if (more) {
          if (s >= 2 && s - 2 < C::NLD) {
            if ((s & 1) == 0) HPFG16_STAGE_STORE(s - 2, nxt, s0a, s0b) else HPFG16_STAGE_STORE(s - 2, nxt, s1a, s1b)
          }
          if (s < C::NLD) {
            if ((s & 1) == 0) HPFG16_STAGE_LOAD(s, nch, nn, nty, ntx, s0a, s0b) else HPFG16_STAGE_LOAD(s, nch, nn, nty, ntx, s1a, s1b)
          }
        }
        if (!(p.math & 0x800)) {
          if (s + 1 < C::KSTEPS) {
            HPFG16_LOAD_B(ch * C::KSTEPS + s + 1, nh, nl)
          } else if (more) {
            HPFG16_LOAD_B(nch * C::KSTEPS, nh, nl)
          }
        }
        int tap = C::KC == 32 ? s : 2 * s + (kg >> 1);
        tap = tap > 8 ? 8 : tap;
        const int toff = ((tap / 3) * C::RS + (tap % 3)) * 16;
        if (!(p.math & 0x200))
#pragma unroll
        for (int m = 0; m < C::MI; ++m) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff + C::PLANE);
#pragma unroll
          for (int j = 0; j < C::NI; ++j) {
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[m][j], 0, 0, 0);
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[m][j], 0, 0, 0);
            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[m][j], 0, 0, 0);
          }
        }
#pragma unroll
        for (int j = 0; j < C::NI; ++j) {
          bh[j] = nh[j];
          bl[j] = nl[j];
        }
      }
      __syncthreads();
    }
    conv16_epilogue<C>(p, acc, ldsf, tid, lane, wm, wn, nt0, cb, n, ty0, tx0, (long)w);
#pragma unroll
    for (int m = 0; m < C::MI; ++m)
#pragma unroll
      for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    w += gridDim.x;
    if (w >= nwork) break;
    n = w / ntiles;
    tl = w % ntiles;
    ty0 = (tl / tiles_x) * C::TH;
    tx0 = (tl % tiles_x) * C::TW;
  }
}

// 1x1: one tile per workgroup, K = 32 input channels per MFMA step, no halo.
template <class C, int KIND>
__global__ __launch_bounds__(256) void conv1x1_bf16x3_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 1, "1x1 path");
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[C::BUF_BYTES + STAT_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds + C::BUF_BYTES);
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
  const int kg = lane >> 4;
  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (kg * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }
  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + C::KC - 1) / C::KC;
  const int ntn = p.CoutPad / 16;
  const int nt0 = (cb * C::WN + wn) * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    for (int i = 0; i < C::NLD; ++i) {
      f32x4 v0, v1;
      HPFG16_STAGE_LOAD(i, ch, n, ty0, tx0, v0, v1)
      HPFG16_STAGE_STORE(i, lds, v0, v1)
    }
    bf16x8 bh[C::NI], bl[C::NI];
    HPFG16_LOAD_B(ch, bh, bl)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < C::MI; ++m) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + aoff[m]);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + aoff[m] + C::PLANE);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) {
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[m][j], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  conv16_epilogue<C>(p, acc, ldsf, tid, lane, wm, wn, nt0, cb, n, ty0, tx0, (long)n * (tiles_x * tiles_y) + tile);
}

#undef HPFG16_STAGE_LOAD
#undef HPFG16_STAGE_STORE
#undef HPFG16_LOAD_B

template <class C, int KIND>
int launch_cfg(const HpfgConvArgs& a, hipStream_t st) {
  int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  if constexpr (C::TAPS == 9) {
    // persistent grid: as many workgroups as stay resident (LDS-limited), never more than there are tiles
    const int lds_bytes = 2 * C::BUF_BYTES + 2 * 4 * C::BN * 4;
    int per_cu = 160 * 1024 / lds_bytes;
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    long nwork = (long)tx * ty * a.N;
    long gx = 256L * per_cu;
    if (gx > nwork) gx = nwork;
    dim3 grid((unsigned)gx, a.CoutPad / C::BN);
    hipLaunchKernelGGL((conv_bf16x3_kernel<C, KIND>), grid, dim3(256), 0, st, a, tx, ty);
  } else {
    dim3 grid(tx * ty, a.N, a.CoutPad / C::BN);
    hipLaunchKernelGGL((conv1x1_bf16x3_kernel<C, KIND>), grid, dim3(256), 0, st, a, tx, ty);
  }
  return hpfg_launch_status("conv_bf16x3_kernel");
}

template <int KIND, int TAPS>
int conv_dispatch_kind(const HpfgConvArgs& a, hipStream_t st) {
  const bool big = (a.H % 16 == 0) && (a.W % 16 == 0);
  const int cp = a.CoutPad;
  constexpr int KCB = TAPS == 9 ? 16 : 32;     // 16x16 tiles of 3x3 convs use the two-tap K packing
  if (big) {
    if (cp % 64 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 4, TAPS, KCB>, KIND>(a, st);
    if (cp % 32 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 2, TAPS, KCB>, KIND>(a, st);
    return launch_cfg<Cfg<16, 16, 4, 1, 1, TAPS, KCB>, KIND>(a, st);
  }
  if (cp % 128 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 2, TAPS, 32>, KIND>(a, st);
  if (cp % 64 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 1, TAPS, 32>, KIND>(a, st);
  if (cp % 32 == 0) return launch_cfg<Cfg<8, 8, 2, 2, 1, TAPS, 32>, KIND>(a, st);
  return launch_cfg<Cfg<8, 8, 4, 1, 1, TAPS, 32>, KIND>(a, st);
}

}  // namespace hpfg_conv16

int hpfg_conv16_launch_plain(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv16_launch_bnact(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv16_launch_pool(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv16_launch_cat(const HpfgConvArgs& a, hipStream_t st);
int hpfg_conv16_launch_dz(const HpfgConvArgs& a, hipStream_t st);
