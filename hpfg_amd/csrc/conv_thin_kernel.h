// 3x3 convolution of the thin layers -- forward, and the dgrad of a 32-channel layer into a 64-channel input (dZ source) -- (16 / 32 input channels, or a 32-channel concat; 16 / 32 output channels; H and W multiples
// of 16) on the bf16 matrix cores with split-precision operands -- the forward twin of fused_bwd_kernel.h, built on what that kernel's
// timelines and counters showed:
//   * the WHOLE 18 x 18 halo tile of the (virtual) input -- all input channels -- is staged at once as [chunk][hi|lo][pixel slot][16 ch]
//     (32 bytes per pixel and plane: conflict-free ds_read_b128 fragments), not chunk by chunk through a double buffer;
//   * every load of the next tile is in flight (in registers) during the MFMAs and the epilogue of the current one, so a tile costs one
//     exposed memory round trip;
//   * nothing in the MFMA phase touches global memory: the weight fragments do not depend on the tile and stay in registers for the whole
//     kernel (<= 80 VGPRs) or in LDS, the BatchNorm tables of the loader live in LDS;
//   * the concat loader's upsampled half reads its low-resolution source patch once per tile (one piece per thread) into LDS and blends
//     the four bilinear taps from there (conv_bf16_kernel.h);
//   * a workgroup walks a contiguous range of its XCD's tiles; the BatchNorm partial sums stay in registers across tiles, one row per
//     workgroup.
// Values and MFMA order are those of conv_bf16x3_kernel, so the raw output is bit-identical to it; only the partition of the BatchNorm
// partial sums (rows = workgroups) differs.
#pragma once
#include <type_traits>
#include "conv_bf16_kernel.h"

namespace hpfg_thin {

using namespace hpfg_stage;
using hpfg_conv16::Cfg;
using hpfg_conv16::clampi;
using hpfg_conv16::up_base;

constexpr int T = 16, HP = 18, RS = 18;                 // tile edge, halo tile edge, row stride in pixel slots
constexpr int NSLOT = (HP * HP + 15) / 16 * 16;         // 336
constexpr int PL = NSLOT * 32;                          // bytes per hi / lo plane of one 16-channel chunk
constexpr int CH = 2 * PL;
constexpr int USH = T / 2 + 3, USW = T / 2 + 3;         // low-resolution source patch of an upsampled chunk (rows x columns)
constexpr int UBYTES = USH * USW * 16 * 4;              // fp32, [pixel][16 ch]

template <int CI, int CO, int AK, int NW>
struct Geo {
  static constexpr bool CATK = AK == HPFG_KIND_CAT;
  static constexpr int AK0 = CATK ? HPFG_KIND_BNACT : AK;          // loader kind of the chunks read through a0 (concat: the skip half)
  static constexpr int NA0 = CATK ? CI / 2 : CI, NA1 = CATK ? CI / 2 : 0;
  static constexpr int NTH = 64 * NW;
  static constexpr int ND = (HP * HP * 2 + NTH - 1) / NTH;         // staging pieces (8 channels of one pixel) per thread and chunk
  static constexpr int MI = 16 / NW;                               // 16-pixel MFMA tiles per wave
  static constexpr int KS = 5 * CI;                                // k-steps: 2 taps x 16 channels each
  static constexpr bool BREG = KS * CO * 8 <= 80;                  // weight fragments: registers or LDS
  static constexpr int TROWS = AK == HPFG_KIND_DZ ? 5 : 2;         // table rows of the loader: scale, shift [, k1, k2, k3 of a dZ source]
  static constexpr int TAB = TROWS * 16 * NA0 * 4;
  static constexpr int STAT = 2 * NW * 16 * CO * 4;
  static constexpr int BFR = BREG ? 0 : KS * CO * 2 * 1024;
  static constexpr int OFF_TAB = CI * CH, OFF_STAT = OFF_TAB + TAB, OFF_U = OFF_STAT + STAT, OFF_B = OFF_U + NA1 * UBYTES;
  static constexpr int LDS = OFF_B + BFR;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(USH * USW * 2 <= NTH, "one source piece per thread");
};

template <int CI, int CO, int AK, int NW, int WGS>
__global__ __launch_bounds__(64 * NW, WGS) void conv_thin_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  using C = Cfg<16, 16, 4, 1, CO, 9, 16>;               // (weight-fragment indexing: CO output-channel tiles)
  using G = Geo<CI, CO, AK, NW>;
  constexpr int AK0 = G::AK0, NA0 = G::NA0, NA1 = G::NA1, NTH = G::NTH, ND = G::ND, MI = G::MI, KS = G::KS;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  float* tabA = reinterpret_cast<float*>(lds + G::OFF_TAB);
  float* ldsf = reinterpret_cast<float*>(lds + G::OFF_STAT);
  float* ldsU = reinterpret_cast<float*>(lds + G::OFF_U);
  const bf16x8* ldsB = reinterpret_cast<const bf16x8*>(lds + G::OFF_B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.H, W = p.W;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * p.N;
  const HpfgAct none = {};
  HpfgAct aS = p.a0;
  if (G::CATK) aS.drop_p = 0.f;                          // (the skip half of a concat carries no dropout: model/unet.py:57 concatenates block outputs)
  const ActCtx cxa = make_ctx(aS);
  const int gsel = tid & 1;                              // this thread's 8-channel group inside a 16-channel chunk

  const int ntn = p.CoutPad / 16;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);
  constexpr int PB = G::BREG ? KS : 0;
  bf16x8 pbh[PB > 0 ? PB : 1][CO], pbl[PB > 0 ? PB : 1][CO];
#pragma unroll
  for (int k = 0; k < PB; ++k) hpfg_conv16::load_b<C>(pbh[k], pbl[k], wpk, k, ntn, 0, lane);
  if (!G::BREG) {
    bf16x8* dst = reinterpret_cast<bf16x8*>(lds + G::OFF_B);
    for (int i = tid; i < KS * CO * 2 * 64; i += NTH) dst[i] = wpk[i];
  }

  // fragment offsets (conv_bf16_kernel.h): pixel tile m of this wave; k-group = (tap parity, 8-channel half)
  const int kg = lane >> 4, gl = kg & 1;
  int aoff[MI], toff[5];
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    const int pxl = (wave * MI + m) * 16 + (lane & 15);
    aoff[m] = ((pxl / T) * RS + (pxl % T)) * 32 + gl * 16;
  }
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    int tap = 2 * s + (kg >> 1);
    tap = tap > 8 ? 8 : tap;                             // tap 9 re-reads tap 8 against zero weights
    toff[s] = ((tap / 3) * RS + (tap % 3)) * 32;
  }
  f32x4 s1[CO], s2[CO], bias[CO];
#pragma unroll
  for (int j = 0; j < CO; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
    bias[j] = s1[j];
    const int co = j * 16 + (lane >> 4) * 4;
    if (p.bias && co < p.CoutPad) bias[j] = ld4(p.bias, co);
  }

  // XCD-aware work mapping (conv_bf16_kernel.h): every XCD group walks a contiguous range of tiles
  const int nx = gridDim.x >= 8 ? 8 : 1;
  const int xg = (int)blockIdx.x % nx, xj = (int)blockIdx.x / nx;
  const int per_x = (nwork + nx - 1) / nx;
  const int wend = (xg + 1) * per_x < nwork ? (xg + 1) * per_x : nwork;
  const int GS = ((int)gridDim.x - xg + nx - 1) / nx;

  RawPiece<AK0> raw[NA0][ND];
  f32x4 rawU[NA1 > 0 ? NA1 : 1][2];

  // every load of one tile; live == false: no further tile -- all lanes read pixel (0, 0, 0), one cache line, and nothing is made of it
  auto issue = [&](int n, int ty0, int tx0, bool live) {
#pragma unroll
    for (int c = 0; c < NA0; ++c) {
      const int c0 = c * 16 + gsel * 8;
      const bool chv = c0 < aS.C;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * NTH, pix = idx >> 1;
        const int gy = ty0 + pix / HP - 1, gx = tx0 + pix % HP - 1;
        const bool ok = live && idx < HP * HP * 2 && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
        issue_piece<AK0>(raw[c][i], aS, none, cxa, n, live ? clampi(gy, 0, H - 1) : 0, live ? clampi(gx, 0, W - 1) : 0, chv ? c0 : 0, ok);
      }
    }
    if (NA1 > 0) {
      const int sy_base = up_base(ty0 - 1, p.a1.Hs), sx_base = up_base(tx0 - 1, p.a1.Ws);
      const int pix = tid >> 1;
      const int sy = live ? clampi(sy_base + pix / USW, 0, p.a1.Hs - 1) : 0, sx = live ? clampi(sx_base + pix % USW, 0, p.a1.Ws - 1) : 0;
#pragma unroll
      for (int c = 0; c < NA1; ++c) {
        const int cu = c * 16 + gsel * 8;
        const int off = ((n * p.a1.Hs + sy) * p.a1.Ws + sx) * p.a1.pstride + (cu < p.a1.C ? cu : 0);
        rawU[c][0] = ld4(p.a1.z, off);
        rawU[c][1] = ld4(p.a1.z, off + 4);
      }
    }
  };

  int w = xg * per_x + xj;
  if (w < wend) issue(w / ntiles, ((w % ntiles) / tiles_x) * T, ((w % ntiles) % tiles_x) * T, true);
  // BatchNorm coefficients of the source -> LDS, behind the first tile's loads (one exposed round trip for both): the table rows, or --
  // forward kinds with HpfgAct.bn_acc -- scale / shift derived here from the producer's sum accumulators (no finalize launch in between)
  if (AK0 != HPFG_KIND_DZ && aS.bn_acc) {
    for (int ch = tid; ch < 16 * NA0; ch += NTH) {
      float sc = 0.f, sh = 0.f;
      if (ch < aS.C) {
        const int cc = aS.bn_coff + ch;
        const float ga = aS.bn_gamma[cc], be = aS.bn_beta[cc];      // (requested with the accumulator words: one round trip)
        double s1, s2;
        hpfg_acc_read2(aS.bn_acc, aS.bn_stride, aS.bn_shards, cc, s1, s2, ga + be);
        const HpfgBnCoef q = hpfg_bn_coef(s1, s2, (double)aS.bn_count, aS.bn_eps, ga, be);
        sc = q.scale;
        sh = q.shift;
      }
      tabA[ch] = sc;
      tabA[16 * NA0 + ch] = sh;
    }
  } else if (AK0 == HPFG_KIND_DZ) {      // rows scale, shift, k1, k2, k3 (table, or k1 .. k3 derived from the backward sum accumulators)
    hpfg_dz_rows_to_lds(aS, tabA, 16 * NA0, 16 * NA0, tid, NTH);
  } else if (AK0 != HPFG_KIND_PLAIN) {
    for (int i = tid; i < G::TROWS * 16 * NA0; i += NTH) {
      const int r = i / (16 * NA0), ch = i % (16 * NA0);
      const int row = r == 0 ? HPFG_BN_SCALE : HPFG_BN_SHIFT;
      tabA[i] = ch < aS.C ? aS.bn[aS.bn_coff + row * aS.bn_stride + ch] : 0.f;
    }
  }
  __syncthreads();                                       // tables, weight fragments

  for (; w < wend; w += GS) {
    const int n = w / ntiles, ty0 = ((w % ntiles) / tiles_x) * T, tx0 = ((w % ntiles) % tiles_x) * T;
    // ---- prefetched raw data -> producer chain -> bf16 hi / lo -> LDS
#pragma unroll
    for (int c = 0; c < NA0; ++c) {
      const int c0 = c * 16 + gsel * 8;
      const bool chv = c0 < aS.C;
      Tab ta;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ta.sc[h] = ld4(tabA, 0 * 16 * NA0 + c0 + 4 * h);
        ta.sh[h] = ld4(tabA, 1 * 16 * NA0 + c0 + 4 * h);
        if (AK0 == HPFG_KIND_DZ) {
          ta.k1[h] = ld4(tabA, 2 * 16 * NA0 + c0 + 4 * h);
          ta.k2[h] = ld4(tabA, 3 * 16 * NA0 + c0 + 4 * h);
          ta.k3[h] = ld4(tabA, 4 * 16 * NA0 + c0 + 4 * h);
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * NTH, pix = idx >> 1;
        const int gy = ty0 + pix / HP - 1, gx = tx0 + pix % HP - 1;
        const bool ok = idx < HP * HP * 2 && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
        f32x4 v0, v1;
        finish_piece<AK0>(v0, v1, raw[c][i], ta, aS, none, cxa, n, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), chv ? c0 : 0, ok);
        if (idx < HP * HP * 2) {
          bf16x8 hi, lo;
          split8(v0, v1, hi, lo);
          unsigned char* o = lds + c * CH + ((pix / HP) * RS + pix % HP) * 32 + gsel * 16;
          *reinterpret_cast<bf16x8*>(o) = hi;
          *reinterpret_cast<bf16x8*>(o + PL) = lo;
        }
      }
    }
    if (NA1 > 0) {
      // park the low-res patches (fp32, [pixel][16 channels]), then every output piece blends its four taps from LDS
      if (tid < USH * USW * 2) {
#pragma unroll
        for (int c = 0; c < NA1; ++c) {
          float* d = ldsU + c * (UBYTES / 4) + (tid >> 1) * 16 + gsel * 8;
          *reinterpret_cast<f32x4*>(d) = rawU[c][0];
          *reinterpret_cast<f32x4*>(d + 4) = rawU[c][1];
        }
      }
      __syncthreads();
      const int sy_base = up_base(ty0 - 1, p.a1.Hs), sx_base = up_base(tx0 - 1, p.a1.Ws);
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * NTH, pix = idx >> 1;
        const int gy = ty0 + pix / HP - 1, gx = tx0 + pix % HP - 1;
        const bool inimg = idx < HP * HP * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        int y0, y1, x0, x1;
        float wy1, wx1;
        up_coord(clampi(gy, 0, H - 1), p.a1.Hs, y0, y1, wy1);
        up_coord(clampi(gx, 0, W - 1), p.a1.Ws, x0, x1, wx1);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const int o0 = (x0 - sx_base) * 16, o1 = (x1 - sx_base) * 16;
#pragma unroll
        for (int c = 0; c < NA1; ++c) {
          const bool ok = inimg && c * 16 + gsel * 8 < p.a1.C;
          const float* r0 = ldsU + c * (UBYTES / 4) + ((y0 - sy_base) * USW) * 16 + gsel * 8;
          const float* r1 = ldsU + c * (UBYTES / 4) + ((y1 - sy_base) * USW) * 16 + gsel * 8;
          const f32x4 a00 = *reinterpret_cast<const f32x4*>(r0 + o0), b00 = *reinterpret_cast<const f32x4*>(r0 + o0 + 4);
          const f32x4 a01 = *reinterpret_cast<const f32x4*>(r0 + o1), b01 = *reinterpret_cast<const f32x4*>(r0 + o1 + 4);
          const f32x4 a10 = *reinterpret_cast<const f32x4*>(r1 + o0), b10 = *reinterpret_cast<const f32x4*>(r1 + o0 + 4);
          const f32x4 a11 = *reinterpret_cast<const f32x4*>(r1 + o1), b11 = *reinterpret_cast<const f32x4*>(r1 + o1 + 4);
          f32x4 v0 = wy0 * (wx0 * a00 + wx1 * a01) + wy1 * (wx0 * a10 + wx1 * a11);     // same expression order as finish_piece<CAT>
          f32x4 v1 = wy0 * (wx0 * b00 + wx1 * b01) + wy1 * (wx0 * b10 + wx1 * b11);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v0[j] = ok ? v0[j] : 0.f;
            v1[j] = ok ? v1[j] : 0.f;
          }
          if (idx < HP * HP * 2) {
            bf16x8 hi, lo;
            split8(v0, v1, hi, lo);
            unsigned char* o = lds + (NA0 + c) * CH + ((pix / HP) * RS + pix % HP) * 32 + gsel * 16;
            *reinterpret_cast<bf16x8*>(o) = hi;
            *reinterpret_cast<bf16x8*>(o + PL) = lo;
          }
        }
      }
    }
    __syncthreads();
    // ---- every load of the next tile goes in flight
    {
      const int w2 = w + GS;
      const bool live = w2 < wend;
      const int ws = live ? w2 : 0;
      issue(ws / ntiles, ((ws % ntiles) / tiles_x) * T, ((ws % ntiles) % tiles_x) * T, live);
    }
    // ---- z tile = A (*) W, K = (tap pair, 16 channels) per MFMA
    f32x4 acc[MI][CO];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < CO; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks / 5, s = ks % 5;
      const unsigned char* cur = lds + c * CH;
      bf16x8 ah[MI], al[MI], bh[CO], bl[CO];
      if (!G::BREG) {
#pragma unroll
        for (int j = 0; j < CO; ++j) {
          bh[j] = ldsB[((ks * CO + j) * 2) * 64 + lane];
          bl[j] = ldsB[((ks * CO + j) * 2 + 1) * 64 + lane];
        }
      }
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        ah[m] = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff[s]);
        al[m] = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff[s] + PL);
      }
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int j = 0; j < CO; ++j) {
          if (G::BREG) {
            HPFG16_MFMA3(acc[m][j], ah[m], al[m], pbh[G::BREG ? ks : 0][j], pbl[G::BREG ? ks : 0][j])
          } else {
            HPFG16_MFMA3(acc[m][j], ah[m], al[m], bh[j], bl[j])
          }
        }
    }
    // ---- epilogue: acc holds D[channel = 4 (lane >> 4) + r][pixel = lane & 15] -> one 16-byte store per lane and tile; BatchNorm sums
#pragma unroll
    for (int j = 0; j < CO; ++j) {
      const int co = j * 16 + (lane >> 4) * 4;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int pxl = (wave * MI + m) * 16 + (lane & 15);
        const int pix = (n * H + ty0 + pxl / T) * W + tx0 + pxl % T;
        if (co < p.Cout) {
          const f32x4 v = acc[m][j] + bias[j];
          *reinterpret_cast<f32x4*>(p.out + pix * p.out_pstride + co) = v;
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
    __syncthreads();                                     // this tile's LDS reads are done: the next one may be written
  }
  if (p.stat_partials || p.stat_acc) {
    // per-lane sums -> over the 16 pixel lanes -> over the waves; row blockIdx.x of stat_partials ([rows][2][CoutPad]): sum(z), sum(z^2)
    // (or, stat_acc: integer atomics into the layer accumulator)
#pragma unroll
    for (int j = 0; j < CO; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[j][r], b = s2[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
        }
        s1[j][r] = a;
        s2[j][r] = b;
      }
    constexpr int BN = 16 * CO;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < CO; ++j) {
        const int cl = j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<f32x4*>(ldsf + (0 * NW + wave) * BN + cl) = s1[j];
        *reinterpret_cast<f32x4*>(ldsf + (1 * NW + wave) * BN + cl) = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cl = tid % BN;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) t += ldsf[(which * NW + k) * BN + cl];
      if (p.stat_acc && cl < p.Cout) hpfg_acc_add(p.stat_acc, p.CoutPad, (int)blockIdx.x & (p.stat_shards - 1), which, cl, t);
      if (p.stat_partials && cl < p.CoutPad) p.stat_partials[((long)blockIdx.x * 2 + which) * p.CoutPad + cl] = t;
    }
  }
}

// workgroups of the launch = rows of stat_partials
template <int CI, int CO, int AK, int NW, int WGS>
inline int thin_grid(const HpfgConvArgs& a) {
  const long nwork = (long)a.N * (a.H / T) * (a.W / T);
  int per_cu = 160 * 1024 / Geo<CI, CO, AK, NW>::LDS;
  if (per_cu > WGS) per_cu = WGS;
  if (per_cu < 1) per_cu = 1;
  const long cap = 256L * per_cu;
  const long rounds = (nwork + cap - 1) / cap;           // equal work per workgroup, all of them resident
  long grid = (nwork + rounds - 1) / rounds;
  grid = (grid + 7) / 8 * 8;                             // the same number of workgroups on every XCD
  return (int)(grid < cap ? grid : cap);
}

}  // namespace hpfg_thin
