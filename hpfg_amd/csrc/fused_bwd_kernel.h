// Fused backward pass of a thin 3x3 layer (16 / 32 / 64 input channels, 16 / 32 output channels, H and W multiples of 16) on the bf16
// matrix cores with split-precision operands: ONE kernel forms dZ from (dA, z) once per tile and uses it for BOTH products
//
//   dX[n,y,x,ci]      = sum_{tap,co} dZ[n, y+dy, x+dx, co] * Wd[tap][co][ci]            (dgrad: the input gradient, conv_bf16_kernel.h)
//   dW[co,ci,tap]     = sum_{n,y,x}  A[n, y+dy, x+dx, ci] * dZ[n,y,x,co]                (wgrad: the weight gradient, wgrad_bf16_kernel.h)
//
// -- the traffic model of SURVEY.md section 8(d): backward reads (dA, z) and the layer input once, where the separate dgrad and wgrad kernels
// each re-form dZ from (dA, z) in a pass of their own (measured: 2.9 GB of the step's 7.2 GB).
//
// Per 16 x 16 pixel tile a workgroup (4 waves) stages
//   * the dZ tile with a 1-pixel halo (18 x 18 pixels x Cout) through the DZ loader chain (stage.h), split into bf16 hi / lo planes in the
//     LDS as [16-channel chunk][hi|lo][pixel slot][16 ch] (32 bytes per pixel and plane);
//   * the INTERIOR 16 x 16 tile of the layer input A (BN + LeakyReLU + Dropout | MaxPool | upsample + concat applied on load), same layout.
//     No halo: with q = p + tap - 1,  dW[tap] = sum_q A[q] * dZ[q - tap + 1], and q - tap + 1 stays inside the dZ halo tile.
// dgrad reads pixel-major 16-byte fragments (8 consecutive channels of one pixel) exactly like conv_bf16x3_kernel; wgrad contracts over
// PIXELS and reads the same images through ds_read_b64_tr_b16 (a lane hands over the address of 4 channels of one pixel and receives 4
// pixels of one channel), so neither operand is ever transposed or staged twice.  With the 16 channels of a pixel adjacent both kinds of read
// are conflict-free: a 32-lane half of a transposing read covers 8 consecutive pixels x 32 bytes = one 256-byte bank row, and the 16-lane
// groups of a ds_read_b128 take the low 16 bytes of eight pixels and the high 16 bytes of the other eight (the conv kernels' layout, one
// plane per 8-channel group, makes one of the two reads 2-way conflicted whatever the padding: 40 % of the LDS cycles by the counters).
// With BWD the raw z of the input tile is parked in LDS as well (fp32, pixel stride padded by 32 bytes): the BatchNorm-backward epilogue of
// the layer below needs exactly those values, and reading them again from memory was a quarter of the kernel's bytes.
//
// The kernel is latency-, not FLOP-bound (a tile is ~100-200 KB of loads against ~2 us of MFMAs), so the loop is built around ONE exposed
// memory round trip per tile: the loads of tile i+1 are put in flight (into registers) before the weight-gradient MFMAs of tile i, and only
// converted / written to LDS once tile i's MFMAs are done.  The in-order vmcnt queue and the register file dictate the rest of the order:
// the dgrad phase -- the only one that needs global data (weight fragments; the first k-steps stay in registers for the whole kernel) --
// and its epilogue run BEFORE the prefetch is issued (their accumulators are dead by then), and the per-channel BatchNorm tables live in
// LDS (they do not depend on the tile), so converting the prefetched data never waits on a younger load.
// The workgroup walks a contiguous range of tiles of its XCD (halos shared through one L2); the weight-gradient accumulators stay in
// registers across all its tiles and are written once as a slab [tap][ci][co] (summed in a fixed order by hpfg_slab_reduce_multi:
// deterministic); the dgrad epilogue matches the conv kernels' (16-byte stores, optional second destination, optional BatchNorm-backward
// sums of the layer below).
#pragma once
#include <type_traits>
#include "conv_bf16_kernel.h"

namespace hpfg_fused {

using namespace hpfg_stage;
using hpfg_conv16::Cfg;
using hpfg_conv16::clampi;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int T = 16, HP = 18, RS = 18;                 // tile edge, halo tile edge, dZ row stride in slots
constexpr int DSLOT = (HP * HP + 15) / 16 * 16;         // 336
constexpr int DPL = DSLOT * 32;                         // bytes per hi / lo plane of one 16-channel chunk of the dZ tile: 32 bytes per pixel slot
constexpr int APL = T * T * 32;                         // ... of the A tile
constexpr int DCH = 2 * DPL, ACH = 2 * APL;             // one 16-channel chunk = (hi, lo)
constexpr int USH = T / 2 + 3, USW = T / 2 + 3;         // low-resolution source patch of an upsampled chunk of the (interior) input tile
constexpr int UBYTES = USH * USW * 16 * 4;              // fp32, [pixel][16 ch]

// NW waves per workgroup: staging pieces (8 channels of one pixel) per thread and 16-channel chunk, pixel tiles per wave in dgrad
template <int NW>
struct Split {
  static constexpr int NTH = 64 * NW;
  static constexpr int ND = (HP * HP * 2 + NTH - 1) / NTH, NA = T * T * 2 / NTH, MI = 16 / NW;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
};

template <int CI, int CO, int AK, int GK, int NW, bool BWD, int BMAX = 48>
struct Geo {
  static constexpr bool CATK = AK == HPFG_KIND_CAT;
  static constexpr int AK0 = CATK ? HPFG_KIND_BNACT : AK;          // loader kind of the chunks read through xa0 (concat: the skip half)
  static constexpr int NA0 = CATK ? CI / 2 : CI, NA1 = CATK ? CI / 2 : 0;
  static constexpr int TABD = GK == HPFG_KIND_DZ ? 5 * 16 * CO * 4 : 0;      // sc, sh, k1, k2, k3 rows of the dZ layer
  static constexpr int TABA = 2 * 16 * NA0 * 4;                              // sc, sh rows of the input's producer
  static constexpr int STAT = 2 * NW * 16 * CI * 4;
  static constexpr int KS = 5 * CO;                                          // dgrad k-steps (2 taps x 16 channels each)
  static constexpr bool BREG = KS * CI * 8 <= BMAX;                            // dgrad weight fragments: registers (few) or LDS
  static constexpr int BFR = BREG ? 0 : KS * CI * 2 * 1024;
  static constexpr bool ZLDS = BWD && AK == HPFG_KIND_BNACT;                 // raw z of the input tile for the epilogue of the layer below
  static constexpr int ZPS = 64 * CI + 32;                                   // bytes per pixel of that tile
  static constexpr int ZL = ZLDS ? T * T * ZPS : 0;
  static constexpr int OFF_STAT = CO * DCH + CI * ACH, OFF_TABD = OFF_STAT + STAT, OFF_TABA = OFF_TABD + TABD, OFF_B = OFF_TABA + TABA,
                       OFF_Z = OFF_B + BFR, OFF_U = OFF_Z + ZL;
  static constexpr int LDS = OFF_U + NA1 * UBYTES;
};

__device__ __forceinline__ bf16x8 tr8(const unsigned char* p0, const unsigned char* p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p1));
  return __builtin_bit_cast(bf16x8, (s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}));
}

// CI = CinPad / 16, CO = CoutPad / 16 of the layer; AK / GK = loader kinds of the layer input and of dZ; NW = waves per workgroup (8 for the
// wider layers: per-thread staging registers and accumulators halve, which is what lets a whole tile be prefetched); WGS = workgroups per CU
// to compile for; NODG = weight gradient only (the first layer: nothing to back-propagate into the network input); PFA = how many of the CI input chunks are prefetched a tile ahead together with dZ (the rest is requested when the tile
// starts); PFPOS = where in the tile loop that prefetch is issued
template <int CI, int CO, int AK, int GK, bool BWD, int NW, int WGS, int PFA, int PFPOS, int BMAX = 48, bool NODG = false>
__global__ __launch_bounds__(64 * NW, WGS) void fused_bwd_kernel(HpfgFusedBwdArgs p, int tiles_x, int tiles_y) {
  using C = Cfg<16, 16, 4, 1, CI, 9, 16>;               // (weight-fragment indexing of the dgrad side: CI output-channel tiles, K = 2 taps x 16 channels)
  using G = Geo<CI, CO, AK, GK, NW, BWD, BMAX>;
  static_assert(!BWD || AK == HPFG_KIND_BNACT, "the backward sums of the layer below need its raw output as this layer's input");
  using SP = Split<NW>;
  constexpr int AK0 = G::AK0, NA0 = G::NA0, NA1 = G::NA1, NTH = SP::NTH, ND = SP::ND, NA = SP::NA, MI = SP::MI;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  unsigned char* ldsD = lds;
  unsigned char* ldsA = lds + CO * DCH;
  float* ldsf = reinterpret_cast<float*>(lds + G::OFF_STAT);
  float* tabD = reinterpret_cast<float*>(lds + G::OFF_TABD);
  float* tabA = reinterpret_cast<float*>(lds + G::OFF_TABA);
  const bf16x8* ldsB = reinterpret_cast<const bf16x8*>(lds + G::OFF_B);
  unsigned char* ldsZ = lds + G::OFF_Z;
  float* ldsU = reinterpret_cast<float*>(lds + G::OFF_U);
  static_assert(NA1 == 0 || USH * USW * 2 <= NTH, "one source piece of the upsampled half per thread");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = p.d.H, W = p.d.W;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * p.d.N;
  const HpfgAct none = {};
  const HpfgAct& aD = p.d.a0;
  HpfgAct aS = p.xa0;                                    // (the skip half of a concat carries no dropout: model/unet.py:57 concatenates block outputs)
  if (G::CATK) aS.drop_p = 0.f;
  const ActCtx cxg = make_ctx(aD), cxa = make_ctx(aS);
#ifdef HPFG_TRACE      // diagnostics build (make TRACE=1, tools/trace_fused.py): thread 0 stamps s_memtime at the phase boundaries into d.bias
  const bool tr_on = (p.d.math & 0x2000) && tid == 0;
  unsigned long long* tr_buf = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.d.bias)) + 256 * (long)blockIdx.x;
  int tr_i = 0;
#endif
  HPFG_TR_REAL(12)
  HPFG_TR(1)
  const int gsel = tid & 1;                              // this thread's 8-channel group inside a 16-channel chunk (256 % 2 == 0)

  // ---- per-channel tables -> LDS, once (rows: scale, shift [, k1, k2, k3])
  if (GK == HPFG_KIND_DZ) hpfg_dz_rows_to_lds(aD, tabD, 16 * CO, 16 * CO, tid, NTH);      // (table rows, or k1 .. k3 from the backward sum accumulators)
  for (int i = tid; i < 2 * 16 * NA0 && AK0 != HPFG_KIND_PLAIN; i += NTH) {
    const int r = i / (16 * NA0), ch = i % (16 * NA0);
    tabA[i] = ch < aS.C ? aS.bn[aS.bn_coff + (r == 0 ? HPFG_BN_SCALE : HPFG_BN_SHIFT) * aS.bn_stride + ch] : 0.f;
  }

  // ---- wgrad work split: pair (i, j) = (input tile, output tile); with fewer than 4 pairs the 9 taps are dealt over the waves
  constexpr int NIJ = CI * CO;
  constexpr int PPW = NIJ >= NW ? NIJ / NW : 1;          // pairs per wave
  constexpr int TSTR = NIJ >= NW ? 1 : NW / NIJ;         // tap stride between the taps of one wave (= waves per pair)
  constexpr int NT = (9 + TSTR - 1) / TSTR;              // taps per wave (padded: a missing tap repeats tap 8 and is dropped)
  static_assert(NIJ == 1 || NIJ == 2 || NIJ == 4 || NIJ == 8, "pairs must split evenly over the waves");
  f32x4 accw[PPW][NT];
#pragma unroll
  for (int a = 0; a < PPW; ++a)
#pragma unroll
    for (int t = 0; t < NT; ++t) accw[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wt0 = NIJ >= NW ? 0 : wave / NIJ;

  // ---- dgrad fragment offsets (conv_bf16_kernel.h): pixel tile m of this wave, taps 2s / 2s+1 by k-group
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
  // ---- wgrad transposing-read offsets: k-group kg covers pixels x = 4 kg .. 4 kg + 3 of two tile rows; lane 4 q + pp of the group
  // addresses pixel q, channels 4 pp .. 4 pp + 3
  const int q4 = (lane & 15) >> 2, pp = lane & 3;
  const int trA = (4 * kg + q4) * 32 + pp * 8, trD = trA;

  // ---- dgrad weight fragments do not depend on the tile: they stay in registers for the whole kernel when they are few (one 16-channel
  // tile in and out), otherwise in LDS -- streamed from L2 per tile they stalled every other k-step (8 waves x 40 KB per tile, each k-step
  // pair shorter than the L2 latency)
  constexpr int KS = G::KS;
  constexpr int PB = G::BREG && !NODG ? KS : 0;
  const int ntn = p.d.CoutPad / 16;                      // dgrad output-channel tiles = CI
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.d.wpk);
  bf16x8 pbh[PB > 0 ? PB : 1][CI], pbl[PB > 0 ? PB : 1][CI];
#pragma unroll
  for (int k = 0; k < PB; ++k) hpfg_conv16::load_b<C>(pbh[k], pbl[k], wpk, k, ntn, 0, lane);
  if (!G::BREG && !NODG) {
    bf16x8* dst = reinterpret_cast<bf16x8*>(lds + G::OFF_B);
    for (int i = tid; i < KS * CI * 2 * 64; i += NTH) dst[i] = wpk[i];
  }

  f32x4 s1[CI], s2[CI];
#pragma unroll
  for (int j = 0; j < CI; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
  }
  ActCtx bcx = {};
  if (BWD) bcx = make_ctx(p.d.bwd_of);

  // ---- XCD-aware work mapping (conv_bf16_kernel.h): every XCD group walks a contiguous range of tiles
  const int nx = gridDim.x >= 8 ? 8 : 1;
  const int xg = (int)blockIdx.x % nx, xj = (int)blockIdx.x / nx;
  const int per_x = (nwork + nx - 1) / nx;
  const int wend = (xg + 1) * per_x < nwork ? (xg + 1) * per_x : nwork;
  const int GS = ((int)gridDim.x - xg + nx - 1) / nx;    // workgroups of this group = tile stride

  RawPiece<GK> rawD[CO][ND];
  RawPiece<AK0> rawA0[NA0][NA];
  f32x4 rawU[NA1 > 0 ? NA1 : 1][2];      // concat: the low-res source patch of the tile (one 8-channel piece per thread and upsampled chunk),
                                         // parked in LDS and blended from there (conv_bf16_kernel.h) -- 8 float4 loads per output piece otherwise

  // loads of one tile.  EARLY: dZ and the first PFA input chunks (a tile ahead); !EARLY: the remaining input chunks.  live == false: the
  // workgroup has no further tile -- every lane then reads pixel (0, 0, 0), one cache line, and nothing is made of it.
  auto issue = [&](auto early_tag, int n, int ty0, int tx0, bool live) {
    constexpr bool EARLY = decltype(early_tag)::value;
    if (EARLY) {
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const int c0 = c * 16 + gsel * 8;
        const bool chv = c0 < aD.C;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int idx = tid + i * NTH, pix = idx >> 1;
          const int gy = ty0 + pix / HP - 1, gx = tx0 + pix % HP - 1;
          const bool ok = live && idx < HP * HP * 2 && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
          issue_piece<GK>(rawD[c][i], aD, none, cxg, n, live ? clampi(gy, 0, H - 1) : 0, live ? clampi(gx, 0, W - 1) : 0, chv ? c0 : 0, ok);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NA0; ++c) {
      if ((c < PFA) != EARLY) continue;
      const int c0 = c * 16 + gsel * 8;
      const bool chv = c0 < aS.C;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int pix = (tid + i * NTH) >> 1;
        issue_piece<AK0>(rawA0[c][i], aS, none, cxa, n, live ? ty0 + pix / T : 0, live ? tx0 + pix % T : 0, chv ? c0 : 0, live && chv);
      }
    }
    if (NA1 > 0) {
      const int sy_base = hpfg_conv16::up_base(ty0, p.xa1.Hs), sx_base = hpfg_conv16::up_base(tx0, p.xa1.Ws);
      const int pix = tid >> 1;
      const int sy = live ? clampi(sy_base + pix / USW, 0, p.xa1.Hs - 1) : 0, sx = live ? clampi(sx_base + pix % USW, 0, p.xa1.Ws - 1) : 0;
#pragma unroll
      for (int c = 0; c < NA1; ++c) {
        if ((NA0 + c < PFA) != EARLY) continue;
        const int cu = c * 16 + gsel * 8;                // channel inside the upsampled tensor
        const int off = ((n * p.xa1.Hs + sy) * p.xa1.Ws + sx) * p.xa1.pstride + (cu < p.xa1.C ? cu : 0);
        rawU[c][0] = ld4(p.xa1.z, off);
        rawU[c][1] = ld4(p.xa1.z, off + 4);
      }
    }
  };

  int w = xg * per_x + xj;
  if (w < wend) issue(std::true_type{}, w / ntiles, ((w % ntiles) / tiles_x) * T, ((w % ntiles) % tiles_x) * T, true);
  __syncthreads();                                       // tables
  HPFG_TR(2)

  for (; w < wend; w += GS) {
    const int n = w / ntiles, ty0 = ((w % ntiles) / tiles_x) * T, tx0 = ((w % ntiles) % tiles_x) * T;
    if (PFA < CI) issue(std::false_type{}, n, ty0, tx0, true);
    HPFG_TR(3)
    // ---- prefetched (and just requested) raw data -> producer chain -> bf16 hi / lo -> LDS
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      const int c0 = c * 16 + gsel * 8;
      const bool chv = c0 < aD.C;
      Tab tg;
      if (GK == HPFG_KIND_DZ) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          tg.sc[h] = ld4(tabD, 0 * 16 * CO + c0 + 4 * h);
          tg.sh[h] = ld4(tabD, 1 * 16 * CO + c0 + 4 * h);
          tg.k1[h] = ld4(tabD, 2 * 16 * CO + c0 + 4 * h);
          tg.k2[h] = ld4(tabD, 3 * 16 * CO + c0 + 4 * h);
          tg.k3[h] = ld4(tabD, 4 * 16 * CO + c0 + 4 * h);
        }
      }
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int idx = tid + i * NTH, pix = idx >> 1;
        const int gy = ty0 + pix / HP - 1, gx = tx0 + pix % HP - 1;
        const bool ok = idx < HP * HP * 2 && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
        f32x4 v0, v1;
        finish_piece<GK>(v0, v1, rawD[c][i], tg, aD, none, cxg, n, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), chv ? c0 : 0, ok);
        if (idx < HP * HP * 2) {
          bf16x8 hi, lo;
          split8(v0, v1, hi, lo);
          unsigned char* o = ldsD + c * DCH + ((pix / HP) * RS + pix % HP) * 32 + gsel * 16;
          *reinterpret_cast<bf16x8*>(o) = hi;
          *reinterpret_cast<bf16x8*>(o + DPL) = lo;
        }
      }
    }
    // (prefetched chunks come first: the ones requested at the top of this iteration get the time of that conversion to arrive)
#pragma unroll
    for (int c = 0; c < NA0; ++c) {
      const int c0 = c * 16 + gsel * 8;
      const bool chv = c0 < aS.C;
      Tab ta;
#pragma unroll
      for (int h = 0; h < 2 && AK0 != HPFG_KIND_PLAIN; ++h) {
        ta.sc[h] = ld4(tabA, 0 * 16 * NA0 + c0 + 4 * h);
        ta.sh[h] = ld4(tabA, 1 * 16 * NA0 + c0 + 4 * h);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int pix = (tid + i * NTH) >> 1;
        f32x4 v0, v1;
        finish_piece<AK0>(v0, v1, rawA0[c][i], ta, aS, none, cxa, n, ty0 + pix / T, tx0 + pix % T, chv ? c0 : 0, chv);
        if (G::ZLDS) {
          *reinterpret_cast<f32x4*>(ldsZ + pix * G::ZPS + c0 * 4) = rawA0[c][i].v[0];
          *reinterpret_cast<f32x4*>(ldsZ + pix * G::ZPS + c0 * 4 + 16) = rawA0[c][i].v[1];
        }
        bf16x8 hi, lo;
        split8(v0, v1, hi, lo);
        unsigned char* o = ldsA + c * ACH + pix * 32 + gsel * 16;
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + APL) = lo;
      }
    }
    if (NA1 > 0) {
      if (tid < USH * USW * 2) {
#pragma unroll
        for (int c = 0; c < NA1; ++c) {
          float* d = ldsU + c * (UBYTES / 4) + (tid >> 1) * 16 + gsel * 8;
          *reinterpret_cast<f32x4*>(d) = rawU[c][0];
          *reinterpret_cast<f32x4*>(d + 4) = rawU[c][1];
        }
      }
      __syncthreads();
      const int sy_base = hpfg_conv16::up_base(ty0, p.xa1.Hs), sx_base = hpfg_conv16::up_base(tx0, p.xa1.Ws);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int pix = (tid + i * NTH) >> 1;
        int y0, y1, x0, x1;
        float wy1, wx1;
        up_coord(ty0 + pix / T, p.xa1.Hs, y0, y1, wy1);
        up_coord(tx0 + pix % T, p.xa1.Ws, x0, x1, wx1);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const int o0 = (x0 - sx_base) * 16, o1 = (x1 - sx_base) * 16;
#pragma unroll
        for (int c = 0; c < NA1; ++c) {
          const bool chv = c * 16 + gsel * 8 < p.xa1.C;
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
            v0[j] = chv ? v0[j] : 0.f;
            v1[j] = chv ? v1[j] : 0.f;
          }
          bf16x8 hi, lo;
          split8(v0, v1, hi, lo);
          unsigned char* o = ldsA + (NA0 + c) * ACH + pix * 32 + gsel * 16;
          *reinterpret_cast<bf16x8*>(o) = hi;
          *reinterpret_cast<bf16x8*>(o + APL) = lo;
        }
      }
    }
    HPFG_TR(4)
    __syncthreads();
    HPFG_TR(5)

    // ---- every load of the next tile goes in flight: here (PFPOS 0: a whole dgrad + epilogue + wgrad of cover, costs the registers through
    // the dgrad phase) or after the epilogue (PFPOS 1)
    const int w2 = w + GS;
    const bool live2 = w2 < wend;
    const int ws2 = live2 ? w2 : 0;
    if (PFPOS == 0) issue(std::true_type{}, ws2 / ntiles, ((ws2 % ntiles) / tiles_x) * T, ((ws2 % ntiles) % tiles_x) * T, live2);
    if constexpr (!NODG) {
    // ---- dgrad: dX tile = dZ (*) Wd, K = (tap pair, 16 channels) per MFMA
    f32x4 acc[MI][CI];
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < CI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = ks / 5, s = ks % 5;
        const unsigned char* cur = ldsD + c * DCH;
        bf16x8 ah[MI], al[MI], bh[CI], bl[CI];
        if (!G::BREG) {
#pragma unroll
          for (int j = 0; j < CI; ++j) {
            bh[j] = ldsB[((ks * CI + j) * 2) * 64 + lane];
            bl[j] = ldsB[((ks * CI + j) * 2 + 1) * 64 + lane];
          }
        }
#pragma unroll
        for (int m = 0; m < MI; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff[s]);
          al[m] = *reinterpret_cast<const bf16x8*>(cur + aoff[m] + toff[s] + DPL);
        }
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
          for (int j = 0; j < CI; ++j) {
            if (G::BREG) {
              HPFG16_MFMA3(acc[m][j], ah[m], al[m], pbh[G::BREG ? ks : 0][j], pbl[G::BREG ? ks : 0][j])
            } else {
              HPFG16_MFMA3(acc[m][j], ah[m], al[m], bh[j], bl[j])
            }
          }
      }
    }
    HPFG_TR(6)
    // ---- dgrad epilogue: acc holds D[channel = 4 (lane >> 4) + r][pixel = lane & 15] -> one 16-byte store per lane and tile;
    // with BWD the backward sums of the layer below: g = dX * dropout * LeakyReLU'(bn(z)), sum(g), sum(g * z) (finished in the flush)
#pragma unroll
    for (int j = 0; j < CI; ++j) {
      const int co = j * 16 + (lane >> 4) * 4;
      f32x4 tsc = {}, tsh = {};
      if (BWD) {      // the layer below is the producer of this layer's input: its scale / shift rows are in LDS already
        tsc = ld4(tabA, 0 * 16 * NA0 + co);
        tsh = ld4(tabA, 1 * 16 * NA0 + co);
      }
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int pxl = (wave * MI + m) * 16 + (lane & 15);
        const int pix = (n * H + ty0 + pxl / T) * W + tx0 + pxl % T;
        const f32x4 v = acc[m][j];
        float* o = (p.d.out_split && co >= p.d.out_split) ? p.d.out2 + pix * p.d.out2_pstride + (co - p.d.out_split) : p.d.out + pix * p.d.out_pstride + co;
        *reinterpret_cast<f32x4*>(o) = v;
        if (BWD) {
          const f32x4 z = *reinterpret_cast<const f32x4*>(ldsZ + pxl * G::ZPS + co * 4);
          uint32_t km = 0xFu;
          if (p.d.bwd_of.drop_p > 0.f) km = keep4(p.d.bwd_of, bcx, (uint32_t)(pix * p.d.bwd_of.C + co));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float gg = (km >> r) & 1u ? v[r] * bcx.inv_keep : 0.f;
            gg = z[r] * tsc[r] + tsh[r] > 0.f ? gg : HPFG_LEAKY * gg;
            s1[j][r] += gg;
            s2[j][r] += gg * z[r];
          }
        }
      }
    }
    }
    HPFG_TR(7)
    if (PFPOS == 1) issue(std::true_type{}, ws2 / ntiles, ((ws2 % ntiles) / tiles_x) * T, ((ws2 % ntiles) % tiles_x) * T, live2);
    HPFG_TR(8)
    // ---- wgrad: dW[tap][ci][co] += A^T dZ(shifted), K = 32 pixels (two tile rows) per MFMA
#pragma unroll
    for (int a = 0; a < PPW; ++a) {
      const int pair = NIJ >= NW ? wave + NW * a : wave % NIJ;
      const int i = pair % CI, j = pair / CI;
      const unsigned char* Ai = ldsA + i * ACH + trA;
      const unsigned char* Dj = ldsD + j * DCH + trD;
#pragma unroll
      for (int ks = 0; ks < T / 2; ++ks) {
        const unsigned char* ap = Ai + (2 * ks * T) * 32;
        const bf16x8 ah = tr8(ap, ap + T * 32), al = tr8(ap + APL, ap + APL + T * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int tap = wt0 + t * TSTR < 9 ? wt0 + t * TSTR : 8;
          const int ky = tap / 3, kx = tap - 3 * ky;
          const unsigned char* dp = Dj + ((2 * ks - ky + 2) * RS + (2 - kx)) * 32;
          const bf16x8 gh = tr8(dp, dp + RS * 32), gl_ = tr8(dp + DPL, dp + DPL + RS * 32);
          accw[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh, accw[a][t], 0, 0, 0);
          accw[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl_, accw[a][t], 0, 0, 0);
          accw[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh, accw[a][t], 0, 0, 0);
        }
      }
    }
    HPFG_TR(9)
    __syncthreads();                                     // this tile's LDS reads are done: the next one may be written
    HPFG_TR(10)
  }
  if (BWD) {
    // per-lane sums -> over the 16 pixel lanes -> over the waves; row blockIdx.x of stat_partials ([rows][2][Cin]) receives sum(g) and
    // sum(g * xhat) = rstd * (sum(g * z) - mean * sum(g))   (conv16_flush_stats)
#pragma unroll
    for (int j = 0; j < CI; ++j)
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
    constexpr int BN = 16 * CI;
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < CI; ++j) {
        const int cl = j * 16 + (lane >> 4) * 4;
        *reinterpret_cast<f32x4*>(ldsf + (0 * NW + wave) * BN + cl) = s1[j];
        *reinterpret_cast<f32x4*>(ldsf + (1 * NW + wave) * BN + cl) = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cl = tid % BN;
      float t = 0.f, sg = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        t += ldsf[(which * NW + k) * BN + cl];
        sg += ldsf[k * BN + cl];
      }
      if (which == 1) {
        const float* tb = p.d.bwd_of.bn + p.d.bwd_of.bn_coff + cl;
        t = tb[HPFG_BN_RSTD * p.d.bwd_of.bn_stride] * (t - tb[HPFG_BN_MEAN * p.d.bwd_of.bn_stride] * sg);
      }
      if (p.d.stat_acc) hpfg_acc_add(p.d.stat_acc, p.d.CoutPad, (int)blockIdx.x & (p.d.stat_shards - 1), which, cl, t);
      if (p.d.stat_partials) p.d.stat_partials[((long)blockIdx.x * 2 + which) * p.d.CoutPad + cl] = t;
    }
  }
  HPFG_TR(11)
  HPFG_TR_REAL(13)
  // ---- slab[blockIdx.x][tap][ci][co]; accumulator rows = ci (4 (lane >> 4) + r), column = co (lane & 15)
  float* slab = p.slab + (long)blockIdx.x * 9 * p.CinPad * p.CoutPad;
#pragma unroll
  for (int a = 0; a < PPW; ++a) {
    const int pair = NIJ >= NW ? wave + NW * a : wave % NIJ;
    const int i = pair % CI, j = pair / CI;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int tap = wt0 + t * TSTR;
      if (tap < 9) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = i * 16 + (lane >> 4) * 4 + r, co = j * 16 + (lane & 15);
          slab[((long)tap * p.CinPad + ci) * p.CoutPad + co] = accw[a][t][r];
        }
      }
    }
  }
}

// workgroups of the launch = slabs = rows of the BatchNorm-backward partial sums
template <int CI, int CO, int AK, int GK, int NW, int WGS, int BMAX = 48>
inline int fused_grid(const HpfgFusedBwdArgs& a) {
  const long nwork = (long)a.d.N * (a.d.H / T) * (a.d.W / T);
  int per_cu = 160 * 1024 / Geo<CI, CO, AK, GK, NW, AK == HPFG_KIND_BNACT, BMAX>::LDS;      // (the same grid with and without the backward sums)
  if (per_cu > WGS) per_cu = WGS;
  if (per_cu < 1) per_cu = 1;
  const long cap = 256L * per_cu;
  const long rounds = (nwork + cap - 1) / cap;           // equal work per workgroup, all of them resident
  long grid = (nwork + rounds - 1) / rounds;
  grid = (grid + 7) / 8 * 8;                             // the same number of workgroups on every XCD
  return (int)(grid < cap ? grid : cap);
}

}  // namespace hpfg_fused
