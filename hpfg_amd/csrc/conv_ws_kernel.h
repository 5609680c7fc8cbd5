// Warp-specialised form of the bf16x3 3x3 convolution (same tiles, LDS images, MFMA mapping and epilogue as conv_bf16_kernel.h).
//
// Why: with every wave doing "request tile -> MFMA -> convert tile -> barrier" in turn, a workgroup's memory latency (~2 us under
// load), loader VALU work and MFMA time add up, and the two or three workgroups a CU can hold run those phases in lockstep.
// Here a workgroup is 8 waves, two per SIMD:
//   waves 0-3  MMA waves     LDS fragment reads + MFMAs + weight-fragment ring + epilogue (stores, BatchNorm partial sums).
//                            Their vector-memory queue holds only weight loads, so the chunk-deep ring never waits on a tile load.
//   waves 4-7  loader waves  request the raw input pieces PD chunk positions ahead (registers only hold raw data, so the ring
//                            can be that deep), apply the producer chain (BN + LeakyReLU + Dropout | MaxPool | bilinear + concat |
//                            dZ), split to bf16 hi/lo and park the result in the LDS image the MMA waves read next.
// One s_barrier per chunk position hands the finished image over and releases the consumed one (two images, as before).  The
// loaders' VALU work runs on the same SIMDs as, and concurrently with, the MMA waves' matrix instructions.  One workgroup per
// CU (8 waves x up to 256 VGPRs).
//
// A workgroup's work is the stream of (tile, input-channel chunk) positions of its tiles, in order; "position k" below.
// The producer tables (BatchNorm scale/shift, dZ k1/k2/k3) live in LDS for the whole kernel instead of per-chunk registers.
#pragma once
#include "conv_bf16_kernel.h"

namespace hpfg_conv16 {

// loader prefetch depth in chunk positions: 3 for the two-load kinds, 2 while the raw ring stays <= 128 VGPRs, else 1
template <class C, int KIND>
constexpr int ws_pd() {
  return RawCount<KIND>::N <= 2 ? 3 : (RawCount<KIND>::N * C::NLD <= 16 ? 2 : 1);
}

constexpr int WS_CT = 512;      // table columns in LDS (input channels, host checks cin <= WS_CT)

template <int KIND>
struct WsTabRows { static constexpr int N = KIND == HPFG_KIND_PLAIN ? 0 : (KIND == HPFG_KIND_DZ ? 5 : 2); };

template <int KIND>
__device__ __forceinline__ void ws_tab_from_lds(Tab& t, const float* ldsT, int c0) {
  if (KIND == HPFG_KIND_PLAIN) return;
  t.sc[0] = *reinterpret_cast<const f32x4*>(ldsT + 0 * WS_CT + c0);
  t.sc[1] = *reinterpret_cast<const f32x4*>(ldsT + 0 * WS_CT + c0 + 4);
  t.sh[0] = *reinterpret_cast<const f32x4*>(ldsT + 1 * WS_CT + c0);
  t.sh[1] = *reinterpret_cast<const f32x4*>(ldsT + 1 * WS_CT + c0 + 4);
  if (KIND == HPFG_KIND_DZ) {
    t.k1[0] = *reinterpret_cast<const f32x4*>(ldsT + 2 * WS_CT + c0);
    t.k1[1] = *reinterpret_cast<const f32x4*>(ldsT + 2 * WS_CT + c0 + 4);
    t.k2[0] = *reinterpret_cast<const f32x4*>(ldsT + 3 * WS_CT + c0);
    t.k2[1] = *reinterpret_cast<const f32x4*>(ldsT + 3 * WS_CT + c0 + 4);
    t.k3[0] = *reinterpret_cast<const f32x4*>(ldsT + 4 * WS_CT + c0);
    t.k3[1] = *reinterpret_cast<const f32x4*>(ldsT + 4 * WS_CT + c0 + 4);
  }
}

// position iterator of one workgroup: tile (n, tyi, txi) in units of tiles, chunk ch, linear tile index w
struct WsPos {
  int n, tyi, txi, ch, w;
};

template <class C, int KIND>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 9, "warp-specialised kernel is the 3x3 path");
  constexpr int NR = RawCount<KIND>::N;
  constexpr int PD = ws_pd<C, KIND>();
  constexpr int U = PD == 3 ? 6 : 2;                  // steps per unrolled round: a multiple of PD and of the two LDS images
  constexpr int TROWS = WsTabRows<KIND>::N;
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * C::BUF_BYTES + STAT_BYTES + (TROWS ? TROWS : 1) * WS_CT * 4];
  float* ldsf = reinterpret_cast<float*>(lds + 2 * C::BUF_BYTES);
  float* ldsT = reinterpret_cast<float*>(lds + 2 * C::BUF_BYTES + STAT_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mma = wave < 4;
#ifdef HPFG_TRACE
  // stamps of MMA wave 0 in entries [0,128), of loader wave 4 in [128,256) of this workgroup's 256-entry record
  const bool tr_on = (p.math & 0x2000) && (tid == 0 || tid == 256);
  unsigned long long* tr_buf = reinterpret_cast<unsigned long long*>(p.stat_partials) + 256 * ((long)blockIdx.y * gridDim.x + blockIdx.x) + (tid == 256 ? 128 : 0);
  int tr_i = 0;
#undef HPFG_TR
#undef HPFG_TR_REAL
#define HPFG_TR(ID)                                                                                        \
  if (tr_on && tr_i < 127) {                                                                               \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
  }
#define HPFG_TR_REAL(ID)                                                                                   \
  if (tr_on && tr_i < 127) {                                                                               \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffull); \
  }
#endif
  HPFG_TR_REAL(11)
  HPFG_TR(1)
  const int cb = blockIdx.y;
  const int H = p.H, W = p.W;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * p.N;
  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + C::KC - 1) / C::KC;

  // same XCD-aware tile distribution as conv_bf16x3_kernel
  const int nx = gridDim.x >= 8 ? 8 : 1;
  const int xg = (int)blockIdx.x % nx, xj = (int)blockIdx.x / nx;
  const int per_x = (nwork + nx - 1) / nx;
  const int wend = (xg + 1) * per_x < nwork ? (xg + 1) * per_x : nwork;
  const int G = ((int)gridDim.x - xg + nx - 1) / nx;
  const int w0 = xg * per_x + xj;
  if (w0 >= wend) {   // no tile for this workgroup: its BatchNorm partial row must still be defined
    if (p.stat_partials && tid < 2 * C::BN) {
      const int co = cb * C::BN + tid % C::BN;
      if (co < p.CoutPad) p.stat_partials[((long)blockIdx.x * 2 + tid / C::BN) * p.CoutPad + co] = 0.f;
    }
    return;
  }
  const int nitems = (wend - w0 + G - 1) / G;
  const int T = nitems * nchunks;                      // chunk positions of this workgroup
  const int Tpad = (T + U - 1) / U * U;
  const int gn = G / ntiles, gty = (G % ntiles) / tiles_x, gtx = (G % ntiles) % tiles_x;

  // producer tables -> LDS (all 512 threads)
  if (TROWS) {
    const int ctab = KIND == HPFG_KIND_CAT ? p.a0.C : cin_total;     // concat: only the skip half carries BatchNorm
    const float* b = p.a0.bn + p.a0.bn_coff;
    for (int i = tid; i < TROWS * WS_CT; i += 512) {
      const int r = i / WS_CT, c = i % WS_CT;
      const int row = r == 0 ? HPFG_BN_SCALE : (r == 1 ? HPFG_BN_SHIFT : (r == 2 ? HPFG_BN_K1 : (r == 3 ? HPFG_BN_K2 : HPFG_BN_K3)));
      ldsT[i] = c < ctab ? b[row * p.a0.bn_stride + c] : 0.f;
    }
  }
  __syncthreads();
  HPFG_TR(2)

  if (!mma) {
    // ================================================================ loader waves
    const int lt = tid - 256;
    const ActCtx cx0 = make_ctx(p.a0);
    Piece pc[C::NLD];
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) pc[i] = make_piece<C>(lt, i);
    const int g8 = (lt % C::NG) * 8;
    RawPiece<KIND> raw[PD][C::NLD];
    int mn[PD], mty[PD], mtx[PD], mc[PD];
    bool mv[PD];
    WsPos it;
    it.w = w0;
    it.n = w0 / ntiles;
    it.tyi = (w0 % ntiles) / tiles_x;
    it.txi = (w0 % ntiles) % tiles_x;
    it.ch = 0;
    int ipos = 0;                                       // position index of `it`

#define HPFG_WS_ISSUE(SLOT)                                                                                      \
  {                                                                                                              \
    const int c0 = it.ch * C::KC + g8;                                                                           \
    mn[SLOT] = it.n;                                                                                             \
    mty[SLOT] = it.tyi * C::TH;                                                                                  \
    mtx[SLOT] = it.txi * C::TW;                                                                                  \
    mc[SLOT] = c0 < cin_total ? c0 : 0;                                                                          \
    mv[SLOT] = ipos < T && c0 < cin_total;                                                                       \
    _Pragma("unroll") for (int i = 0; i < C::NLD; ++i) {                                                         \
      const int gy = mty[SLOT] + pc[i].ly, gx = mtx[SLOT] + pc[i].lx;                                            \
      const bool ok = pc[i].ok && mv[SLOT] && gy >= 0 && gy < H && gx >= 0 && gx < W;                            \
      issue_piece<KIND>(raw[SLOT][i], p.a0, p.a1, cx0, mn[SLOT], clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), mc[SLOT], ok); \
    }                                                                                                            \
  }
#define HPFG_WS_FINISH(SLOT, BUF)                                                                                \
  {                                                                                                              \
    Tab tab;                                                                                                     \
    ws_tab_from_lds<KIND>(tab, ldsT, mc[SLOT]);                                                                  \
    _Pragma("unroll") for (int i = 0; i < C::NLD; ++i) {                                                         \
      const int gy = mty[SLOT] + pc[i].ly, gx = mtx[SLOT] + pc[i].lx;                                            \
      const bool ok = pc[i].ok && mv[SLOT] && gy >= 0 && gy < H && gx >= 0 && gx < W;                            \
      f32x4 v0, v1;                                                                                              \
      finish_piece<KIND>(v0, v1, raw[SLOT][i], tab, p.a0, p.a1, cx0, mn[SLOT], clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), mc[SLOT], ok); \
      store_piece<C, KIND>(lds + (BUF) * C::BUF_BYTES, pc[i], v0, v1);                                                 \
    }                                                                                                            \
  }
    // advance `it` by one position; past the end it parks on the last position (reloads of valid addresses, results unused)
#define HPFG_WS_ADVANCE()                                                                                        \
  {                                                                                                              \
    ++ipos;                                                                                                      \
    if (ipos < T) {                                                                                              \
      if (++it.ch == nchunks) {                                                                                  \
        it.ch = 0;                                                                                               \
        it.w += G;                                                                                               \
        it.txi += gtx;                                                                                           \
        it.tyi += gty;                                                                                           \
        it.n += gn;                                                                                              \
        if (it.txi >= tiles_x) {                                                                                 \
          it.txi -= tiles_x;                                                                                     \
          ++it.tyi;                                                                                              \
        }                                                                                                        \
        if (it.tyi >= tiles_y) {                                                                                 \
          it.tyi -= tiles_y;                                                                                     \
          ++it.n;                                                                                                \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }

    // prologue: position 0 goes straight into image 0, positions 1..PD into the ring (position j lives in slot j % PD)
    HPFG_WS_ISSUE(0)
#pragma unroll
    for (int j = 1; j < PD; ++j) {
      HPFG_WS_ADVANCE()
      HPFG_WS_ISSUE(j)
    }
    HPFG_TR(3)
    HPFG_WS_FINISH(0, 0)
    HPFG_WS_ADVANCE()
    HPFG_WS_ISSUE(0)       // position PD reuses slot 0
    HPFG_TR(4)
    __syncthreads();
    HPFG_TR(5)
    for (int k0 = 0; k0 < Tpad; k0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // step k = k0 + u: park position k + 1 (slot (k+1) % PD) in image (k+1) & 1, then reuse its slot for position k + 1 + PD
        if (k0 + u + 1 < T) HPFG_WS_FINISH((u + 1) % PD, (u + 1) & 1)
        HPFG_TR(6)
        HPFG_WS_ADVANCE()
        if (ipos < T) HPFG_WS_ISSUE((u + 1) % PD)
        HPFG_TR(7)
        __syncthreads();
        HPFG_TR(8)
      }
    }
    if (p.stat_partials && !(p.math & 0x1000)) __syncthreads();     // pairs with the barrier inside the MMA waves' statistics flush
    HPFG_TR(10)
    HPFG_TR_REAL(12)
#undef HPFG_WS_ISSUE
#undef HPFG_WS_FINISH
#undef HPFG_WS_ADVANCE
    return;
  }

  // ================================================================== MMA waves
  const int wm = wave % C::WM, wn = wave / C::WM;
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
    const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (gl * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }
  int toff[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    int tap = C::KC == 32 ? s : 2 * s + (kg >> 1);
    tap = tap > 8 ? 8 : tap;
    toff[s] = ((tap / 3) * C::RS + (tap % 3)) * 16;
  }
  const int ntn = p.CoutPad / 16;
  const int nt0 = (cb * C::WN + wn) * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);
  f32x4 s1[C::NI], s2[C::NI], bias[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
  }
  conv16_load_bias<C>(p, bias, lane, nt0);
  // weight-fragment ring, one chunk deep where the registers allow (see conv_bf16x3_kernel); only these loads are in this wave's queue
  constexpr int BR = C::KSTEPS == 5 ? 5 : (C::NI == 1 ? 9 : 3), BD = BR - 1;
  static_assert(C::KSTEPS % BR == 0, "ring size must divide the k-steps of a chunk");
  const int kperiod = nchunks * C::KSTEPS;
  bf16x8 bh[BR][C::NI], bl[BR][C::NI];
#pragma unroll
  for (int d = 0; d < BD; ++d) load_b<C>(bh[d], bl[d], wpk, d % kperiod, ntn, nt0, lane);
  int n = w0 / ntiles, tyi = (w0 % ntiles) / tiles_x, txi = (w0 % ntiles) % tiles_x;
  int ch = 0;
  HPFG_TR(4)
  __syncthreads();       // image 0 is ready
  HPFG_TR(5)
  for (int k0 = 0; k0 < Tpad; k0 += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u;
      const unsigned char* cur = lds + (u & 1) * C::BUF_BYTES;
      if (k < T) {
        constexpr int Q = C::KSTEPS * C::MI;
        constexpr int AD0 = C::NI == 1 ? 4 : 2;
        constexpr int AD = AD0 < Q ? AD0 : Q;
        constexpr int AR = AD + 1;
        bf16x8 ah[AR], al[AR];
#pragma unroll
        for (int q = 0; q < AD; ++q) {
          ah[q % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q % C::MI] + toff[q / C::MI]);
          al[q % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q % C::MI] + toff[q / C::MI] + C::PLANE);
        }
        __builtin_amdgcn_sched_barrier(0x216);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int s = q / C::MI, m = q % C::MI;
          if (q + AD < Q) {
            const int q2 = q + AD;
            ah[q2 % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q2 % C::MI] + toff[q2 / C::MI]);
            al[q2 % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q2 % C::MI] + toff[q2 / C::MI] + C::PLANE);
          }
          if (m == 0) {
            int ksn = ch * C::KSTEPS + s + BD;
            ksn = ksn >= kperiod ? ksn - kperiod : ksn;
            load_b<C>(bh[(s + BD) % BR], bl[(s + BD) % BR], wpk, ksn, ntn, nt0, lane);
          }
#pragma unroll
          for (int j = 0; j < C::NI; ++j) { HPFG16_MFMA3(acc[m][j], ah[q % AR], al[q % AR], bh[s % BR][j], bl[s % BR][j]) }
          __builtin_amdgcn_sched_barrier(0x216);   // VALU, SALU, VMEM and DS writes may cross; DS reads and MFMAs may not
        }
      }
      HPFG_TR(6)
      __syncthreads();     // image (k+1) & 1 is ready, image k & 1 may be overwritten
      HPFG_TR(8)
      if (k < T) {
        if (++ch == nchunks) {        // tile finished: epilogue from registers while the loaders work on
          ch = 0;
          conv16_store_tile<C>(p, acc, s1, s2, bias, lane, wm, nt0, n, tyi * C::TH, txi * C::TW);
#pragma unroll
          for (int m = 0; m < C::MI; ++m)
#pragma unroll
            for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          txi += gtx;
          tyi += gty;
          n += gn;
          if (txi >= tiles_x) {
            txi -= tiles_x;
            ++tyi;
          }
          if (tyi >= tiles_y) {
            tyi -= tiles_y;
            ++n;
          }
          HPFG_TR(9)
        }
      }
    }
  }
  conv16_flush_stats<C>(p, s1, s2, ldsf, tid, lane, wm, wn, cb, (int)blockIdx.x);
  HPFG_TR(10)
  HPFG_TR_REAL(12)
}

template <class C, int KIND>
int ws_grid(const HpfgConvArgs& a) {
  const int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  const long nwork = (long)tx * ty * a.N;
  const long gy = a.CoutPad / C::BN;
  long cap = 256L / (gy > 0 ? gy : 1);      // one workgroup per CU, all resident
  if (cap < 1) cap = 1;
  const long rounds = (nwork + cap - 1) / cap;
  return (int)((nwork + rounds - 1) / rounds);
}

// Which layer classes run warp-specialised (HPFG_CONV_WS bit mask, for A/B runs): 1 channel-rich small tiles (KC = 32),
// 2 thin 16x16 tiles behind a two-load loader (PLAIN / BNACT), 4 thin tiles behind the dZ loader, 8 thin tiles behind the
// pool / concat loaders.
inline int ws_mask() {
  static const int m = [] {
    const char* e = getenv("HPFG_CONV_WS");
    return e ? atoi(e) : 0;      // experimental: per-kernel wins on the channel-rich layers do not shorten the step
  }();
  return m;
}
template <class C, int KIND>
inline bool ws_enabled() {
  if (KIND == HPFG_KIND_PLANES) return false;      // the experiment predates the PLANES source (its loader waves keep BatchNorm tables in LDS)
  const int cls = C::KC == 32 ? 1 : (RawCount<KIND>::N <= 2 ? 2 : (RawCount<KIND>::N == 4 ? 4 : 8));
  return (ws_mask() & cls) != 0;
}

}  // namespace hpfg_conv16
