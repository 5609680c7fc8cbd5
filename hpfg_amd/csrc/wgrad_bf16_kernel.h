// Weight gradient of the 3x3 convolutions on the bf16 matrix cores with split-precision (bf16x3) operands.
//
//   dW[co,ci,tap] = sum_{n,y,x} A[n, y+dy, x+dx, ci] * dZ[n,y,x,co]        (GEMM: M = ci, N = co, K = pixels)
//
// The contraction index is the PIXEL, while both operands arrive channel-contiguous (NHWC).  gfx950's transposed LDS read
// ds_read_b64_tr_b16 removes the transpose: a 16-lane group hands over the addresses of 4 rows (pixels) x 16 columns (channels)
// of a pixel-major bf16 tile and every lane receives ONE channel of those 4 pixels -- exactly the k-contiguous fragment
// v_mfma_f32_16x16x32_bf16 wants.  So the tiles are staged pixel-major ([pixel][16 ch], 32 B per pixel, hi and lo planes), a tap
// shift is just a different row address, and no operand is ever transposed in registers or LDS.
//   k index of one MFMA (32 pixels = 2 tile rows x 16): k-group g (lanes 16g..16g+15) holds pixels x = 4g..4g+3 of row 2*ks
//   (elements 0..3) and of row 2*ks+1 (elements 4..7), for A (shifted by the tap) and dZ alike; the two 16-lane groups of a
//   32-lane half read 256 contiguous bytes -> bank-conflict free.
// Workgroup = 4 waves (one per SIMD, two workgroups per CU) for NI input-channel tiles x NJ output-channel tiles of 16:
// wave w owns (input tile w % NI, output tile (w / NI) % NJ) and the taps t with t % TSTR == w / (NI * NJ), TSTR = 4 / (NI * NJ):
// 9 / 5+4 / 3+2+2+2 taps per wave (a wave with fewer real taps repeats tap 8 and drops the result: no divergence in the MFMA
// phase).  Shapes: 2 x 2 where both channel counts allow (every staged A / dZ tile is then shared by two waves: the loaders'
// VALU work, which bounds this kernel, is re-done Cout/32 + Cin/32 times instead of Cout/64 + Cin/16 times), else 1 x {4, 2, 1}.
// A workgroup walks a strided list of (image, 8x16 tile) work items with the accumulators in registers and writes one slab;
// wgrad.hip's slab reduction sums the slabs in a fixed order.  Products are hi*hi + hi*lo + lo*hi (fp32 accumulate) as in
// conv_bf16_kernel.h.
#pragma once
#include "stage.h"

namespace hpfg_wg16 {

using namespace hpfg_stage;
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;
constexpr int G_PLANE = TH * TW * 32;          // bytes per hi / lo plane of one output-channel tile
constexpr int NTHR = 256;

template <int TAPS>
struct Geo {                                   // A tile: 3x3 with a 1-pixel halo, 1x1 without
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HP = TH + 2 * HALO, WP = TW + 2 * HALO;
  static constexpr int A_SLOTS = HP * WP;      // pixel slots of 32 B (180 / 128)
  static constexpr int A_PLANE = A_SLOTS * 32; // bytes per hi / lo plane of one input-channel tile
};

template <int NI, int NJ, int TAPS>
struct Lds {
  static constexpr int A_BYTES = NI * 2 * Geo<TAPS>::A_PLANE;
  static constexpr int G_BYTES = NJ * 2 * G_PLANE;
  static constexpr int BYTES = A_BYTES + G_BYTES;
};

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* row0, const unsigned char* row1) {
  // row0/row1: this lane's addresses inside the two 4-pixel blocks (see header); result = 8 k-values of this lane's channel
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(row0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(row1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// diagnostics build (make TRACE=1): math bit 0x2000 -> thread 0 of every workgroup stamps s_memtime into slab + 256 * block (u64)
#ifdef HPFG_TRACE
#define HPFG_WTR(ID)                                                                                        \
  if (tr_on && tr_i < 255) {                                                                                \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
  }
#define HPFG_WTR_REAL(ID)                                                                                   \
  if (tr_on && tr_i < 255) {                                                                                \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffull); \
  }
#else
#define HPFG_WTR(ID)
#define HPFG_WTR_REAL(ID)
#endif

// One staging piece of a thread: tile-invariant local pixel and LDS byte offset (ly < 0 marks a padding piece beyond the tile).
struct WPiece {
  short ly, lx;
  int lds;
  bool real;
};

template <int NI, int NJ, int AK, int GK, int TAPS = 9>
__global__ __launch_bounds__(NTHR, 2) void wgrad_bf16x3_kernel(HpfgWgradArgs p, int tiles_x, int tiles_y) {
  using L = Lds<NI, NJ, TAPS>;
  constexpr int HALO = Geo<TAPS>::HALO, WP = Geo<TAPS>::WP, A_SLOTS = Geo<TAPS>::A_SLOTS, A_PLANE = Geo<TAPS>::A_PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[L::BYTES];
  unsigned char* ldsA = lds;
  unsigned char* ldsG = lds + L::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = blockIdx.y * 16 * NI, co0 = blockIdx.z * 16 * NJ;
  const int H = p.H, W = p.W;
  const ActCtx cxa = make_ctx(p.a0), cxg = make_ctx(p.g);
  const HpfgAct none = {};
#ifdef HPFG_TRACE
  const bool tr_on = (p.math & 0x2000) && tid == 0;
  unsigned long long* tr_buf =
      reinterpret_cast<unsigned long long*>(p.slab) + 256 * (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  int tr_i = 0;
#endif
  HPFG_WTR_REAL(11)
  HPFG_WTR(1)

  constexpr int TSTR = 4 / (NI * NJ);               // tap stride between the taps of one wave
  constexpr int NT = (TAPS + TSTR - 1) / TSTR;      // taps per wave (padded)
  const int wi = wave % NI, wj = (wave / NI) % NJ, wt0 = wave / (NI * NJ);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A thread stages the same 8-channel group in every piece (NTHR is a multiple of the groups per pixel): group tid % (2 NI) of
  // the 16 NI input channels, group tid % (2 NJ) of the 16 NJ output channels -- its tables are loaded once.
  constexpr int GA = 2 * NI, GG = 2 * NJ;
  const int ga = (tid % GA) * 8, gg = (tid % GG) * 8;
  const int cin_total = p.a0.C + p.a1.C;
  Tab ta, tg;
  const int ca = ci0 + ga, cg = co0 + gg;
  const bool cva = ca < cin_total, cvg = cg < p.g.C;
  load_tables<AK>(ta, p.a0, cva ? ca : 0, true);
  load_tables<GK>(tg, p.g, cvg ? cg : 0, true);

  // Staging is batched: every global load of a batch is in flight before the first one is consumed (a thread owns NA pieces of
  // the A tile and NG pieces of the dZ tile; one exposed memory latency per batch instead of one per piece).  Batch 0 = the A
  // pieces + the first GB dZ pieces, and batch 0 of the NEXT work item is requested before the MFMA phase of the current one.
  constexpr int NRA = RawCount<AK>::N, NRG = RawCount<GK>::N;
  constexpr int NA = (A_SLOTS * GA + NTHR - 1) / NTHR;
  constexpr int NG = (TH * TW * GG + NTHR - 1) / NTHR;
  constexpr int GB = NG < 2 ? NG : 2;                       // dZ pieces per batch
  constexpr int NGB = (NG + GB - 1) / GB;
  constexpr bool PREA = NRA * NA <= 8;                      // pool / concat sources: no registers to hold all of A across the MFMA phase
  WPiece pa[NA], pg[NG];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + i * NTHR;
    const int pix = idx / GA < A_SLOTS ? idx / GA : A_SLOTS - 1;
    pa[i].real = idx < A_SLOTS * GA;
    pa[i].ly = (short)(pix / WP - HALO);
    pa[i].lx = (short)(pix % WP - HALO);
    pa[i].lds = (ga >> 4) * 2 * A_PLANE + pix * 32 + (ga & 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int idx = tid + i * NTHR;
    const int pix = idx / GG < TH * TW ? idx / GG : TH * TW - 1;
    pg[i].real = idx < TH * TW * GG;
    pg[i].ly = (short)(pix / TW);
    pg[i].lx = (short)(pix % TW);
    pg[i].lds = (gg >> 4) * 2 * G_PLANE + pix * 32 + (gg & 8) * 2;
  }
  RawPiece<AK> rawA[PREA ? NA : 1];
  RawPiece<GK> rawG[GB];

#define HPFG_WG_A_COORD(I)                                                                            \
  const int gy = ty0 + pa[I].ly, gx = tx0 + pa[I].lx;                                                 \
  const bool ok = pa[I].real && cva && gy >= 0 && gy < H && gx >= 0 && gx < W;                        \
  const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx), cac = cva ? ca : 0;
#define HPFG_WG_G_COORD(I)                                                                            \
  const int gy = ty0 + pg[I].ly, gx = tx0 + pg[I].lx;                                                 \
  const bool ok = pg[I].real && cvg && gy < H && gx < W;                                              \
  const int gyc = gy > H - 1 ? H - 1 : gy, gxc = gx > W - 1 ? W - 1 : gx, cgc = cvg ? cg : 0;

  // per-lane transposed-read offsets: group g = lane>>4 covers x = 4g..4g+3; lane 4q+p of the group addresses row (pixel) q, cols 4p..
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int xoff = 4 * grp + q;                 // pixel x inside the 16-wide tile row handled by this lane's address
  const int ntiles = tiles_x * tiles_y;
  const int nwork = p.N * ntiles;

  int wk = blockIdx.x;
  if (wk < nwork) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    if (PREA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        HPFG_WG_A_COORD(i)
        issue_piece<AK>(rawA[PREA ? i : 0], p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      HPFG_WG_G_COORD(i)
      issue_piece<GK>(rawG[i], p.g, none, cxg, n, gyc, gxc, cgc, ok);
    }
  }
  HPFG_WTR(2)
  for (; wk < nwork; wk += gridDim.x) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    __syncthreads();            // the previous item's MFMA phase has finished reading the tiles
    HPFG_WTR(3)
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      HPFG_WG_A_COORD(i)
      f32x4 v0, v1;
      if (!PREA) issue_piece<AK>(rawA[0], p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);      // one piece at a time
      finish_piece<AK>(v0, v1, rawA[PREA ? i : 0], ta, p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
      if (pa[i].real) {
        bf16x8 hi, lo;
        split8(v0, v1, hi, lo);
        *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds) = hi;
        *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds + A_PLANE) = lo;
      }
    }
    HPFG_WTR(4)
#pragma unroll
    for (int b = 0; b < NGB; ++b) {
      if (b > 0) {
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          if (b * GB + i < NG) {
            HPFG_WG_G_COORD(b * GB + i)
            issue_piece<GK>(rawG[i], p.g, none, cxg, n, gyc, gxc, cgc, ok);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        if (b * GB + i < NG) {
          HPFG_WG_G_COORD(b * GB + i)
          f32x4 v0, v1;
          finish_piece<GK>(v0, v1, rawG[i], tg, p.g, none, cxg, n, gyc, gxc, cgc, ok);
          if (pg[b * GB + i].real) {
            bf16x8 hi, lo;
            split8(v0, v1, hi, lo);
            *reinterpret_cast<bf16x8*>(ldsG + pg[b * GB + i].lds) = hi;
            *reinterpret_cast<bf16x8*>(ldsG + pg[b * GB + i].lds + G_PLANE) = lo;
          }
        }
      }
    }
    HPFG_WTR(5)
    {   // batch 0 of the next work item goes out before the MFMA phase (clamped to the last item: harmless reload, no branch)
      const int wk2 = wk + (int)gridDim.x < nwork ? wk + (int)gridDim.x : wk;
      const int n = wk2 / ntiles, tile = wk2 % ntiles;
      const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
      if (PREA) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          HPFG_WG_A_COORD(i)
          issue_piece<AK>(rawA[PREA ? i : 0], p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
        }
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        HPFG_WG_G_COORD(i)
        issue_piece<GK>(rawG[i], p.g, none, cxg, n, gyc, gxc, cgc, ok);
      }
    }
    HPFG_WTR(6)
    __syncthreads();
    HPFG_WTR(7)
    // ---- 4 MFMA k-steps of 32 pixels (2 tile rows each)
#pragma unroll
    for (int ks = 0; ks < TH / 2; ++ks) {
      const int r0 = 2 * ks;
      const unsigned char* b = ldsG + wj * 2 * G_PLANE + (r0 * TW + xoff) * 32 + pp * 8;
      const bf16x8 gh = tr_read8(b, b + TW * 32);
      const bf16x8 gl = tr_read8(b + G_PLANE, b + G_PLANE + TW * 32);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = wt0 + t * TSTR < TAPS ? wt0 + t * TSTR : TAPS - 1;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const unsigned char* a = ldsA + wi * 2 * A_PLANE + ((r0 + ky) * WP + xoff + kx) * 32 + pp * 8;
        const bf16x8 ah = tr_read8(a, a + WP * 32);
        const bf16x8 al = tr_read8(a + A_PLANE, a + A_PLANE + WP * 32);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh, acc[t], 0, 0, 0);
      }
    }
    HPFG_WTR(8)
  }
#undef HPFG_WG_A_COORD
#undef HPFG_WG_G_COORD
  // slab[s][tap][ci][co]; C/D layout: row (ci) = (lane>>4)*4 + r, col (co) = lane & 15
#ifdef HPFG_TRACE
  if (p.math & 0x2000) {
    HPFG_WTR(9)
    HPFG_WTR_REAL(12)
    return;
  }
#endif
  float* slab = p.slab + (long)blockIdx.x * p.taps * p.CinPad * p.CoutPad;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tap = wt0 + t * TSTR;
    if (tap < TAPS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + wi * 16 + (lane >> 4) * 4 + r, co = co0 + wj * 16 + (lane & 15);
        slab[((long)tap * p.CinPad + ci) * p.CoutPad + co] = acc[t][r];
      }
    }
  }
}

// workgroup shape (input tiles x output tiles of 16 channels) for a layer
inline void pick_shape(int CinPad, int CoutPad, int* ni, int* nj) {
  if (CinPad % 32 == 0 && CoutPad % 32 == 0) {
    *ni = 2;
    *nj = 2;
  } else {
    *ni = 1;
    *nj = CoutPad % 64 == 0 ? 4 : (CoutPad % 32 == 0 ? 2 : 1);
  }
}

template <int AK, int GK, int TAPS>
int launch_wgrad16_taps(const HpfgWgradArgs& a, hipStream_t st) {
  int ni, nj;
  pick_shape(a.CinPad, a.CoutPad, &ni, &nj);
  const int tx = (a.W + TW - 1) / TW, ty = (a.H + TH - 1) / TH;
  dim3 grid(a.S, a.CinPad / (16 * ni), a.CoutPad / (16 * nj));
  if (ni == 2) hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, 2, AK, GK, TAPS>), grid, dim3(NTHR), 0, st, a, tx, ty);
  else if (nj == 4) hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 4, AK, GK, TAPS>), grid, dim3(NTHR), 0, st, a, tx, ty);
  else if (nj == 2) hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 2, AK, GK, TAPS>), grid, dim3(NTHR), 0, st, a, tx, ty);
  else hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, 1, AK, GK, TAPS>), grid, dim3(NTHR), 0, st, a, tx, ty);
  return hpfg_launch_status("wgrad_bf16x3_kernel");
}

template <int AK, int GK>
int launch_wgrad16(const HpfgWgradArgs& a, hipStream_t st) {
  return launch_wgrad16_taps<AK, GK, 9>(a, st);
}

}  // namespace hpfg_wg16

int hpfg_wgrad16_launch_dz(const HpfgWgradArgs& a, int akind, hipStream_t st);
int hpfg_wgrad16_launch_plain(const HpfgWgradArgs& a, int akind, hipStream_t st);
int hpfg_wgrad16_launch_1x1(const HpfgWgradArgs& a, int akind, hipStream_t st);   // 1x1 convs: dZ is a plain gradient tensor
