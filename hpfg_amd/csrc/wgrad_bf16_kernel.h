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

// concat A source: the upsampled half's bilinear taps of an (8+2)x(16+2) tile come from a 7x11 low-res patch (see conv_bf16_kernel.h)
constexpr int UP_SH = TH / 2 + 3, UP_SW = TW / 2 + 3;

template <int NI, int NJ, int TAPS>
struct Lds {
  static constexpr int A_BYTES = NI * 2 * Geo<TAPS>::A_PLANE;
  static constexpr int G_BYTES = NJ * 2 * G_PLANE;
  static constexpr int UP_BYTES = UP_SH * UP_SW * 16 * NI * 4;      // fp32 patch [pixel][16 NI channels] (concat sources only)
  static constexpr int BYTES = A_BYTES + G_BYTES;
};

// a 32-byte zero piece: an out-of-image (or padding) piece of a SPLIT16 source is LOADED from here -- one 64-bit address select instead of
// eight value selects per piece, and no clamped coordinates to compute
static __device__ __attribute__((aligned(32))) const float hpfg_zero_piece[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

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

__device__ __forceinline__ int up_base_w(int o0, int L) {      // first low-res row / column a tile with halo touches
  const float r = L > 1 ? (float)(L - 1) / (float)(2 * L - 1) : 0.f;
  return (int)(r * (float)(o0 < 0 ? 0 : o0));
}

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
  // Concat input (torch.cat([skip, up])): a workgroup's 16 NI input channels lie entirely in the skip half (a0.C % (16 NI) == 0)
  // or entirely in the upsampled half.  Skip workgroups load BN + LeakyReLU pieces like a plain activated source; upsampled
  // workgroups fetch the low-res patch of the tile (<= 2 pieces of 2 float4 per thread, prefetched across the MFMA phase like
  // every other batch) into LDS and blend the four taps of each A piece from there -- the 8-float4-per-piece loader it replaces
  // had to go one piece at a time and paid three memory latencies per work item.
  constexpr bool CATA = AK == HPFG_KIND_CAT;
  constexpr int SA = CATA ? HPFG_KIND_BNACT : AK;          // loader kind of the per-pixel A pieces
  __shared__ __attribute__((aligned(16))) unsigned char lds[L::BYTES + (CATA ? L::UP_BYTES : 0)];
  unsigned char* ldsA = lds;
  unsigned char* ldsG = lds + L::A_BYTES;
  float* ldsU = reinterpret_cast<float*>(lds + L::BYTES);
  (void)ldsU;
  // dZ producer tables (scale, shift, k1, k2, k3 of the 16 NJ output channels) live in LDS and are re-read per work item: as
  // registers they were 40 VGPRs held across the MFMA phase for nothing
  __shared__ __attribute__((aligned(16))) float ldsTG[5][16 * NJ];
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
  const bool a_up = CATA && ci0 >= p.a0.C;                // workgroup-uniform
  if (!a_up) load_tables<SA>(ta, p.a0, cva ? ca : 0, true);
  if (GK == HPFG_KIND_DZ) {      // (table rows, or k1 .. k3 derived from the backward sum accumulators: HpfgAct.bn_acc)
    HpfgAct gs = p.g;          // this workgroup's 16 NJ output channels
    gs.bn_coff += co0;
    gs.C = p.g.C - co0 < 16 * NJ ? p.g.C - co0 : 16 * NJ;
    hpfg_dz_rows_to_lds(gs, &ldsTG[0][0], 16 * NJ, 16 * NJ, tid, NTHR);
  }

  // Staging is batched: every global load of a batch is in flight before the first one is consumed (a thread owns NA pieces of
  // the A tile and NG pieces of the dZ tile; one exposed memory latency per batch instead of one per piece).  Batch 0 = the A
  // pieces + the first GB dZ pieces, and batch 0 of the NEXT work item is requested before the MFMA phase of the current one.
  constexpr int NRA = RawCount<SA>::N, NRG = RawCount<GK>::N;
  constexpr int CW = 16 * NI;                               // channels of the patch
  constexpr int NU = CATA ? (UP_SH * UP_SW * GA + NTHR - 1) / NTHR : 1;   // patch pieces per thread
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
  // SPLIT16 operands (side tensors stored already split, dense at the layer's size): a piece's address is the tile's base plus a
  // tile-invariant offset, and its staging is a copy -- no producer chain, no clamps, no value selects (VALU per MFMA 5.1 -> 3.9 from the copy
  // alone, profiles/r05_step_sq_counters.txt; the address arithmetic below is what was left of the loader)
  constexpr bool SPA = SA == HPFG_KIND_SPLIT, SPG = GK == HPFG_KIND_SPLIT;
  int relA[SPA ? NA : 1], relG[SPG ? NG : 1];
  if constexpr (SPA) {
#pragma unroll
    for (int i = 0; i < NA; ++i) relA[i] = (pa[i].ly * W + pa[i].lx) * p.a0.pstride + (cva ? ca : 0);
  }
  if constexpr (SPG) {
#pragma unroll
    for (int i = 0; i < NG; ++i) relG[i] = (pg[i].ly * W + pg[i].lx) * p.g.pstride + (cvg ? cg : 0);
  }
  (void)relA;
  (void)relG;
  RawPiece<SA> rawA[PREA ? NA : 1];
  RawPiece<GK> rawG[GB];
  f32x4 rawU[NU][2];
#define HPFG_WG_U_ISSUE()                                                                              \
  {                                                                                                   \
    const int sy_b = up_base_w(ty0 - 1, p.a1.Hs), sx_b = up_base_w(tx0 - 1, p.a1.Ws);                 \
    _Pragma("unroll") for (int i = 0; i < NU; ++i) {                                                  \
      const int pix = (tid + i * NTHR) / GA;                                                          \
      int sy = sy_b + pix / UP_SW, sx = sx_b + pix % UP_SW;                                           \
      sy = sy > p.a1.Hs - 1 ? p.a1.Hs - 1 : sy;                                                       \
      sx = sx > p.a1.Ws - 1 ? p.a1.Ws - 1 : sx;                                                       \
      const int off = ((n * p.a1.Hs + sy) * p.a1.Ws + sx) * p.a1.pstride + (cva ? ca - p.a0.C : 0);   \
      rawU[i][0] = ld4(p.a1.z, off);                                                                  \
      rawU[i][1] = ld4(p.a1.z, off + 4);                                                              \
    }                                                                                                 \
  }

#define HPFG_WG_A_COORD(I)                                                                            \
  const int gy = ty0 + pa[I].ly, gx = tx0 + pa[I].lx;                                                 \
  const bool ok = pa[I].real && cva && gy >= 0 && gy < H && gx >= 0 && gx < W;                        \
  const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx), cac = cva ? ca : 0;
#define HPFG_WG_G_COORD(I)                                                                            \
  const int gy = ty0 + pg[I].ly, gx = tx0 + pg[I].lx;                                                 \
  const bool ok = pg[I].real && cvg && gy < H && gx < W;                                              \
  const int gyc = gy > H - 1 ? H - 1 : gy, gxc = gx > W - 1 ? W - 1 : gx, cgc = cvg ? cg : 0;

  // issue of one A / dZ piece of work item (n, ty0, tx0) into RAW: the SPLIT16 fast path or the generic loader
#define HPFG_WG_A_ISSUE(I, RAW)                                                                                               \
  if constexpr (SPA) {                                                                                                        \
    const int gy_ = ty0 + pa[I].ly, gx_ = tx0 + pa[I].lx;                                                                    \
    const bool ok_ = pa[I].real && cva && (unsigned)gy_ < (unsigned)H && (unsigned)gx_ < (unsigned)W;                         \
    const float* src_ = ok_ ? p.a0.z + ((n * H + ty0) * W + tx0) * p.a0.pstride + relA[I] : hpfg_zero_piece;                  \
    (RAW).v[0] = ld4(src_, 0);                                                                                                \
    (RAW).v[1] = ld4(src_, 4);                                                                                                \
  } else {                                                                                                                    \
    HPFG_WG_A_COORD(I)                                                                                                        \
    issue_piece<SA>(RAW, p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);                                                              \
  }
#define HPFG_WG_G_ISSUE(I, RAW)                                                                                               \
  if constexpr (SPG) {                                                                                                        \
    const int gy_ = ty0 + pg[I].ly, gx_ = tx0 + pg[I].lx;                                                                    \
    const bool ok_ = pg[I].real && cvg && gy_ < H && gx_ < W;                                                                 \
    const float* src_ = ok_ ? p.g.z + ((n * H + ty0) * W + tx0) * p.g.pstride + relG[I] : hpfg_zero_piece;                    \
    (RAW).v[0] = ld4(src_, 0);                                                                                                \
    (RAW).v[1] = ld4(src_, 4);                                                                                                \
  } else {                                                                                                                    \
    HPFG_WG_G_COORD(I)                                                                                                        \
    issue_piece<GK>(RAW, p.g, none, cxg, n, gyc, gxc, cgc, ok);                                                               \
  }

  // per-lane transposed-read offsets: group g = lane>>4 covers x = 4g..4g+3; lane 4q+p of the group addresses row (pixel) q, cols 4p..
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int xoff = 4 * grp + q;                 // pixel x inside the 16-wide tile row handled by this lane's address
  const int ntiles = tiles_x * tiles_y;
  const int nwork = p.N * ntiles;

  // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (the linear workgroup id is x + S * (...), S % 8 == 0, so
  // x % 8 names the XCD and every (ci, co) pair of one pixel range already shares an L2).  Give every XCD a CONTIGUOUS range of pixel tiles,
  // swept side by side by its workgroups, so that tiles sharing a halo meet in one L2 as well.
  const int nxcd = gridDim.x % 8 == 0 ? 8 : 1;
  const int per_x = (nwork + nxcd - 1) / nxcd, wstep = (int)gridDim.x / nxcd;
  const int wend = ((int)blockIdx.x % nxcd + 1) * per_x < nwork ? ((int)blockIdx.x % nxcd + 1) * per_x : nwork;
  int wk = ((int)blockIdx.x % nxcd) * per_x + (int)blockIdx.x / nxcd;
  if (wk < wend) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    if (a_up) {
      HPFG_WG_U_ISSUE()
    } else if (PREA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        HPFG_WG_A_ISSUE(i, rawA[PREA ? i : 0])
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      HPFG_WG_G_ISSUE(i, rawG[i])
    }
  }
  HPFG_WTR(2)
  for (; wk < wend; wk += wstep) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    __syncthreads();            // the previous item's MFMA phase has finished reading the tiles
    HPFG_WTR(3)
    if (a_up) {
      // park the low-res patch, then blend every A piece's four taps from LDS (same expression order as finish_piece<CAT>)
      const int sy_b = up_base_w(ty0 - 1, p.a1.Hs), sx_b = up_base_w(tx0 - 1, p.a1.Ws);
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const int idx = tid + i * NTHR;
        if (idx < UP_SH * UP_SW * GA) {
          float* d = ldsU + (idx / GA) * CW + ga;
          *reinterpret_cast<f32x4*>(d) = rawU[i][0];
          *reinterpret_cast<f32x4*>(d + 4) = rawU[i][1];
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        HPFG_WG_A_COORD(i)
        int y0, y1, x0, x1;
        float wy1, wx1;
        up_coord(gyc, p.a1.Hs, y0, y1, wy1);
        up_coord(gxc, p.a1.Ws, x0, x1, wx1);
        const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
        const float* r0 = ldsU + ((y0 - sy_b) * UP_SW) * CW + ga;
        const float* r1 = ldsU + ((y1 - sy_b) * UP_SW) * CW + ga;
        const int o0 = (x0 - sx_b) * CW, o1 = (x1 - sx_b) * CW;
        const f32x4 a00 = *reinterpret_cast<const f32x4*>(r0 + o0), b00 = *reinterpret_cast<const f32x4*>(r0 + o0 + 4);
        const f32x4 a01 = *reinterpret_cast<const f32x4*>(r0 + o1), b01 = *reinterpret_cast<const f32x4*>(r0 + o1 + 4);
        const f32x4 a10 = *reinterpret_cast<const f32x4*>(r1 + o0), b10 = *reinterpret_cast<const f32x4*>(r1 + o0 + 4);
        const f32x4 a11 = *reinterpret_cast<const f32x4*>(r1 + o1), b11 = *reinterpret_cast<const f32x4*>(r1 + o1 + 4);
        f32x4 v0 = wy0 * (wx0 * a00 + wx1 * a01) + wy1 * (wx0 * a10 + wx1 * a11);
        f32x4 v1 = wy0 * (wx0 * b00 + wx1 * b01) + wy1 * (wx0 * b10 + wx1 * b11);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v0[j] = ok ? v0[j] : 0.f;
          v1[j] = ok ? v1[j] : 0.f;
        }
        if (pa[i].real) {
          bf16x8 hi, lo;
          split8(v0, v1, hi, lo);
          *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds) = hi;
          *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds + A_PLANE) = lo;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if constexpr (SPA) {          // a copy: the (hi | lo) words as they were loaded
          if (pa[i].real) {
            *reinterpret_cast<f32x4*>(ldsA + pa[i].lds) = rawA[PREA ? i : 0].v[0];
            *reinterpret_cast<f32x4*>(ldsA + pa[i].lds + A_PLANE) = rawA[PREA ? i : 0].v[1];
          }
        } else {
          HPFG_WG_A_COORD(i)
          f32x4 v0, v1;
          if (!PREA) issue_piece<SA>(rawA[0], p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);      // one piece at a time
          finish_piece<SA>(v0, v1, rawA[PREA ? i : 0], ta, p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
          if (pa[i].real) {
            bf16x8 hi, lo;
            split_piece<SA>(v0, v1, hi, lo);
            *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds) = hi;
            *reinterpret_cast<bf16x8*>(ldsA + pa[i].lds + A_PLANE) = lo;
          }
        }
      }
    }
    HPFG_WTR(4)
    if (GK == HPFG_KIND_DZ) {
      tg.sc[0] = *reinterpret_cast<const f32x4*>(&ldsTG[0][gg]);
      tg.sc[1] = *reinterpret_cast<const f32x4*>(&ldsTG[0][gg + 4]);
      tg.sh[0] = *reinterpret_cast<const f32x4*>(&ldsTG[1][gg]);
      tg.sh[1] = *reinterpret_cast<const f32x4*>(&ldsTG[1][gg + 4]);
      tg.k1[0] = *reinterpret_cast<const f32x4*>(&ldsTG[2][gg]);
      tg.k1[1] = *reinterpret_cast<const f32x4*>(&ldsTG[2][gg + 4]);
      tg.k2[0] = *reinterpret_cast<const f32x4*>(&ldsTG[3][gg]);
      tg.k2[1] = *reinterpret_cast<const f32x4*>(&ldsTG[3][gg + 4]);
      tg.k3[0] = *reinterpret_cast<const f32x4*>(&ldsTG[4][gg]);
      tg.k3[1] = *reinterpret_cast<const f32x4*>(&ldsTG[4][gg + 4]);
    }
#pragma unroll
    for (int b = 0; b < NGB; ++b) {
      if (b > 0) {
#pragma unroll
        for (int i = 0; i < GB; ++i) {
          if (b * GB + i < NG) {
            HPFG_WG_G_ISSUE(b * GB + i, rawG[i])
          }
        }
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        if (b * GB + i < NG) {
          if constexpr (SPG) {
            if (pg[b * GB + i].real) {
              *reinterpret_cast<f32x4*>(ldsG + pg[b * GB + i].lds) = rawG[i].v[0];
              *reinterpret_cast<f32x4*>(ldsG + pg[b * GB + i].lds + G_PLANE) = rawG[i].v[1];
            }
          } else {
            HPFG_WG_G_COORD(b * GB + i)
            f32x4 v0, v1;
            finish_piece<GK>(v0, v1, rawG[i], tg, p.g, none, cxg, n, gyc, gxc, cgc, ok);
            if (pg[b * GB + i].real) {
              bf16x8 hi, lo;
              split_piece<GK>(v0, v1, hi, lo);
              *reinterpret_cast<bf16x8*>(ldsG + pg[b * GB + i].lds) = hi;
              *reinterpret_cast<bf16x8*>(ldsG + pg[b * GB + i].lds + G_PLANE) = lo;
            }
          }
        }
      }
    }
    HPFG_WTR(5)
    {   // batch 0 of the next work item goes out before the MFMA phase (clamped to the last item: harmless reload, no branch)
      const int wk2 = wk + wstep < wend ? wk + wstep : wk;
      const int n = wk2 / ntiles, tile = wk2 % ntiles;
      const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
      if (a_up) {
        HPFG_WG_U_ISSUE()
      } else if (PREA) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          HPFG_WG_A_ISSUE(i, rawA[PREA ? i : 0])
        }
      }
#pragma unroll
      for (int i = 0; i < GB; ++i) {
        HPFG_WG_G_ISSUE(i, rawG[i])
      }
    }
    HPFG_WTR(6)
    __syncthreads();
    HPFG_WTR(7)
    // ---- 4 MFMA k-steps of 32 pixels (2 tile rows each)
    if constexpr (NT == 9 && TAPS == 9) {
      // A wave that owns all nine taps (the 2 x 2 channel-rich shape): the A fragment of (k-step ks, kernel row ky) covers tile rows
      // 2 ks + ky and 2 ks + ky + 1, so (ks, ky = 2) and (ks + 1, ky = 0) are the SAME registers -- carried over instead of read again:
      // 27 instead of 36 fragment pairs per work item.  The MFMA phase of this kernel is bound by its transposing LDS reads (40 per 27
      // MFMAs before, LDS shared by the 8 waves of a CU), not by the matrix pipe.  Every accumulator sees the same MFMAs in the same order.
      bf16x8 ch[3], cl[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const unsigned char* a = ldsA + wi * 2 * A_PLANE + (xoff + kx) * 32 + pp * 8;
        ch[kx] = tr_read8(a, a + WP * 32);
        cl[kx] = tr_read8(a + A_PLANE, a + A_PLANE + WP * 32);
      }
#pragma unroll
      for (int ks = 0; ks < TH / 2; ++ks) {
        const int r0 = 2 * ks;
        const unsigned char* b = ldsG + wj * 2 * G_PLANE + (r0 * TW + xoff) * 32 + pp * 8;
        const bf16x8 gh = tr_read8(b, b + TW * 32);
        const bf16x8 gl = tr_read8(b + G_PLANE, b + G_PLANE + TW * 32);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            bf16x8 ah = ch[kx], al = cl[kx];
            if (ky > 0) {
              const unsigned char* a = ldsA + wi * 2 * A_PLANE + ((r0 + ky) * WP + xoff + kx) * 32 + pp * 8;
              ah = tr_read8(a, a + WP * 32);
              al = tr_read8(a + A_PLANE, a + A_PLANE + WP * 32);
              if (ky == 2) {
                ch[kx] = ah;
                cl[kx] = al;
              }
            }
            const int t = ky * 3 + kx;
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh, acc[t], 0, 0, 0);
          }
        }
      }
    } else {
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
    }
    HPFG_WTR(8)
  }
#undef HPFG_WG_A_ISSUE
#undef HPFG_WG_G_ISSUE
#undef HPFG_WG_A_COORD
#undef HPFG_WG_G_COORD
#undef HPFG_WG_U_ISSUE
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
int launch_wgrad16_taps(const HpfgWgradArgs& a_in, hipStream_t st) {
  HpfgWgradArgs a = a_in;
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
int hpfg_wgrad16_launch_split(const HpfgWgradArgs& a, int akind, hipStream_t st);   // an operand stored already split (HPFG_ACT_SPLIT16)
