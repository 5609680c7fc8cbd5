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
// Workgroup = 3 waves; wave w owns kernel row ky = w (taps 3w..3w+2) for all NJ output-channel tiles; it walks a strided list
// of (image, 8x16 tile) work items with the accumulators in registers and writes one slab; wgrad.hip's slab_reduce_kernel sums
// the slabs in a fixed order.  Products are hi*hi + hi*lo + lo*hi (fp32 accumulate) as in conv_bf16_kernel.h.
#pragma once
#include "stage.h"

namespace hpfg_wg16 {

using namespace hpfg_stage;
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16;
constexpr int HP = TH + 2, WP = TW + 2;
constexpr int A_SLOTS = HP * WP;               // 180 pixel slots of 32 B
constexpr int A_PLANE = A_SLOTS * 32;          // bytes per hi / lo plane
constexpr int G_PLANE = TH * TW * 32;          // per output-channel tile, per hi / lo plane
constexpr int NTHR = 192;

template <int NJ>
struct Lds {
  static constexpr int A_BYTES = 2 * A_PLANE;
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

template <int NJ, int AK, int GK>
__global__ __launch_bounds__(NTHR) void wgrad_bf16x3_kernel(HpfgWgradArgs p, int tiles_x, int tiles_y) {
  using L = Lds<NJ>;
  __shared__ __attribute__((aligned(16))) unsigned char lds[L::BYTES];
  unsigned char* ldsA = lds;
  unsigned char* ldsG = lds + L::A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16 * NJ;
  const int H = p.H, W = p.W;
  const ActCtx cxa = make_ctx(p.a0), cxg = make_ctx(p.g);
  const HpfgAct none = {};

  f32x4 acc[3][NJ];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tables of this thread's channel group: A staging uses group (tid & 1) of the 16 input channels of this workgroup,
  // dZ staging uses group tid % (2*NJ) of the 16*NJ output channels
  const int ga = (tid & 1) * 8, gg = (tid % (2 * NJ)) * 8;
  const int cin_total = p.a0.C + p.a1.C;
  Tab ta, tg;
  const int ca = ci0 + ga, cg = co0 + gg;
  const bool cva = ca < cin_total, cvg = cg < p.g.C;
  load_tables<AK>(ta, p.a0, ca, cva);
  load_tables<GK>(tg, p.g, cg, cvg);

  // per-lane transposed-read offsets: group g = lane>>4 covers x = 4g..4g+3; lane 4q+p of the group addresses row (pixel) q, cols 4p..
  const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int xoff = 4 * grp + q;                 // pixel x inside the 16-wide tile row handled by this lane's address
  const int ntiles = tiles_x * tiles_y;
  const int nwork = p.N * ntiles;
  for (int wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
    __syncthreads();
    // ---- stage A (with halo): 180 pixels x 2 channel groups
#pragma unroll 1
    for (int idx = tid; idx < A_SLOTS * 2; idx += NTHR) {
      const int pix = idx >> 1;
      const int gy = ty0 + pix / WP - 1, gx = tx0 + pix % WP - 1;
      const bool ok = cva && gy >= 0 && gy < H && gx >= 0 && gx < W;
      f32x4 raw[RawCount<AK>::N], v0, v1;
      const int gyc = gy < 0 ? 0 : (gy > H - 1 ? H - 1 : gy), gxc = gx < 0 ? 0 : (gx > W - 1 ? W - 1 : gx), cac = cva ? ca : 0;
      issue_piece<AK>(raw, p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
      finish_piece<AK>(v0, v1, raw, ta, p.a0, p.a1, cxa, n, gyc, gxc, cac, ok);
      bf16x8 hi, lo;
      split8(v0, v1, hi, lo);
      *reinterpret_cast<bf16x8*>(ldsA + pix * 32 + ga * 2) = hi;
      *reinterpret_cast<bf16x8*>(ldsA + A_PLANE + pix * 32 + ga * 2) = lo;
    }
    // ---- stage dZ: 128 pixels x 2*NJ channel groups, planes [co tile][hi|lo][pixel][16 ch]
#pragma unroll 1
    for (int idx = tid; idx < TH * TW * 2 * NJ; idx += NTHR) {
      const int pix = idx / (2 * NJ);
      const int gy = ty0 + pix / TW, gx = tx0 + pix % TW;
      const bool ok = cvg && gy < H && gx < W;
      f32x4 raw[RawCount<GK>::N], v0, v1;
      const int gyc = gy > H - 1 ? H - 1 : gy, gxc = gx > W - 1 ? W - 1 : gx, cgc = cvg ? cg : 0;
      issue_piece<GK>(raw, p.g, none, cxg, n, gyc, gxc, cgc, ok);
      finish_piece<GK>(v0, v1, raw, tg, p.g, none, cxg, n, gyc, gxc, cgc, ok);
      bf16x8 hi, lo;
      split8(v0, v1, hi, lo);
      unsigned char* d = ldsG + (gg >> 4) * 2 * G_PLANE + pix * 32 + (gg & 8) * 2;
      *reinterpret_cast<bf16x8*>(d) = hi;
      *reinterpret_cast<bf16x8*>(d + G_PLANE) = lo;
    }
    __syncthreads();
    // ---- 4 MFMA k-steps of 32 pixels (2 tile rows each)
#pragma unroll
    for (int ks = 0; ks < TH / 2; ++ks) {
      const int r0 = 2 * ks;
      bf16x8 gh[NJ], gl[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const unsigned char* b = ldsG + j * 2 * G_PLANE + (r0 * TW + xoff) * 32 + pp * 8;
        gh[j] = tr_read8(b, b + TW * 32);
        gl[j] = tr_read8(b + G_PLANE, b + G_PLANE + TW * 32);
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const unsigned char* a = ldsA + ((r0 + wave) * WP + xoff + kx) * 32 + pp * 8;
        const bf16x8 ah = tr_read8(a, a + WP * 32);
        const bf16x8 al = tr_read8(a + A_PLANE, a + A_PLANE + WP * 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gh[j], acc[kx][j], 0, 0, 0);
          acc[kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, gl[j], acc[kx][j], 0, 0, 0);
          acc[kx][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, gh[j], acc[kx][j], 0, 0, 0);
        }
      }
    }
  }
  // slab[s][tap][ci][co]; C/D layout: row (ci) = (lane>>4)*4 + r, col (co) = lane & 15
  float* slab = p.slab + (long)blockIdx.x * p.taps * p.CinPad * p.CoutPad;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tap = wave * 3 + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + (lane >> 4) * 4 + r, co = co0 + j * 16 + (lane & 15);
        slab[((long)tap * p.CinPad + ci) * p.CoutPad + co] = acc[t][j][r];
      }
    }
}

inline int pick_nj(int CoutPad) { return CoutPad % 64 == 0 ? 4 : (CoutPad % 32 == 0 ? 2 : 1); }

template <int AK, int GK>
int launch_wgrad16(const HpfgWgradArgs& a, hipStream_t st) {
  const int nj = pick_nj(a.CoutPad);
  const int tx = (a.W + TW - 1) / TW, ty = (a.H + TH - 1) / TH;
  dim3 grid(a.S, a.CinPad / 16, a.CoutPad / (16 * nj));
  if (nj == 4) hipLaunchKernelGGL((wgrad_bf16x3_kernel<4, AK, GK>), grid, dim3(NTHR), 0, st, a, tx, ty);
  else if (nj == 2) hipLaunchKernelGGL((wgrad_bf16x3_kernel<2, AK, GK>), grid, dim3(NTHR), 0, st, a, tx, ty);
  else hipLaunchKernelGGL((wgrad_bf16x3_kernel<1, AK, GK>), grid, dim3(NTHR), 0, st, a, tx, ty);
  return hpfg_launch_status("wgrad_bf16x3_kernel");
}

}  // namespace hpfg_wg16

int hpfg_wgrad16_launch_dz(const HpfgWgradArgs& a, int akind, hipStream_t st);
int hpfg_wgrad16_launch_plain(const HpfgWgradArgs& a, int akind, hipStream_t st);
