// Weight gradient of the 3x3 / 1x1 convolutions on gfx950 matrix cores (exact fp32, v_mfma_f32_16x16x4_f32).
//
//   dW[co,ci,tap] = sum_{n,y,x} A[n, y+dy(tap), x+dx(tap), ci] * dZ[n,y,x,co]
//
// Replaces autograd's convolution_backward (weight) for nn.Conv2d at model/unet.py:18,22,50,99.  Both operands are virtual:
// A is re-activated from the producer's raw output exactly as in the forward pass, dZ is rebuilt from the upstream gradient
// and the saved raw output via the BatchNorm-backward coefficients (common.h, HPFG_ACT_DZ).
//
// GEMM view: M = input channel (16 per workgroup), N = output channel (16*NJ per workgroup), K = pixels.  A workgroup walks
// a strided list of (image, 16x16 pixel tile) work items, keeps the per-tap accumulators in registers, and finally writes a
// private slab; a second kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics) and emits the
// gradient in PyTorch's [Cout][Cin][k][k] layout.  3x3: three waves, wave w owns kernel row ky = w.  1x1: wave w owns
// n-tile w.
#include "common.h"

namespace {

constexpr int T = 16;          // pixel tile edge
constexpr int PSA = 17;        // LDS pixel stride of the A tile (16 channels + 1)

template <int TAPS, int NJ>
struct WCfg {
  static constexpr int NW = TAPS == 9 ? 3 : NJ;           // waves per workgroup
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int TP = T + 2 * HALO;
  static constexpr int PSB = 16 * NJ + 1;                  // LDS pixel stride of the dZ tile
  static constexpr int LDS_A = TP * TP * PSA;
  static constexpr int LDS_B = T * T * PSB;
  static constexpr int NT = TAPS == 9 ? 3 : 1;             // taps per wave
  static constexpr int NA = TAPS == 9 ? NJ : 1;            // n-tiles per wave
};

template <int TAPS, int NJ>
__global__ __launch_bounds__((TAPS == 9 ? 192 : 64 * NJ)) void wgrad_kernel(HpfgWgradArgs p, int tiles_x, int tiles_y) {
  using C = WCfg<TAPS, NJ>;
  __shared__ float ldsA[C::LDS_A];
  __shared__ float ldsB[C::LDS_B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = 64 * C::NW;
  const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16 * NJ;
  const int H = p.H, W = p.W;
  const ActCtx cxa0 = make_ctx(p.a0), cxa1 = make_ctx(p.a1), cxg = make_ctx(p.g);

  f32x4 acc[C::NT][C::NA];
#pragma unroll
  for (int t = 0; t < C::NT; ++t)
#pragma unroll
    for (int j = 0; j < C::NA; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntiles = tiles_x * tiles_y;
  const int nwork = p.N * ntiles;
  for (int wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * T, tx0 = (tile % tiles_x) * T;
    __syncthreads();
    for (int idx = tid; idx < C::TP * C::TP * 4; idx += nthr) {
      int pix = idx >> 2, cq = idx & 3;
      int gy = ty0 + pix / C::TP - C::HALO, gx = tx0 + pix % C::TP - C::HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = cat_load4(p.a0, cxa0, p.a1, cxa1, n, gy, gx, ci0 + cq * 4);
      float* d = ldsA + pix * PSA + cq * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    for (int idx = tid; idx < T * T * 4 * NJ; idx += nthr) {
      int pix = idx / (4 * NJ), cq = idx % (4 * NJ);
      int gy = ty0 + pix / T, gx = tx0 + pix % T;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy < H && gx < W) v = act_load4(p.g, cxg, n, gy, gx, co0 + cq * 4);
      float* d = ldsB + pix * C::PSB + cq * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < T * T / 4; ++ks) {
      const int pix = ks * 4 + (lane >> 4);            // k index = pixel
      const int r = pix / T, c = pix % T;
      if (TAPS == 9) {
        float b[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = ldsB[pix * C::PSB + j * 16 + (lane & 15)];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float a = ldsA[((r + wave) * C::TP + c + kx) * PSA + (lane & 15)];
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

// dw[co][ci][tap] = sum_s slab[s][tap][ci][co]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int taps, int Cin,
                                                          int CinPad, int Cout, int CoutPad) {
  const long per = (long)taps * CinPad * CoutPad;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
    int co = (int)(i % CoutPad);
    int ci = (int)((i / CoutPad) % CinPad);
    int tap = (int)(i / ((long)CoutPad * CinPad));
    if (co >= Cout || ci >= Cin) continue;
    float t = 0.f;
    for (int s = 0; s < S; ++s) t += slab[s * per + i];
    dw[((long)co * Cin + ci) * taps + tap] = t;
  }
}

// db[c] = sum_p g[p*pstride + c]: two-stage deterministic reduction
__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ g, int pstride, long npix, int C, float* __restrict__ scratch) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int c = tid % C, pl = tid / C, PL = 256 / C;
  float a = 0.f;
  if (pl < PL)
    for (long pix = (long)blockIdx.x * PL + pl; pix < npix; pix += (long)gridDim.x * PL) a += g[pix * pstride + c];
  red[tid] = a;
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int l = 0; l < PL; ++l) t += red[l * C + tid];
    scratch[(long)blockIdx.x * C + tid] = t;
  }
}
__global__ __launch_bounds__(64) void channel_sum_stage2(const float* __restrict__ scratch, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) a += (double)scratch[(long)i * C + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (threadIdx.x == 0) out[c] = (float)a;
}

int pick_nj(int CoutPad) { return CoutPad % 64 == 0 ? 4 : (CoutPad % 32 == 0 ? 2 : 1); }

}  // namespace

extern "C" int hpfg_wgrad_splits(int N, int H, int W, int CinPad, int CoutPad, int taps) {
  (void)taps;
  int nj = pick_nj(CoutPad);
  long pairs = (long)(CinPad / 16) * (CoutPad / (16 * nj));
  long nwork = (long)N * ((H + T - 1) / T) * ((W + T - 1) / T);
  long s = 1536 / pairs;
  if (s < 1) s = 1;
  if (s > nwork) s = nwork;
  return (int)s;
}

extern "C" long hpfg_wgrad_slab_floats(int N, int H, int W, int CinPad, int CoutPad, int taps) {
  return (long)hpfg_wgrad_splits(N, H, W, CinPad, CoutPad, taps) * taps * CinPad * CoutPad;
}

extern "C" int hpfg_wgrad(const HpfgWgradArgs* a, void* stream) {
  HPFG_ARG_CHECK(a && a->slab && a->dw_oihw, "wgrad: null pointer");
  HPFG_ARG_CHECK(a->taps == 9 || a->taps == 1, "wgrad: taps must be 1 or 9");
  HPFG_ARG_CHECK(a->CinPad % 16 == 0 && a->CoutPad % 16 == 0 && a->Cin <= a->CinPad && a->Cout <= a->CoutPad, "wgrad: bad channel padding");
  HPFG_ARG_CHECK(a->S >= 1, "wgrad: S < 1");
  HPFG_ARG_CHECK(a->a1.mode == HPFG_ACT_NONE || a->a0.C % 16 == 0, "wgrad: concat needs a0.C %% 16 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int nj = pick_nj(a->CoutPad);
  const int tx = (a->W + T - 1) / T, ty = (a->H + T - 1) / T;
  dim3 grid(a->S, a->CinPad / 16, a->CoutPad / (16 * nj));
#define HPFG_WG(TAPS, NJ) hipLaunchKernelGGL((wgrad_kernel<TAPS, NJ>), grid, dim3(64 * WCfg<TAPS, NJ>::NW), 0, st, *a, tx, ty)
  if (a->taps == 9) {
    if (nj == 4) HPFG_WG(9, 4); else if (nj == 2) HPFG_WG(9, 2); else HPFG_WG(9, 1);
  } else {
    if (nj == 4) HPFG_WG(1, 4); else if (nj == 2) HPFG_WG(1, 2); else HPFG_WG(1, 1);
  }
#undef HPFG_WG
  int rc = hpfg_launch_status("wgrad_kernel");
  if (rc) return rc;
  long per = (long)a->taps * a->CinPad * a->CoutPad;
  int rb = (int)((per + 255) / 256);
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, st, a->slab, a->dw_oihw, a->S, a->taps, a->Cin, a->CinPad, a->Cout, a->CoutPad);
  return hpfg_launch_status("slab_reduce_kernel");
}

extern "C" int hpfg_channel_sum(const float* g, int pstride, long npix, int C, float* out, float* scratch, void* stream) {
  HPFG_ARG_CHECK(g && out && scratch && C >= 1 && C <= 256 && npix > 0, "channel_sum: bad args");
  int PL = 256 / C;
  long want = (npix + PL - 1) / PL;
  int nblk = (int)(want < 512 ? want : 512);   // scratch must hold 512*C floats
  hipLaunchKernelGGL(channel_sum_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, pstride, npix, C, scratch);
  hipLaunchKernelGGL(channel_sum_stage2, dim3(C), dim3(64), 0, (hipStream_t)stream, scratch, nblk, C, out);
  return hpfg_launch_status("channel_sum");
}
