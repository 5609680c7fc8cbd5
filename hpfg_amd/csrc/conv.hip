// Implicit-GEMM 3x3 / 1x1 convolution on gfx950 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
//   out[n,y,x,co] = bias[co] + sum_{tap,ci} A[n, y+dy(tap), x+dx(tap), ci] * W[co,ci,tap]
//
// A is a *virtual* activation (common.h): the producer's BatchNorm + LeakyReLU + Dropout (+ MaxPool2d(2) or bilinear x2
// upsample + channel concat) are applied while the input tile is staged into LDS, so no activated tensor ever exists in
// HBM.  Reference ops replaced: nn.Conv2d (model/unet.py:18,22,50,99), nn.BatchNorm2d statistics (:19,23),
// nn.LeakyReLU (:20,24), nn.Dropout (:21), nn.MaxPool2d (:37), nn.Upsample (:51), torch.cat (:57).
// The same kernel computes dgrad (A = virtual dZ, W = transposed/flipped weights).
//
// Mapping: one workgroup (4 waves) owns a TH x TW pixel tile of one image and a slice of output channels.  GEMM view:
// M = pixels, N = output channels, K = (tap, input channel).  The input tile (+1 halo) for 16 input channels lives in LDS
// pixel-major with stride 17 floats (conflict-free ds_read_b32 for the A operand: 16 pixels x 4 channels per MFMA);
// B operands come straight from global memory in pre-packed fragment order (one float4 per lane covers 4 k-steps).
#include "common.h"

namespace {

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

template <class C>
__global__ __launch_bounds__(256) void conv_mfma_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  __shared__ float lds[C::LDS_FLOATS > 2 * 4 * C::BN ? C::LDS_FLOATS : 2 * 4 * C::BN];
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

  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    // ---- stage the activated input tile for channels [ch*16, ch*16+16) ----
    for (int idx = tid; idx < C::HP * C::WP * 4; idx += 256) {
      int pix = idx >> 2, cq = idx & 3;
      int ly = pix / C::WP, lx = pix % C::WP;
      int gy = ty0 + ly - C::HALO, gx = tx0 + lx - C::HALO;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = cat_load4(p.a0, cx0, p.a1, cx1, n, gy, gx, ch * KC + cq * 4);
      float* d = lds + pix * PS + cq * 4;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < C::TAPS; ++tap) {
      const int toff = C::TAPS == 9 ? ((tap / 3) * C::WP + (tap % 3)) * PS : 0;
      f32x4 bf[C::NI];
#pragma unroll
      for (int j = 0; j < C::NI; ++j) bf[j] = wpk[((long)(tap * nchunks + ch) * ntn + nt0 + j) * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float a[C::MI];
#pragma unroll
        for (int m = 0; m < C::MI; ++m) a[m] = lds[aoff[m] + toff + ks * 4];
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
          p.out[((long)(n * H + gy) * W + gx) * p.out_pstride + co] = v;
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

template <class C>
int launch_cfg(const HpfgConvArgs& a, hipStream_t st) {
  int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  dim3 grid(tx * ty, a.N, a.CoutPad / C::BN);
  hipLaunchKernelGGL(conv_mfma_kernel<C>, grid, dim3(256), 0, st, a, tx, ty);
  return hpfg_launch_status("conv_mfma_kernel");
}

template <int TAPS>
int dispatch(const HpfgConvArgs& a, hipStream_t st) {
  const bool big = (a.H % 16 == 0) && (a.W % 16 == 0);
  const int cp = a.CoutPad;
  if (big) {
    if (cp % 64 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 4, TAPS>>(a, st);
    if (cp % 32 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 2, TAPS>>(a, st);
    return launch_cfg<Cfg<16, 16, 4, 1, 1, TAPS>>(a, st);
  }
  if (cp % 128 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 2, TAPS>>(a, st);
  if (cp % 64 == 0) return launch_cfg<Cfg<8, 8, 1, 4, 1, TAPS>>(a, st);
  if (cp % 32 == 0) return launch_cfg<Cfg<8, 8, 2, 2, 1, TAPS>>(a, st);
  return launch_cfg<Cfg<8, 8, 4, 1, 1, TAPS>>(a, st);
}

inline bool tile_is_big(int H, int W) { return (H % 16 == 0) && (W % 16 == 0); }

// ---- first layer: Cin <= 4, direct fp32 FMA (K = 9*Cin is too small for the matrix cores) --------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(HpfgAct x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, float* __restrict__ stat, int N, int H, int W,
                                                         int Cin, int tiles_x, int tiles_y) {
  constexpr int T = 16, TP = T + 2, CO = 16;
  __shared__ float tin[4][TP * TP];
  __shared__ float wl[9 * 4 * CO];
  __shared__ float red[2][4][CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x, n = blockIdx.y;
  const int ty0 = (tile / tiles_x) * T, tx0 = (tile % tiles_x) * T;
  for (int i = tid; i < 9 * Cin * CO; i += 256) {   // wl[(tap*Cin+ci)*16+co] = w[co][ci][tap]
    int co = i % CO, r = i / CO, ci = r % Cin, tap = r / Cin;
    wl[i] = w[(co * Cin + ci) * 9 + tap];
  }
  for (int i = tid; i < Cin * TP * TP; i += 256) {
    int ci = i / (TP * TP), pix = i % (TP * TP);
    int gy = ty0 + pix / TP - 1, gx = tx0 + pix % TP - 1;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x.z[(long)n * x.sn + (long)ci * x.sc + (long)gy * x.sy + (long)gx * x.sx];
    tin[ci][pix] = v;
  }
  __syncthreads();
  const int ly = tid / T, lx = tid % T;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = bias[co];
  for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float a = tin[ci][(ly + tap / 3) * TP + lx + tap % 3];
      const float* wr = wl + (tap * Cin + ci) * CO;
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[co] = fmaf(a, wr[co], acc[co]);
    }
  const int gy = ty0 + ly, gx = tx0 + lx;
  const bool valid = gy < H && gx < W;
  if (valid) {
    f32x4* o = reinterpret_cast<f32x4*>(out + ((long)(n * H + gy) * W + gx) * CO);
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  }
  if (stat) {
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      float v = valid ? acc[co] : 0.f, v2 = v * v;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        v += __shfl_xor(v, o);
        v2 += __shfl_xor(v2, o);
      }
      if (lane == 0) {
        red[0][wave][co] = v;
        red[1][wave][co] = v2;
      }
    }
    __syncthreads();
    if (tid < 2 * CO) {
      int which = tid / CO, co = tid % CO;
      float t = red[which][0][co] + red[which][1][co] + red[which][2][co] + red[which][3][co];
      long blk = (long)n * (tiles_x * tiles_y) + tile;
      stat[(blk * 2 + which) * CO + co] = t;
    }
  }
}

}  // namespace

extern "C" int hpfg_conv_stat_blocks(int N, int H, int W) {
  int t = tile_is_big(H, W) ? 16 : 8;
  return N * ((H + t - 1) / t) * ((W + t - 1) / t);
}

extern "C" int hpfg_conv_fwd(const HpfgConvArgs* a, void* stream) {
  HPFG_ARG_CHECK(a && a->wpk && a->out, "conv_fwd: null pointer");
  HPFG_ARG_CHECK(a->taps == 9 || a->taps == 1, "conv_fwd: taps must be 1 or 9 (got %d)", a->taps);
  HPFG_ARG_CHECK(a->CoutPad % 16 == 0 && a->Cout <= a->CoutPad && a->Cout > 0, "conv_fwd: bad Cout %d / pad %d", a->Cout, a->CoutPad);
  HPFG_ARG_CHECK(a->N > 0 && a->H > 0 && a->W > 0 && a->N < 65536, "conv_fwd: bad N/H/W");
  HPFG_ARG_CHECK(a->a0.mode != HPFG_ACT_NONE && a->a0.C > 0, "conv_fwd: a0 empty");
  HPFG_ARG_CHECK(a->a1.mode == HPFG_ACT_NONE || a->a0.C % 16 == 0, "conv_fwd: concat needs a0.C %% 16 == 0");
  HPFG_ARG_CHECK(a->out_pstride >= a->Cout, "conv_fwd: out_pstride < Cout");
  hipStream_t st = (hipStream_t)stream;
  return a->taps == 9 ? dispatch<9>(*a, st) : dispatch<1>(*a, st);
}

extern "C" int hpfg_conv3x3_first_fwd(const HpfgAct* x, const float* w_oihw, const float* bias, float* out, float* stat_partials,
                                      int N, int H, int W, int Cin, int Cout, void* stream) {
  HPFG_ARG_CHECK(x && w_oihw && bias && out, "conv_first: null pointer");
  HPFG_ARG_CHECK(Cin >= 1 && Cin <= 4 && Cout == 16, "conv_first: needs Cin<=4, Cout==16 (got %d,%d)", Cin, Cout);
  HPFG_ARG_CHECK(x->mode == HPFG_ACT_STRIDED, "conv_first: input must be a STRIDED source");
  int tx = (W + 15) / 16, ty = (H + 15) / 16;
  // stat partial layout must match hpfg_conv_stat_blocks(): 16x16 tiles only when H,W are multiples of 16
  HPFG_ARG_CHECK(stat_partials == nullptr || tile_is_big(H, W), "conv_first: BN partials need H,W multiples of 16 (got %dx%d)", H, W);
  hipLaunchKernelGGL(conv_first_kernel, dim3(tx * ty, N), dim3(256), 0, (hipStream_t)stream, *x, w_oihw, bias, out, stat_partials, N, H, W,
                     Cin, tx, ty);
  return hpfg_launch_status("conv_first_kernel");
}
