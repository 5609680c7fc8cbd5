// Backward of an UpBlock's front end in ONE launch (model/unet.py:50-56: x1 = conv1x1(x1); x1 = upsample(x1)): the transpose of the
// bilinear x2 (align_corners=True) resize gathered straight into the A operand of the 1x1 conv's input gradient.
//
//   dU[n,y,x,c]      = sum_{taps} wy * wx * dUp[n, Y, X, c]              (upsample_bilinear2d backward; also written out: the 1x1 conv's
//                                                                          weight gradient contracts it later, on the side stream)
//   dA[n,y,x,ci]     = sum_c dU[n,y,x,c] * W[c][ci]                       (convolution_backward (input) of nn.Conv2d(C1, C2, 1))
//   csum[row][c]     = sum over the workgroup's pixels of dU              (rows of the 1x1 conv's bias gradient)
//   + (bwd_stats) the BatchNorm / LeakyReLU backward sums of the layer below from the finished dA tile, as in conv_bf16_kernel.h.
//
// Round 3 ran this as upsample_bwd_kernel -> conv1x1_bf16x3_kernel: two dependent launches per decoder level on the critical chain of
// backward (14-32 us + 11-27 us), with dU written and read back in between.  A launch on the chain costs the step more than the same
// work inside a neighbouring kernel (profiles/r04_bn_acc.txt), so the gather became the conv's loader.  Workgroup = an 8 x 8 low-resolution
// pixel tile of one image x ALL output channels (the gather is done once), 4 waves; per 32-channel chunk a thread gathers one 8-channel
// piece as two halves of 25 unconditional float4 loads (tap tables padded to 5 per axis with zero weights: misc.hip::upsample_bwd_kernel).
#include "conv_bf16_kernel.h"

namespace {

using namespace hpfg_conv16;
constexpr int UT = 5;

__device__ inline void up_taps5(int lo, int L, short* idx, float* wgt) {
  const int O = 2 * L;
  const float r = O > 1 ? (float)(L - 1) / (float)(O - 1) : 0.f;
  int cnt = 0, b = 2 * lo - 2, e = 2 * lo + 4;
  if (b < 0) b = 0;
  if (e > O - 1) e = O - 1;
  for (int o = b; o <= e; ++o) {
    const float f = r * (float)o;
    const int i0 = (int)f, i1 = i0 + (i0 < L - 1 ? 1 : 0);
    const float w1 = f - (float)i0, w0 = 1.f - w1;
    float w = 0.f;
    if (i0 == lo) w += w0;
    if (i1 == lo) w += w1;
    if ((i0 == lo || i1 == lo) && cnt < UT) {
      idx[cnt] = (short)o;
      wgt[cnt] = w;
      ++cnt;
    }
  }
  for (int k = cnt; k < UT; ++k) {      // padding: a valid index, weight 0
    idx[k] = cnt > 0 ? idx[cnt - 1] : 0;
    wgt[k] = 0.f;
  }
}

template <class C>
__device__ __forceinline__ void upfuse_body(const HpfgUpDgradArgs& q, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 1 && C::TH == 8 && C::TW == 8 && C::NLD == 1, "8 x 8 tiles, one piece per thread and chunk");
  const HpfgConvArgs& p = q.d;
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[C::BUF_BYTES + STAT_BYTES];
  __shared__ short t_i[2][8][UT];
  __shared__ float t_w[2][8][UT];
  __shared__ float csl[4][32];
  float* ldsf = reinterpret_cast<float*>(lds + C::BUF_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % C::WM, wn = wave / C::WM;
  const int tile = blockIdx.x, n = blockIdx.y;
  const int ty0 = (tile / tiles_x) * 8, tx0 = (tile % tiles_x) * 8;
  const int H = p.H, W = p.W, Ho = 2 * H, Wo = 2 * W, C2 = q.C2, ps = q.dup_pstride;
  if (tid < 16) {
    const int which = tid >> 3, k = tid & 7, L = which ? W : H;
    int lo = (which ? tx0 : ty0) + k;
    lo = lo < L ? lo : L - 1;
    up_taps5(lo, L, t_i[which][k], t_w[which][k]);
  }
  f32x4 acc[C::MI][C::NI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kg = lane >> 4;
  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (kg * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }
  const int g8 = (tid % C::NG) * 8;
  const int nchunks = (C2 + C::KC - 1) / C::KC;
  const int ntn = p.CoutPad / 16;
  const int nt0 = wn * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);
  const Piece pc = make_piece<C>(tid, 0);
  __syncthreads();
  int rowoff[UT], ix[UT];
  float wy[UT], wx[UT];
#pragma unroll
  for (int k = 0; k < UT; ++k) {
    rowoff[k] = ((n * Ho + t_i[0][pc.ly][k]) * Wo) * ps;
    wy[k] = t_w[0][pc.ly][k];
    ix[k] = t_i[1][pc.lx][k] * ps;
    wx[k] = t_w[1][pc.lx][k];
  }
  const int gy = ty0 + pc.ly, gx = tx0 + pc.lx;
  const bool inimg = gy < H && gx < W;
  const int row = n * (tiles_x * tiles_y) + tile;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * C::KC + g8;
    const bool ok = inimg && c0 < C2;
    const float* base = q.dup + (c0 < C2 ? c0 : 0);
    // the gather, one high-resolution row at a time: 10 unconditional float4 loads (5 horizontal taps x 8 channels) per row, the next row's
    // loads in flight while this row is reduced (the whole 50 at once cost 256 VGPRs + 76 AGPRs: one wave per SIMD)
    f32x4 v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 t[2][UT][2];
#pragma unroll
    for (int b = 0; b < UT; ++b) {
      t[0][b][0] = ld4(base, rowoff[0] + ix[b]);
      t[0][b][1] = ld4(base, rowoff[0] + ix[b] + 4);
    }
#pragma unroll
    for (int a = 0; a < UT; ++a) {
      if (a + 1 < UT) {
#pragma unroll
        for (int b = 0; b < UT; ++b) {
          t[(a + 1) & 1][b][0] = ld4(base, rowoff[a + 1] + ix[b]);
          t[(a + 1) & 1][b][1] = ld4(base, rowoff[a + 1] + ix[b] + 4);
        }
      }
      f32x4 r0 = wx[0] * t[a & 1][0][0], r1 = wx[0] * t[a & 1][0][1];
#pragma unroll
      for (int b = 1; b < UT; ++b) {
        r0 += wx[b] * t[a & 1][b][0];
        r1 += wx[b] * t[a & 1][b][1];
      }
      v[0] += wy[a] * r0;
      v[1] += wy[a] * r1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[0][j] = ok ? v[0][j] : 0.f;
      v[1][j] = ok ? v[1][j] : 0.f;
    }
    __syncthreads();          // the previous chunk's fragment reads (and its channel-sum row) are done
    store_piece<C>(lds, pc, v[0], v[1]);
    if (ok && q.dU) {
      float* o = q.dU + ((long)(n * H + gy) * W + gx) * C2 + c0;
      *reinterpret_cast<f32x4*>(o) = v[0];
      *reinterpret_cast<f32x4*>(o + 4) = v[1];
    }
    // channel sums of the chunk over the tile: a wave holds 16 pixels x 4 channel groups (lane & 3 = group)
    f32x4 a0 = v[0], a1 = v[1];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0[j] += __shfl_xor(a0[j], o);
        a1[j] += __shfl_xor(a1[j], o);
      }
    if (lane < 4) {
      *reinterpret_cast<f32x4*>(&csl[wave][lane * 8]) = a0;
      *reinterpret_cast<f32x4*>(&csl[wave][lane * 8 + 4]) = a1;
    }
    bf16x8 bh[C::NI], bl[C::NI];
    load_b<C>(bh, bl, wpk, ch, ntn, nt0, lane);
    __syncthreads();
    if (tid < 32 && q.csum && ch * C::KC + tid < C2)
      q.csum[(long)row * C2 + ch * C::KC + tid] = (csl[0][tid] + csl[1][tid]) + (csl[2][tid] + csl[3][tid]);
#pragma unroll
    for (int m = 0; m < C::MI; ++m) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + aoff[m]);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + aoff[m] + C::PLANE);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) { HPFG16_MFMA3(acc[m][j], ah, al, bh[j], bl[j]) }
    }
  }
  f32x4 s1[C::NI], s2[C::NI], bias[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
    bias[j] = s1[j];
  }
  conv16_store_tile<C, true>(p, acc, s1, s2, bias, lane, wm, nt0, n, ty0, tx0);
  __syncthreads();
  conv16_flush_stats<C, true>(p, s1, s2, ldsf, tid, lane, wm, wn, 0, row);
}

// (the occupancy target keeps the 25 loads of a half-piece in flight: left to itself the scheduler trades them for registers and the gather
// becomes load / wait / multiply in a row -- misc.hip::upsample_bwd_kernel)
template <class C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void upfuse_kernel(HpfgUpDgradArgs q, int tiles_x, int tiles_y) {
  upfuse_body<C>(q, tiles_x, tiles_y);
}

template <class C>
int launch(const HpfgUpDgradArgs& a, hipStream_t st) {
  const int tx = (a.d.W + 7) / 8, ty = (a.d.H + 7) / 8;
  hipLaunchKernelGGL((upfuse_kernel<C>), dim3(tx * ty, a.d.N), dim3(256), 0, st, a, tx, ty);
  return hpfg_launch_status("upfuse_kernel");
}

}  // namespace

extern "C" int hpfg_up_dgrad_rows(int N, int H, int W) { return N * ((H + 7) / 8) * ((W + 7) / 8); }

extern "C" int hpfg_up_dgrad(const HpfgUpDgradArgs* a, void* stream) {
  HPFG_ARG_CHECK(a && a->dup && a->d.wpk && a->d.out, "up_dgrad: null pointer");
  const HpfgConvArgs& d = a->d;
  HPFG_ARG_CHECK((d.math & 0xff) == HPFG_MATH_BF16X3 && d.taps == 1 && !d.bias && !d.out_split, "up_dgrad: a bf16x3 1x1 input gradient without bias / out_split");
  HPFG_ARG_CHECK(d.N > 0 && d.N < 65536 && d.H > 0 && d.W > 0 && d.H <= 16384 && d.W <= 16384, "up_dgrad: bad N/H/W");
  HPFG_ARG_CHECK(a->C2 >= 8 && a->C2 % 8 == 0 && a->C2 <= 256 && a->dup_pstride >= a->C2 && a->dup_pstride % 4 == 0, "up_dgrad: bad C2 %d / pixel stride %d", a->C2,
                 a->dup_pstride);
  HPFG_ARG_CHECK(d.Cout == d.CoutPad && (d.CoutPad == 32 || d.CoutPad == 64 || d.CoutPad == 128 || d.CoutPad == 256) && d.out_pstride >= d.Cout &&
                     d.out_pstride % 4 == 0,
                 "up_dgrad: the conv's input channels must be 32 / 64 / 128 / 256 (got %d)", d.Cout);
  HPFG_ARG_CHECK((long)d.N * 4 * d.H * d.W * a->dup_pstride < (1L << 31), "up_dgrad: the gradient tensor exceeds 32-bit element offsets");
  if (d.bwd_stats) {
    HPFG_ARG_CHECK((d.stat_partials || d.stat_acc) && d.bwd_of.z && d.bwd_of.bn && d.bwd_of.C == d.Cout && d.bwd_of.Hs == d.H && d.bwd_of.Ws == d.W &&
                       d.bwd_of.pstride % 4 == 0,
                   "up_dgrad: bwd_stats needs stat_partials or stat_acc and bwd_of describing the layer at the output size");
    HPFG_ARG_CHECK(!d.stat_acc || (d.stat_shards >= 1 && d.stat_shards <= HPFG_ACC_MAX_SHARDS && (d.stat_shards & (d.stat_shards - 1)) == 0), "up_dgrad: bad stat_shards");
  } else {
    HPFG_ARG_CHECK(!d.stat_partials && !d.stat_acc, "up_dgrad: stat_partials / stat_acc without bwd_stats");
  }
  hipStream_t st = (hipStream_t)stream;
  switch (d.CoutPad) {
    case 32: return launch<Cfg<8, 8, 2, 2, 1, 1, 32>>(*a, st);
    case 64: return launch<Cfg<8, 8, 1, 4, 1, 1, 32>>(*a, st);
    case 128: return launch<Cfg<8, 8, 1, 4, 2, 1, 32>>(*a, st);
    default: return launch<Cfg<8, 8, 1, 4, 4, 1, 32>>(*a, st);
  }
}
