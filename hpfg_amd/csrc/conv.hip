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
#include "conv_kernel.h"
#include "conv_bf16_kernel.h"

namespace {

struct HpfgFirstConvArgs {   // kernel argument of the first-layer kernels
  HpfgAct x;            // STRIDED source: the network input
  const float* w_oihw;
  const float* bias;
  float* out;
  float* stat_partials; // or NULL
  long long* stat_acc;  // or NULL: the sums go into the layer accumulator by integer atomics instead (HpfgConvArgs.stat_acc)
  int stat_shards;
};

inline bool tile_is_big(int H, int W) { return (H % 16 == 0) && (W % 16 == 0); }

// ---- first layer: Cin <= 4, direct fp32 FMA (K = 9*Cin is too small for the matrix cores) --------------------------
// Persistent: a workgroup walks tiles w = blockIdx.x, + gridDim.x, ...; the BatchNorm partial sums stay in registers across its
// tiles and are reduced once (the per-tile wave butterfly of 32 values used to cost more than the 144 FMAs of the convolution),
// so the layer also hands only gridDim.x rows to the finalize instead of one per tile.
__global__ __launch_bounds__(256) HPFG_NO_PK_F32 void conv_first_kernel(HpfgFirstConvArgs q, int N, int H, int W, int Cin, int tiles_x, int tiles_y) {
  constexpr int T = 16, TP = T + 2, CO = 16;
  const HpfgAct& x = q.x;
  const float* __restrict__ w = q.w_oihw;
  const float* __restrict__ bias = q.bias;
  float* __restrict__ out = q.out;
  float* __restrict__ stat = q.stat_partials;
  long long* __restrict__ sacc = q.stat_acc;
  __shared__ float tin[4][TP * TP];
  __shared__ __attribute__((aligned(16))) float wl[9 * 4 * CO];
  __shared__ float red[2][4][CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * N;
  for (int i = tid; i < 9 * Cin * CO; i += 256) {   // wl[(tap*Cin+ci)*16+co] = w[co][ci][tap]
    int co = i % CO, r = i / CO, ci = r % Cin, tap = r / Cin;
    wl[i] = w[(co * Cin + ci) * 9 + tap];
  }
  f32x4 bq[4], s1[4], s2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bq[q] = *reinterpret_cast<const f32x4*>(bias + 4 * q);
    s1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[q] = s1[q];
  }
  const int ly = tid / T, lx = tid % T;
  for (int wk = blockIdx.x; wk < nwork; wk += gridDim.x) {
    const int n = wk / ntiles, tile = wk % ntiles;
    const int ty0 = (tile / tiles_x) * T, tx0 = (tile % tiles_x) * T;
    __syncthreads();          // previous tile's reads of tin are done (and wl is complete before the first use)
    for (int i = tid; i < Cin * TP * TP; i += 256) {
      int ci = i / (TP * TP), pix = i % (TP * TP);
      int gy = ty0 + pix / TP - 1, gx = tx0 + pix % TP - 1;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x.z[(long)n * x.sn + (long)ci * x.sc + (long)gy * x.sy + (long)gx * x.sx];
      tin[ci][pix] = v;
    }
    __syncthreads();
    f32x4 acc[4] = {bq[0], bq[1], bq[2], bq[3]};
    for (int ci = 0; ci < Cin; ++ci)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float a = tin[ci][(ly + tap / 3) * TP + lx + tap % 3];
        const f32x4* wr = reinterpret_cast<const f32x4*>(wl + (tap * Cin + ci) * CO);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += a * wr[q];
      }
    const int gy = ty0 + ly, gx = tx0 + lx;
    if (gy < H && gx < W) {
      f32x4* o = reinterpret_cast<f32x4*>(out + ((long)(n * H + gy) * W + gx) * CO);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o[q] = acc[q];
        s1[q] += acc[q];
        s2[q] += acc[q] * acc[q];
      }
    }
  }
  if (stat || sacc) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = s1[q][j], v2 = s2[q][j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          v += __shfl_xor(v, o);
          v2 += __shfl_xor(v2, o);
        }
        if (lane == 0) {
          red[0][wave][4 * q + j] = v;
          red[1][wave][4 * q + j] = v2;
        }
      }
    __syncthreads();
    if (tid < 2 * CO) {
      int which = tid / CO, co = tid % CO;
      float t = red[which][0][co] + red[which][1][co] + red[which][2][co] + red[which][3][co];
      if (sacc) hpfg_acc_add(sacc, CO, (int)blockIdx.x & (q.stat_shards - 1), which, co, t);
      if (stat) stat[((long)blockIdx.x * 2 + which) * CO + co] = t;
    }
  }
}

// ---- the 1- / 3-channel first layer on the matrix cores -------------------------------------------------------------------------------------
// K = 9 taps (x 3 channels: 27) is three (seven) k-steps of v_mfma_f32_16x16x4_f32 (exact fp32, an fmaf chain in tap order -- the arithmetic of the loop above): the
// 9 x 16 weights are the A operand (M = output channel; three VGPRs per lane for the whole launch), the B operand is gathered from the 18 x 18
// input tile in LDS (N = 16 pixels of one tile row, k = tap: one ds_read_b32 per k-step), and a lane ends up with 4 consecutive output
// channels of one pixel -- a 16-byte store.  The VALU form reads its weights as 36 broadcast ds_read_b128 per pixel and spends half of its
// cycles in the LDS pipe; this one issues 3 four-byte gathers per 16 pixels.  The next tile's input is in flight (registers) while a tile
// multiplies.  Same partial-sum rows (one per workgroup) and the same output as conv_first_kernel.
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(HpfgFirstConvArgs q, int N, int H, int W, int tiles_x, int tiles_y) {
  constexpr int T = 16, TP = T + 2, CO = 16, NPIX = TP * TP, KT = 9 * CIN, KS = (KT + 3) / 4, NLD = (CIN * NPIX + 255) / 256;
  const HpfgAct& x = q.x;
  const float* __restrict__ w = q.w_oihw;
  float* __restrict__ out = q.out;
  float* __restrict__ stat = q.stat_partials;
  long long* __restrict__ sacc = q.stat_acc;
  __shared__ float tin[CIN * NPIX];
  __shared__ float red[2][4][CO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * N;
  const int co_a = lane & 15, kq = lane >> 4;                 // A: row = output channel, k-local = lane >> 4; B: column = pixel, k-local = lane >> 4
  // k = ci * 9 + tap, ascending: the order of the VALU loop (for ci: for tap), so the fmaf chain of the MFMA reproduces its sums
  float wa[KS];
  int boff[KS];
#pragma unroll
  for (int s_ = 0; s_ < KS; ++s_) {
    const int k = 4 * s_ + kq, ci = k / 9, tap = k - 9 * ci;
    wa[s_] = k < KT ? w[co_a * KT + k] : 0.f;                // W[co][ci][tap]
    boff[s_] = k < KT ? ci * NPIX + (tap / 3) * TP + (lane & 15) + tap % 3 : 0;      // (k >= KT in the last k-step is padding: its B value is forced to 0)
  }
  const f32x4 b4 = *reinterpret_cast<const f32x4*>(q.bias + 4 * kq);          // D rows 4 * (lane >> 4) + j
  // BatchNorm partial sums in the VALU form's association, so that both forms hand the finalize bit-identical rows: there a thread owns the
  // pixel (ly, lx) of every tile and the wave butterfly adds lanes (ly % 4, lx) as ((0 + 2) + (1 + 3)), then over lx by xor 8, 4, 2, 1
  f32x4 s1[4], s2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s1[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[r] = s1[r];
  }
  // tile coordinates are derived once per tile (two run-time divisions), for the tile whose input is being fetched; the multiply phase of the
  // next iteration inherits them
  int pci[NLD], ppy[NLD], ppx[NLD];          // staging piece i of this thread: element tid + 256 i of the [CIN][18][18] tile
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i, pix = e % NPIX;
    pci[i] = e / NPIX;
    ppy[i] = pix / TP - 1;
    ppx[i] = pix % TP - 1;
  }
  int n = 0, ty0 = 0, tx0 = 0;
  auto coords = [&](int wk) {
    n = wk / ntiles;
    const int tile = wk - n * ntiles, tyi = tile / tiles_x;
    ty0 = tyi * T;
    tx0 = (tile - tyi * tiles_x) * T;
  };
  auto fetch = [&](float (&v)[NLD]) {
    const float* xb = x.z + (long)n * x.sn;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int gy = ty0 + ppy[i], gx = tx0 + ppx[i];
      v[i] = (tid + 256 * i < CIN * NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? xb[(long)pci[i] * x.sc + (long)gy * x.sy + (long)gx * x.sx] : 0.f;
    }
  };
  float v[NLD];
  int wk = blockIdx.x;
  if (wk < nwork) {
    coords(wk);
    fetch(v);
  }
#pragma unroll 1
  for (; wk < nwork; wk += gridDim.x) {
    const int cn = n, cy0 = ty0, cx0 = tx0;          // this tile
    __syncthreads();          // the previous tile's gathers are done
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (tid + 256 * i < CIN * NPIX) tin[tid + 256 * i] = v[i];
    __syncthreads();
    if (wk + (int)gridDim.x < nwork) {
      coords(wk + gridDim.x);
      fetch(v);
    }
    const int gx = cx0 + (lane & 15);
    float* orow = out + ((long)(cn * H + cy0 + 4 * wave) * W + gx) * CO + 4 * kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ly = 4 * wave + r;
      f32x4 acc = b4;
#pragma unroll
      for (int s_ = 0; s_ < KS; ++s_) {
        float bv = tin[ly * TP + boff[s_]];
        if (s_ == KS - 1 && 4 * s_ + kq >= KT) bv = 0.f;          // 0 * (whatever the tile holds there) must not be NaN
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_], bv, acc, 0, 0, 0);
      }
      if (cy0 + ly < H && gx < W) *reinterpret_cast<f32x4*>(orow + (long)r * W * CO) = acc;
      s1[r] += acc;          // unconditionally: statistics are only requested for sizes that are whole tiles (conv_first_impl), and a
      s2[r] += acc * acc;    // conditional update of the accumulator arrays makes the compiler copy all 32 registers around each one
    }
  }
  if (stat || sacc) {
    const f32x4 t1 = (s1[0] + s1[2]) + (s1[1] + s1[3]), t2 = (s2[0] + s2[2]) + (s2[1] + s2[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a1 = t1[j], a2 = t2[j];
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) {          // over the 16 pixel lanes of this channel quad
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
      }
      if ((lane & 15) == 0) {
        red[0][wave][4 * kq + j] = a1;
        red[1][wave][4 * kq + j] = a2;
      }
    }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid / CO, co = tid % CO;
      const float t = red[which][0][co] + red[which][1][co] + red[which][2][co] + red[which][3][co];
      if (sacc) hpfg_acc_add(sacc, CO, (int)blockIdx.x & (q.stat_shards - 1), which, co, t);
      if (stat) stat[((long)blockIdx.x * 2 + which) * CO + co] = t;
    }
  }
}

inline int conv_first_grid(int N, int H, int W) {
  const long nwork = (long)N * ((H + 15) / 16) * ((W + 15) / 16);
  return (int)(nwork < 1024 ? nwork : 1024);
}

}  // namespace

extern "C" int hpfg_conv_stat_blocks(int N, int H, int W) {
  if (tile_is_big(H, W)) return (N * (H / 16) * (W / 16) + 7) / 8 * 8;      // (conv_thin_kernel launches a multiple of 8 workgroups, one row each)
  const int a = N * ((H + 7) / 8) * ((W + 7) / 8), b = N * ((H + 3) / 4) * ((W + 15) / 16);   // 8x8 (fp32 path) / 4x16 (bf16x3 3x3 path)
  return a > b ? a : b;
}

extern "C" int hpfg_conv_first_rows(int N, int H, int W) { return conv_first_grid(N, H, W); }

static int conv_fwd_impl(const HpfgConvArgs* a, void* stream, int* rows_only);

extern "C" int hpfg_conv_fwd(const HpfgConvArgs* a, void* stream) { return conv_fwd_impl(a, stream, nullptr); }

// rows of stat_partials ([rows][2][CoutPad]) that hpfg_conv_fwd(args) fills; <0 on argument errors
extern "C" int hpfg_conv_stat_rows(const HpfgConvArgs* a) {
  int rows = -1;
  int rc = conv_fwd_impl(a, nullptr, &rows);
  return rc ? -1 : rows;
}

static int conv_fwd_impl(const HpfgConvArgs* a, void* stream, int* rows_only) {
  HPFG_ARG_CHECK(a && a->wpk && a->out, "conv_fwd: null pointer");
  HPFG_ARG_CHECK(a->taps == 9 || a->taps == 1, "conv_fwd: taps must be 1 or 9 (got %d)", a->taps);
  HPFG_ARG_CHECK(a->CoutPad % 16 == 0 && a->Cout <= a->CoutPad && a->Cout > 0, "conv_fwd: bad Cout %d / pad %d", a->Cout, a->CoutPad);
  HPFG_ARG_CHECK(a->N > 0 && a->H > 0 && a->W > 0 && a->N < 65536, "conv_fwd: bad N/H/W");
  HPFG_ARG_CHECK(a->a0.mode != HPFG_ACT_NONE && a->a0.C > 0, "conv_fwd: a0 empty");
  HPFG_ARG_CHECK(a->a1.mode == HPFG_ACT_NONE || a->a0.C % 16 == 0, "conv_fwd: concat needs a0.C %% 16 == 0");
  HPFG_ARG_CHECK(a->out_split ? (a->out_pstride >= a->out_split) : (a->out_pstride >= a->Cout), "conv_fwd: out_pstride too small");
  HPFG_ARG_CHECK(a->out_split == 0 || (a->out2 && a->out_split % 16 == 0 && a->out_split < a->Cout && a->out2_pstride >= a->Cout - a->out_split &&
                                       !a->bwd_stats),
                 "conv_fwd: out_split needs out2, a multiple of 16 below Cout, and no bwd_stats");
  HPFG_ARG_CHECK(!a->stat_acc || ((a->math & 0xff) == HPFG_MATH_BF16X3 && a->Cout == a->CoutPad && a->stat_shards >= 1 &&
                                  a->stat_shards <= HPFG_ACC_MAX_SHARDS && (a->stat_shards & (a->stat_shards - 1)) == 0),
                 "conv_fwd: stat_acc is a bf16x3 forward feature (Cout == CoutPad, stat_shards a power of two <= %d)", HPFG_ACC_MAX_SHARDS);
  for (const HpfgAct* s : {&a->a0, &a->a1}) {
    if (!s->bn_acc) continue;
    HPFG_ARG_CHECK((a->math & 0xff) == HPFG_MATH_BF16X3 && (s->mode == HPFG_ACT_BNACT || s->mode == HPFG_ACT_BNACT_POOL || s->mode == HPFG_ACT_DZ) &&
                       s->bn_gamma && (s->bn_beta || s->mode == HPFG_ACT_DZ) && s->bn_count >= 1.f && s->bn_count <= 16777216.f && s->C <= 256 && s->bn_shards >= 1 &&
                       s->bn_shards <= HPFG_ACC_MAX_SHARDS,
                   "conv_fwd: bn_acc needs a BNACT / BNACT_POOL / DZ source of the bf16x3 kernels with gamma (beta), a count of at most 2^24 (a float holds N*H*W exactly up to there) and at most 256 channels");
  }
  HPFG_ARG_CHECK((a->math & 0xff) != HPFG_MATH_BF16X3 || !(a->a0.mode == HPFG_ACT_BNACT || a->a0.mode == HPFG_ACT_BNACT_POOL || a->a0.mode == HPFG_ACT_DZ) ||
                     a->a0.C <= 256,
                 "conv_fwd(bf16x3): a BatchNorm'd source has at most 256 channels (got %d)", a->a0.C);
  const bool upb = a->a0.mode == HPFG_ACT_UPBWD;
  HPFG_ARG_CHECK(!upb || ((a->math & 0xff) == HPFG_MATH_BF16X3 && a->taps == 1 && a->a1.mode == HPFG_ACT_NONE && a->a0.C % 8 == 0 && a->a0.pstride % 4 == 0 &&
                          a->a0.Hs == a->H && a->a0.Ws == a->W && a->H <= 16383 && a->W <= 16383 && (long)a->N * 4 * a->H * a->W * a->a0.pstride < (1L << 31)),
                 "conv_fwd: an UPBWD source belongs to the 1x1 bf16x3 dgrad (channels a multiple of 8, pixel stride of 4, Hs x Ws == H x W)");
  HPFG_ARG_CHECK(!a->side_sums || upb, "conv_fwd: side_sums goes with an UPBWD source");
  HPFG_ARG_CHECK(!a->stage_out || upb || ((a->math & 0xff) == HPFG_MATH_BF16X3 && a->taps == 9 && hpfg_kind_of(a->a0, a->a1) > HPFG_KIND_PLAIN &&
                                    (a->a0.C + a->a1.C) % 8 == 0 && (a->H % 16 || a->W % 16)),
                 "conv_fwd: stage_out is a feature of the 3x3 bf16x3 kernels (non-PLAIN source, channels a multiple of 8, H or W not a multiple of 16)");
  hipStream_t st = (hipStream_t)stream;
  if (a->bwd_stats) {
    const int kind = hpfg_kind_of(a->a0, a->a1);
    HPFG_ARG_CHECK((a->math & 0xff) == HPFG_MATH_BF16X3 && (kind == HPFG_KIND_DZ || kind == HPFG_KIND_PLAIN || (kind == HPFG_KIND_UPB && a->bwd_stats == 1)),
                   "conv_fwd: bwd_stats is a dgrad feature of the bf16x3 kernels (DZ, PLAIN or UPBWD source)");
    HPFG_ARG_CHECK((a->stat_partials || a->stat_acc) && a->bwd_of.z && a->bwd_of.bn && !a->bias,
                   "conv_fwd: bwd_stats needs stat_partials or stat_acc, bwd_of.z / .bn and no bias");
    const int up = a->bwd_stats == 2 ? 2 : 1;      // 2: `out` is the gradient w.r.t. MaxPool2d(2) of bwd_of's activation (bwd_of at twice the size)
    HPFG_ARG_CHECK(a->bwd_stats == 1 || a->bwd_stats == 2, "conv_fwd: bwd_stats must be 0, 1 or 2");
    HPFG_ARG_CHECK(a->bwd_of.C == a->Cout && a->Cout == a->CoutPad && a->bwd_of.Hs == up * a->H && a->bwd_of.Ws == up * a->W && a->bwd_of.pstride % 4 == 0,
                   "conv_fwd: bwd_of must describe a layer with C == Cout == CoutPad (%d/%d/%d) at the output size (twice it for bwd_stats == 2)",
                   a->bwd_of.C, a->Cout, a->CoutPad);
    HPFG_ARG_CHECK(a->bwd_stats == 1 || (a->taps == 9 && (a->H % 16 || a->W % 16) && a->bwd_of.aux && a->bwd_of.aux_pstride % 4 == 0 && a->bwd_of.drop_p == 0.f && !a->out_split),
                   "conv_fwd: bwd_stats == 2 (max-pool backward in the epilogue) needs a 3x3 dgrad at a size that is not a multiple of 16, bwd_of.aux (the gradient so far) and no dropout behind bwd_of");
  }
  if ((a->math & 0xff) == HPFG_MATH_BF16X3) {
    {      // the thin 16-pixel-aligned layers have a kernel of their own
      const int r = hpfg_conv_thin_try(*a, st, rows_only);
      if (r != HPFG_THIN_NONE) return r;
    }
    switch (hpfg_kind_of(a->a0, a->a1)) {
      case HPFG_KIND_PLAIN: return hpfg_conv16_launch_plain(*a, st, rows_only);
      case HPFG_KIND_BNACT: return hpfg_conv16_launch_bnact(*a, st, rows_only);
      case HPFG_KIND_POOL: return hpfg_conv16_launch_pool(*a, st, rows_only);
      case HPFG_KIND_CAT: return hpfg_conv16_launch_cat(*a, st, rows_only);
      case HPFG_KIND_DZ: return hpfg_conv16_launch_dz(*a, st, rows_only);
      case HPFG_KIND_UPB: return hpfg_conv16_launch_upb(*a, st, rows_only);
      default: break;
    }
    hpfg_set_error("conv_fwd(bf16x3): unsupported source combination (a0.mode=%d, a1.mode=%d)", a->a0.mode, a->a1.mode);
    return -1;
  }
  if (rows_only) {   // fp32 kernels: one row per 16x16 (or 8x8) tile
    const int tl = tile_is_big(a->H, a->W) ? 16 : 8;
    *rows_only = a->N * ((a->H + tl - 1) / tl) * ((a->W + tl - 1) / tl);
    return 0;
  }
  switch (hpfg_kind_of(a->a0, a->a1)) {
    case HPFG_KIND_PLAIN: return hpfg_conv_launch_plain(*a, st);
    case HPFG_KIND_BNACT: return hpfg_conv_launch_bnact(*a, st);
    case HPFG_KIND_POOL: return hpfg_conv_launch_pool(*a, st);
    case HPFG_KIND_CAT: return hpfg_conv_launch_cat(*a, st);
    case HPFG_KIND_DZ: return hpfg_conv_launch_dz(*a, st);
    default: break;
  }
  hpfg_set_error("conv_fwd: unsupported source combination (a0.mode=%d, a1.mode=%d)", a->a0.mode, a->a1.mode);
  return -1;
}

static int conv_first_impl(const HpfgFirstConvArgs* q, int N, int H, int W, int Cin, int Cout, void* stream) {
  HPFG_ARG_CHECK(Cin >= 1 && Cin <= 4 && Cout == 16, "conv_first: needs Cin<=4, Cout==16 (got %d,%d)", Cin, Cout);
  HPFG_ARG_CHECK(q->x.z && q->w_oihw && q->bias && q->out, "conv_first: null pointer");
  HPFG_ARG_CHECK(q->x.mode == HPFG_ACT_STRIDED, "conv_first: input must be a STRIDED source");
  // stat partial layout must match hpfg_conv_stat_blocks(): 16x16 tiles only when H,W are multiples of 16
  HPFG_ARG_CHECK((q->stat_partials == nullptr && q->stat_acc == nullptr) || tile_is_big(H, W), "conv_first: BN sums need H,W multiples of 16 (got %dx%d)", H, W);
  int tx = (W + 15) / 16, ty = (H + 15) / 16;
  const dim3 g1(conv_first_grid(N, H, W));
  if (q->stat_acc) HPFG_ACC_CHECK(g1.x, q->stat_shards, "conv_first");
  if ((Cin == 1 || Cin == 3) && hpfg_opt(HPFG_OPT_FIRST_MFMA) != 0) {      // grey-scale (ACDC / LIDC) and RGB (CPS config) inputs (option 0: the VALU form, A/B runs)
    if (Cin == 1) hipLaunchKernelGGL((conv_first_mfma_kernel<1>), g1, dim3(256), 0, (hipStream_t)stream, *q, N, H, W, tx, ty);
    else hipLaunchKernelGGL((conv_first_mfma_kernel<3>), g1, dim3(256), 0, (hipStream_t)stream, *q, N, H, W, tx, ty);
    return hpfg_launch_status("conv_first_mfma_kernel");
  }
  hipLaunchKernelGGL(conv_first_kernel, g1, dim3(256), 0, (hipStream_t)stream, *q, N, H, W, Cin, tx, ty);
  return hpfg_launch_status("conv_first_kernel");
}

extern "C" int hpfg_conv3x3_first_fwd(const HpfgAct* x, const float* w_oihw, const float* bias, float* out, float* stat_partials,
                                      int N, int H, int W, int Cin, int Cout, void* stream) {
  HPFG_ARG_CHECK(x, "conv_first: null pointer");
  HpfgFirstConvArgs a = {*x, w_oihw, bias, out, stat_partials, nullptr, 1};
  return conv_first_impl(&a, N, H, W, Cin, Cout, stream);
}

extern "C" int hpfg_conv3x3_first_fwd_acc(const HpfgAct* x, const float* w_oihw, const float* bias, float* out, float* stat_partials, long long* stat_acc,
                                          int stat_shards, int N, int H, int W, int Cin, int Cout, void* stream) {
  HPFG_ARG_CHECK(x && stat_acc, "conv_first_acc: null pointer");
  HPFG_ARG_CHECK(stat_shards >= 1 && stat_shards <= HPFG_ACC_MAX_SHARDS && (stat_shards & (stat_shards - 1)) == 0, "conv_first_acc: bad shard count %d", stat_shards);
  HpfgFirstConvArgs a = {*x, w_oihw, bias, out, stat_partials, stat_acc, stat_shards};
  return conv_first_impl(&a, N, H, W, Cin, Cout, stream);
}

