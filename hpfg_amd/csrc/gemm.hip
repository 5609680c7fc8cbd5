// Small dense GEMM in exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain, MI355X_MICROARCH.md "FP32-input
// MFMA"), with arbitrary element strides on both operands so that one kernel serves  Y = X W^T + b (ReLU) ,  dX = dY W ,  dW = dY^T X
// of the projection necks (reference model/unet.py:125-138: Linear / 1x1 conv stacks) and the Gram matrices of Dense_Loss
// (utils/loss/dense_loss.py:24).  These products are a few hundred MFLOP per step: what matters is that they are ours (no rocBLAS /
// MIOpen on the path), exact, deterministic (no split-K, no atomics) and capturable.
//
//   C[m, n] = act( sum_k A(m, k) * B(k, n) + bias[n] ),   A(m, k) = A[m*sam + k*sak],   B(k, n) = B[k*sbk + n*sbn],   C row-major (ldc)
//
// Workgroup = 256 threads = 4 waves, 64 x 64 output tile (each wave 32 x 32 = 2 x 2 MFMA tiles), K in chunks of 16 through LDS.
// LDS images are k-major with a row stride of 80 floats: the four k rows one MFMA operand read touches start 16 banks apart.
#include "common.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, LDT = 80;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  long sam, sak, sbk, sbn, ldc;
  int M, N, K, relu, accumulate;
};

// blockIdx.z = K split: split z covers k in [z * kper, (z + 1) * kper) and, when there are several splits, writes its partial product
// (no bias / activation) to C + z * M * ldc; gemm_splitk_finish_kernel adds them in a fixed order.
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p, int kper) {
  __shared__ float As[TK * LDT], Bs[TK * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper < p.K ? kbeg + kper : p.K;
  const bool split = gridDim.z > 1;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging map: consecutive threads walk the operand's unit-stride index (coalesced for either orientation)
  const bool a_k_fast = p.sak == 1, b_n_fast = p.sbn == 1;
  int am[4], ak[4], bn_[4], bk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (a_k_fast) {
      ak[i] = tid & 15;
      am[i] = (tid >> 4) + 16 * i;
    } else {
      am[i] = tid & 63;
      ak[i] = (tid >> 6) + 4 * i;
    }
    if (b_n_fast) {
      bn_[i] = tid & 63;
      bk[i] = (tid >> 6) + 4 * i;
    } else {
      bk[i] = tid & 15;
      bn_[i] = (tid >> 4) + 16 * i;
    }
  }
  float ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + am[i], gk = k0 + ak[i], gn = n0 + bn_[i], gkb = k0 + bk[i];
      ra[i] = (gm < p.M && gk < kend) ? p.A[(long)gm * p.sam + (long)gk * p.sak] : 0.f;
      rb[i] = (gn < p.N && gkb < kend) ? p.B[(long)gkb * p.sbk + (long)gn * p.sbn] : 0.f;
    }
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[ak[i] * LDT + am[i]] = ra[i];
      Bs[bk[i] * LDT + bn_[i]] = rb[i];
    }
    __syncthreads();
    if (k0 + TK < kend) fetch(k0 + TK);          // the next chunk's loads fly across the MFMA phase
#pragma unroll
    for (int ks = 0; ks < TK / 4; ++ks) {
      const int kk = ks * 4 + (lane >> 4);
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk * LDT + wm + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kk * LDT + wn + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // D layout: row = (lane >> 4) * 4 + r, col = lane & 15
  float* Cz = p.C + (split ? (long)blockIdx.z * p.M * p.ldc : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn + j * 16 + (lane & 15);
      if (gn >= p.N) continue;
      const float bv = (p.bias && !split) ? p.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        if (gm >= p.M) continue;
        float v = acc[i][j][r] + bv;
        float* c = Cz + (long)gm * p.ldc + gn;
        if (!split) {
          if (p.accumulate) v += *c;
          if (p.relu) v = fmaxf(v, 0.f);
        }
        *c = v;
      }
    }
}

// out[m][n] = act(sum_s part[s][m][n] + bias[n] (+ out[m][n])): the K splits of gemm_f32_kernel in a fixed order
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const float* __restrict__ part, int S, int M, int N, long ldp, float* __restrict__ out, long ldc,
                                                                 const float* __restrict__ bias, int relu, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * N) return;
  const int m = (int)(i / N), n = (int)(i % N);
  float v = bias ? bias[n] : 0.f;
  for (int s = 0; s < S; ++s) v += part[((long)s * M + m) * ldp + n];
  float* c = out + (long)m * ldc + n;
  if (accumulate) v += *c;
  if (relu) v = fmaxf(v, 0.f);
  *c = v;
}

inline int f32_splits(int M, int N, int K) {
  const long tiles = (long)((M + TM - 1) / TM) * ((N + TN - 1) / TN);
  if (tiles >= 128 || K < 256) return 1;          // enough workgroups already, or nothing to split
  long s = 256 / tiles;
  const long by_k = K / 128;                      // at least 128 of K per split
  if (s > by_k) s = by_k;
  return (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
}

// out[c] = sum_r x[r, c]  (bias gradients of the neck layers): one thread per column, rows in a fixed order
__global__ void col_sum_kernel(const float* x, long R, int M, long ldx, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= M) return;
  float s = 0.f;
  for (long r = 0; r < R; ++r) s += x[r * ldx + c];
  out[c] = s;
}

// dy *= (y > 0)   (ReLU backward on the stored post-activation)
__global__ void relu_bwd_kernel(float* dy, const float* y, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dy[i] = y[i] > 0.f ? dy[i] : 0.f;
}

}  // namespace

extern "C" int hpfg_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* Cm, long ldc, int M, int N, int K,
                             const float* bias, int relu, int accumulate, void* stream) {
  HPFG_ARG_CHECK(A && B && Cm && M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_f32: bad args (M=%d N=%d K=%d ldc=%ld)", M, N, K, ldc);
  GemmArgs p{A, B, Cm, bias, sam, sak, sbk, sbn, ldc, M, N, K, relu, accumulate};
  dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, K);
  return hpfg_launch_status("gemm_f32_kernel");
}

/* K splits for a product with few output tiles and a long contraction (the 2048-wide neck layers: 2 x 8 tiles, K = 2048) */
extern "C" int hpfg_gemm_f32_splits(int M, int N, int K) { return f32_splits(M, N, K); }

/* hpfg_gemm_f32 with the contraction split over workgroups; scratch: hpfg_gemm_f32_splits(M, N, K) * M * N floats (unused when 1 split) */
extern "C" int hpfg_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* Cm, long ldc, int M, int N, int K,
                                    const float* bias, int relu, int accumulate, float* scratch, void* stream) {
  const int S = f32_splits(M, N, K);
  if (S <= 1) return hpfg_gemm_f32(A, sam, sak, B, sbk, sbn, Cm, ldc, M, N, K, bias, relu, accumulate, stream);
  HPFG_ARG_CHECK(A && B && Cm && scratch && M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_f32_splitk: bad args");
  int kper = (K + S - 1) / S;
  kper = (kper + TK - 1) / TK * TK;
  GemmArgs p{A, B, scratch, nullptr, sam, sak, sbk, sbn, (long)N, M, N, K, 0, 0};
  dim3 grid((N + TN - 1) / TN, (M + TM - 1) / TM, (K + kper - 1) / kper);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, kper);
  const long total = (long)M * N;
  hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scratch, (int)grid.z, M, N, (long)N, Cm,
                     ldc, bias, relu, accumulate);
  return hpfg_launch_status("gemm_f32_splitk");
}

extern "C" int hpfg_col_sum(const float* x, long R, int M, long ldx, float* out, void* stream) {
  HPFG_ARG_CHECK(x && out && R > 0 && M > 0 && ldx >= M, "col_sum: bad args");
  hipLaunchKernelGGL(col_sum_kernel, dim3((M + 127) / 128), dim3(128), 0, (hipStream_t)stream, x, R, M, ldx, out);
  return hpfg_launch_status("col_sum_kernel");
}

extern "C" int hpfg_relu_bwd(float* dy, const float* y, long n, void* stream) {
  HPFG_ARG_CHECK(dy && y && n > 0, "relu_bwd: bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, n);
  return hpfg_launch_status("relu_bwd_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") GEMM for the token-layout layers of the SegFormer branch (reference model/segformer.py: every nn.Linear, the
// kernel == stride spatial-reduction conv, the patch embeddings after im2col, the head's 1x1 convs): fp32 operands are split as
// x = hi + lo (bf16 each) while they are staged into LDS and every product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation -- the arithmetic of conv_bf16_kernel.h, ~2^-17 relative per product at a sixth of the fp32-MFMA cycles.
// Same contract as hpfg_gemm_f32 (explicit element strides, so Y = X W^T + b, dX = dY W and dW = dY^T X are one kernel), plus the
// requirement that each operand has a unit stride along m/n or k and 4-element alignment (checked by the host wrapper, which falls back
// to hpfg_gemm_f32 otherwise).
//   Workgroup = 256 threads, 128 x 128 output tile (wave w: rows (w & 1) * 64.., cols (w >> 1) * 64..: 4 x 4 MFMA tiles), K chunks of 32.
//   LDS image per operand: [128 rows][32 k] bf16, hi and lo planes, row stride 80 B (16 rows of a fragment read start on distinct banks).
//   The MFMA is issued as D = B-fragment x A-fragment, so a lane ends up with 4 consecutive n of one m: 16-byte stores.
//   Global loads of chunk c+1 are in flight (registers) while chunk c multiplies.
namespace {

typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 g_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BK = 32, ROWB = 80;               // bytes per LDS row (64 B of bf16 + 16 B pad)
constexpr int PLANE_B = BM * ROWB;                        // one hi or lo plane of one operand

__device__ __forceinline__ void split4(const f32x4& v, g_bf16x4& hi, g_bf16x4& lo) {
  const g_bf16x2 h0 = __builtin_convertvector(g_f32x2{v[0], v[1]}, g_bf16x2), h1 = __builtin_convertvector(g_f32x2{v[2], v[3]}, g_bf16x2);
  const uint32_t w0 = __builtin_bit_cast(uint32_t, h0), w1 = __builtin_bit_cast(uint32_t, h1);
  const g_f32x2 f0 = {__builtin_bit_cast(float, w0 << 16), __builtin_bit_cast(float, w0 & 0xFFFF0000u)};
  const g_f32x2 f1 = {__builtin_bit_cast(float, w1 << 16), __builtin_bit_cast(float, w1 & 0xFFFF0000u)};
  const g_bf16x2 l0 = __builtin_convertvector(g_f32x2{v[0], v[1]} - f0, g_bf16x2), l1 = __builtin_convertvector(g_f32x2{v[2], v[3]} - f1, g_bf16x2);
  hi = g_bf16x4{h0[0], h0[1], h1[0], h1[1]};
  lo = g_bf16x4{l0[0], l0[1], l1[0], l1[1]};
}

// One operand tile: `rows` index m (or n), element (row, k) at base[row * srow + k * sk]; exactly one of srow / sk is 1.
struct OpTile {
  const float* base;
  long srow, sk;
  int nrows, K;
};

// global -> registers: 4 float4 per thread.  k-fast: thread owns k quad (t & 7) of rows (t >> 3) + 32 i; row-fast: row quad (t & 31) at
// k = (t >> 5) + 8 i.  Out-of-range quads read as zero (nrows and K are multiples of 4 where the quad runs along them).
__device__ __forceinline__ void tile_load(const OpTile& o, int row0, int k0, int tid, f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (o.sk == 1) {
      const int row = row0 + (tid >> 3) + 32 * i, k = k0 + 4 * (tid & 7);
      if (row < o.nrows && k < o.K) r[i] = *reinterpret_cast<const f32x4*>(o.base + (long)row * o.srow + k);
    } else {
      const int row = row0 + 4 * (tid & 31), k = k0 + (tid >> 5) + 8 * i;
      if (row < o.nrows && k < o.K) r[i] = *reinterpret_cast<const f32x4*>(o.base + (long)k * o.sk + row);
    }
  }
}

constexpr int TRS2 = 288;       // row-fast operands: LDS image [32 k][128 rows] bf16, k-row stride in bytes (256 B + 32 B: the 4 k rows of a transposing read sit 8 banks apart)

__device__ __forceinline__ void tile_store(const OpTile& o, unsigned char* lds, int tid, const f32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    g_bf16x4 hi, lo;
    split4(r[i], hi, lo);
    // k-fast: [row][k] image, the fragment is a plain 16-byte read.  row-fast (the contraction index is the operand's slow index, e.g. W in
    // dX = dY W): stored as it lies in memory, [k][row], and read through ds_read_b64_tr_b16 -- no scattered 2-byte stores.
    unsigned char* p = o.sk == 1 ? lds + ((tid >> 3) + 32 * i) * ROWB + 8 * (tid & 7) : lds + ((tid >> 5) + 8 * i) * TRS2 + 8 * (tid & 31);
    *reinterpret_cast<g_bf16x4*>(p) = hi;
    *reinterpret_cast<g_bf16x4*>(p + PLANE_B) = lo;
  }
}

typedef short g2_s16x4 __attribute__((ext_vector_type(4)));
typedef short g2_s16x8 __attribute__((ext_vector_type(8)));
// fragment (16 rows starting at row0, this lane's 8 k values) of an operand image, either layout
__device__ __forceinline__ g_bf16x8 op_frag(const unsigned char* plane, bool tr, int row0, int lane) {
  if (!tr) return *reinterpret_cast<const g_bf16x8*>(plane + (row0 + (lane & 15)) * ROWB + (lane >> 4) * 16);
  const unsigned char* p = plane + (8 * (lane >> 4) + ((lane & 15) >> 2)) * TRS2 + (row0 + 4 * (lane & 3)) * 2;
  const g2_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g2_s16x4 __attribute__((address_space(3)))*)(p));
  const g2_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g2_s16x4 __attribute__((address_space(3)))*)(p + 4 * TRS2));
  return __builtin_bit_cast(g_bf16x8, (g2_s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}));
}

struct Gemm16Args {
  OpTile a, b;            // a: rows = m, b: rows = n
  float* C;
  const float* bias;
  long ldc;
  int M, N, K, relu, accumulate;
};

// blockIdx.z = K split (as gemm_f32_kernel): split z covers k in [z * kper, (z + 1) * kper), kper a multiple of 32, and with several splits writes
// its partial product (no bias / activation) to C + z * M * ldc; gemm_splitk_finish_kernel adds them in a fixed order.
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(Gemm16Args p, int kper) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * PLANE_B];      // A hi | A lo | B hi | B lo
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + 2 * PLANE_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BM;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ra[4], rb[4];
  const int kbeg = blockIdx.z * kper, kend = kbeg + kper < p.K ? kbeg + kper : p.K;
  p.a.K = kend;            // (the loaders read zeros beyond their operand's K)
  p.b.K = kend;
  p.K = kend;
  if (gridDim.z > 1) p.C += (long)blockIdx.z * p.M * p.ldc;
  tile_load(p.a, m0, kbeg, tid, ra);
  tile_load(p.b, n0, kbeg, tid, rb);
  const bool atr = p.a.sk != 1, btr = p.b.sk != 1;          // workgroup-uniform
  for (int k0 = kbeg; k0 < p.K; k0 += BK) {
    __syncthreads();            // the previous chunk's fragment reads are done
    tile_store(p.a, ldsA, tid, ra);
    tile_store(p.b, ldsB, tid, rb);
    __syncthreads();
    if (k0 + BK < p.K) {        // next chunk's global loads fly across the MFMA phase
      tile_load(p.a, m0, k0 + BK, tid, ra);
      tile_load(p.b, n0, k0 + BK, tid, rb);
    }
    g_bf16x8 bh[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = op_frag(ldsB, btr, wn + 16 * j, lane);
      bl[j] = op_frag(ldsB + PLANE_B, btr, wn + 16 * j, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const g_bf16x8 ah = op_frag(ldsA, atr, wm + 16 * i, lane);
      const g_bf16x8 al = op_frag(ldsA + PLANE_B, atr, wm + 16 * i, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
      }
    }
  }
  // D[n = (lane >> 4) * 4 + r][m = lane & 15]
  const bool vec = (p.N & 3) == 0 && (p.ldc & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + wm + 16 * i + (lane & 15);
    if (gm >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + wn + 16 * j + (lane >> 4) * 4;
      if (gn >= p.N) continue;
      f32x4 v = acc[i][j];
      float* c = p.C + (long)gm * p.ldc + gn;
      if (vec) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + gn);
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(c);
        if (p.relu) v = __builtin_elementwise_max(v, f32x4{0.f, 0.f, 0.f, 0.f});
        *reinterpret_cast<f32x4*>(c) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (gn + r >= p.N) break;
          float t = v[r] + (p.bias ? p.bias[gn + r] : 0.f);
          if (p.accumulate) t += c[r];
          if (p.relu) t = fmaxf(t, 0.f);
          c[r] = t;
        }
      }
    }
  }
}

inline bool op_ok(const float* base, long srow, long sk, int nrows, int K) {
  if (((uintptr_t)base & 15) != 0) return false;
  if (sk == 1) return (K & 3) == 0 && (srow & 3) == 0;
  if (srow == 1) return (nrows & 3) == 0 && (sk & 3) == 0;
  return false;
}

}  // namespace

// 1 if hpfg_gemm_bf16x3 accepts these operands (unit stride along one index of each, 4-element alignment), else 0
extern "C" int hpfg_gemm_bf16x3_ok(const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N, int K) {
  return op_ok(A, sam, sak, M, K) && op_ok(B, sbn, sbk, N, K) ? 1 : 0;
}

extern "C" int hpfg_gemm_bf16x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* Cm, long ldc, int M, int N, int K,
                                const float* bias, int relu, int accumulate, void* stream) {
  HPFG_ARG_CHECK(A && B && Cm && M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_bf16x3: bad args (M=%d N=%d K=%d ldc=%ld)", M, N, K, ldc);
  HPFG_ARG_CHECK(hpfg_gemm_bf16x3_ok(A, sam, sak, B, sbk, sbn, M, N, K), "gemm_bf16x3: operands need a unit stride and 4-element alignment");
  Gemm16Args p{{A, sam, sak, M, K}, {B, sbn, sbk, N, K}, Cm, bias, ldc, M, N, K, relu, accumulate};
  dim3 grid((N + BM - 1) / BM, (M + BM - 1) / BM);
  hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, (K + BK - 1) / BK * BK);
  return hpfg_launch_status("gemm_bf16x3_kernel");
}

/* K splits for a product with few 128 x 128 output tiles and a long contraction -- the 1568-token layers of the SegFormer branch (stage 4, and
 * the keys / values after the spatial reduction: 13 row tiles, K up to 2048): one workgroup per tile walked its 64 chunks alone (121 us for
 * 13 MB of operands).  1 = do not split. */
extern "C" int hpfg_gemm_bf16x3_splits(int M, int N, int K) {
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BM - 1) / BM);
  if (tiles >= 128 || K < 256) return 1;
  long s = 256 / tiles;
  const long by_k = K / 128;                      // at least 4 chunks of K per split
  if (s > by_k) s = by_k;
  return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}

/* hpfg_gemm_bf16x3 with the contraction split over workgroups; scratch: hpfg_gemm_bf16x3_splits(M, N, K) * M * N floats (unused when 1 split).
 * Deterministic: the partial products are added in split order by one finishing launch, which also applies bias / ReLU / accumulate. */
extern "C" int hpfg_gemm_bf16x3_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* Cm, long ldc, int M, int N, int K,
                                       const float* bias, int relu, int accumulate, float* scratch, void* stream) {
  const int S = hpfg_gemm_bf16x3_splits(M, N, K);
  if (S <= 1) return hpfg_gemm_bf16x3(A, sam, sak, B, sbk, sbn, Cm, ldc, M, N, K, bias, relu, accumulate, stream);
  HPFG_ARG_CHECK(A && B && Cm && scratch && M > 0 && N > 0 && K > 0 && ldc >= N, "gemm_bf16x3_splitk: bad args");
  HPFG_ARG_CHECK(hpfg_gemm_bf16x3_ok(A, sam, sak, B, sbk, sbn, M, N, K), "gemm_bf16x3_splitk: operands need a unit stride and 4-element alignment");
  int kper = (K + S - 1) / S;
  kper = (kper + BK - 1) / BK * BK;
  Gemm16Args p{{A, sam, sak, M, K}, {B, sbn, sbk, N, K}, scratch, nullptr, (long)N, M, N, K, 0, 0};
  dim3 grid((N + BM - 1) / BM, (M + BM - 1) / BM, (K + kper - 1) / kper);
  hipLaunchKernelGGL(gemm_bf16x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, kper);
  const long total = (long)M * N;
  hipLaunchKernelGGL(gemm_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scratch, (int)grid.z, M, N, (long)N, Cm,
                     ldc, bias, relu, accumulate);
  return hpfg_launch_status("gemm_bf16x3_splitk");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of a token-layout layer on the matrix cores:  dW[n][k] = sum_r dY[r][n] * X[r][k]   (R tokens, N x K weights)
// The contraction index r is the SLOW index of both operands.  They are staged exactly as they lie in memory ([r][n], [r][k]: coalesced
// float4 loads, 8-byte LDS stores of the hi / lo bf16 quads) and gfx950's transposing LDS read ds_read_b64_tr_b16 hands every lane the
// r-contiguous fragment v_mfma_f32_16x16x32_bf16 wants (same idiom as wgrad_bf16_kernel.h).  The rows are split over blockIdx.z; each
// split writes a partial [N][K] matrix and gemm_rows_sum_kernel adds the partials in a fixed order (deterministic, no atomics).
//   Workgroup = 256 threads, 64 (n) x 64 (k) output tile, 32 rows per step; wave w owns n-range (w & 1) * 32, k-range (w >> 1) * 32.
namespace {

typedef short g_s16x4 __attribute__((ext_vector_type(4)));
typedef short g_s16x8 __attribute__((ext_vector_type(8)));
constexpr int TR = 32, TT = 64, TRS = 144;            // rows per step, tile edge, LDS row stride in bytes (128 B of bf16 + 16 B: conflict-free tr reads)
constexpr int TPLANE = TR * TRS;

__device__ __forceinline__ g_bf16x8 tn_frag(const unsigned char* p) {      // p: this lane's address in the first 4-row block; the second is 4 rows below
  const g_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_s16x4 __attribute__((address_space(3)))*)(p));
  const g_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g_s16x4 __attribute__((address_space(3)))*)(p + 4 * TRS));
  return __builtin_bit_cast(g_bf16x8, (g_s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}));
}

// rows [r0, r0 + 32) x columns [c0, c0 + 64) of M ([R][C], leading dimension ld) -> hi / lo planes [32][64] bf16 (row stride TRS)
template <bool ALIGNED>
__device__ __forceinline__ void tn_load(const float* __restrict__ Mx, long ld, long r0, long r1, int c0, int C, int tid, f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rr = (tid >> 4) + 16 * i, c = c0 + 4 * (tid & 15);
    const long r = r0 + rr;
    v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (r < r1 && c < C) {
      const float* p = Mx + r * ld + c;
      if (ALIGNED) {
        v[i] = *reinterpret_cast<const f32x4*>(p);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (c + j < C) v[i][j] = p[j];
      }
    }
  }
}
__device__ __forceinline__ void tn_store(unsigned char* lds, int tid, const f32x4 (&v)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    g_bf16x4 hi, lo;
    split4(v[i], hi, lo);
    unsigned char* p = lds + ((tid >> 4) + 16 * i) * TRS + 8 * (tid & 15);
    *reinterpret_cast<g_bf16x4*>(p) = hi;
    *reinterpret_cast<g_bf16x4*>(p + TPLANE) = lo;
  }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void gemm_tn_bf16x3_kernel(const float* __restrict__ dy, long ldy, const float* __restrict__ x, long ldx,
                                                             float* __restrict__ part, long R, int N, int K, int rows_per_split, int with_db,
                                                             long part_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * TPLANE];      // dY hi | dY lo | X hi | X lo
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + 2 * TPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * TT, k0 = blockIdx.y * TT;
  const int wn = (wave & 1) * 32, wk = (wave >> 1) * 32;
  const long r0 = (long)blockIdx.z * rows_per_split, r1 = r0 + rows_per_split < R ? r0 + rows_per_split : R;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // transposed-read address of this lane: k-group g = lane >> 4 covers rows 8g .. 8g+7; lane 4q + p of the group addresses row 8g + q, columns 4p ..
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int fr = (8 * g + q) * TRS + 8 * pp;
  f32x4 va[2], vb[2];
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};          // with_db: column sums of this thread's dY column quad (the bias gradient), blockIdx.y == 0 only
  const bool do_db = with_db && blockIdx.y == 0;
  tn_load<ALIGNED>(dy, ldy, r0, r1, n0, N, tid, va);
  tn_load<ALIGNED>(x, ldx, r0, r1, k0, K, tid, vb);
  for (long rb = r0; rb < r1; rb += TR) {
    __syncthreads();
    if (do_db) csum += va[0] + va[1];
    tn_store(ldsA, tid, va);
    tn_store(ldsB, tid, vb);
    __syncthreads();
    if (rb + TR < r1) {
      tn_load<ALIGNED>(dy, ldy, rb + TR, r1, n0, N, tid, va);
      tn_load<ALIGNED>(x, ldx, rb + TR, r1, k0, K, tid, vb);
    }
    g_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ah[i] = tn_frag(ldsA + fr + (wn + 16 * i) * 2);
      al[i] = tn_frag(ldsA + fr + (wn + 16 * i) * 2 + TPLANE);
      bh[i] = tn_frag(ldsB + fr + (wk + 16 * i) * 2);
      bl[i] = tn_frag(ldsB + fr + (wk + 16 * i) * 2 + TPLANE);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {      // D[row = k][col = n]: a lane ends up with 4 consecutive k of one n
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
      }
  }
  float* o = part + (long)blockIdx.z * part_stride;
  if (do_db) {          // 16 row-threads share a column quad: fixed-order sum through LDS; db partial sits behind the N*K weight entries
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(lds);
    red[tid] = csum;
    __syncthreads();
    if (tid < 16) {
      f32x4 t = red[tid];
      for (int k = 1; k < 16; ++k) t += red[tid + 16 * k];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + 4 * tid + j < N) o[(long)N * K + n0 + 4 * tid + j] = t[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn + 16 * i + (lane & 15);
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk + 16 * j + (lane >> 4) * 4;
      if (ALIGNED) {
        if (k < K) *reinterpret_cast<f32x4*>(o + (long)n * K + k) = acc[i][j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (k + r < K) o[(long)n * K + k + r] = acc[i][j][r];
      }
    }
  }
}

// out[o] = sum_s part[s * n + o]: lane <-> output (coalesced), the four waves take every fourth partial, fixed combination order
__global__ __launch_bounds__(256) void gemm_rows_sum_kernel(const float* __restrict__ part, int S, long n, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long o = (long)blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f;
  if (o < n) {
    int s = wave;
    for (; s + 4 < S; s += 8) {
      a0 += part[(long)s * n + o];
      a1 += part[(long)(s + 4) * n + o];
    }
    if (s < S) a0 += part[(long)s * n + o];
  }
  red[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && o < n) out[o] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// column sums, stage 1: grid (ceil(M / 64), S); a workgroup = 4 row lanes x 64 columns over its slice of the rows
__global__ __launch_bounds__(256) void col_sum_part_kernel(const float* __restrict__ x, long R, int M, long ldx, int rows_per_split, float* __restrict__ part) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * rows_per_split, r1 = r0 + rows_per_split < R ? r0 + rows_per_split : R;
  float a0 = 0.f, a1 = 0.f;
  if (c < M) {
    long r = r0 + wave;
    for (; r + 4 < r1; r += 8) {
      a0 += x[r * ldx + c];
      a1 += x[(r + 4) * ldx + c];
    }
    if (r < r1) a0 += x[r * ldx + c];
  }
  red[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && c < M) part[(long)blockIdx.y * M + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

inline int tn_splits(long R, int N, int K) {
  const long tiles = (long)((N + TT - 1) / TT) * ((K + TT - 1) / TT);
  long s = 1024 / tiles;                       // ~4 workgroups per CU in total
  const long by_rows = (R + 255) / 256;        // at least 256 rows per split
  if (s > by_rows) s = by_rows;
  return (int)(s < 1 ? 1 : (s > 512 ? 512 : s));
}

}  // namespace

extern "C" int hpfg_gemm_tn_splits(long R, int N, int K) { return tn_splits(R, N, K); }

/* dW[N][K] = dY^T X over R rows (dY [R][N], X [R][K], contiguous) and, with_db, db[N] = column sums of dY written right behind dW
 * (dw_db holds N*K (+ N) floats); partials: hpfg_gemm_tn_splits(R, N, K) * (N*K + N) floats of scratch */
extern "C" int hpfg_gemm_tn_bf16x3(const float* dy, const float* x, float* dw_db, float* partials, long R, int N, int K, int with_db, void* stream) {
  HPFG_ARG_CHECK(dy && x && dw_db && partials && R > 0 && N > 0 && K > 0, "gemm_tn_bf16x3: bad args");
  const int S = tn_splits(R, N, K);
  int per = (int)((R + S - 1) / S);
  per = (per + TR - 1) / TR * TR;
  dim3 grid((N + TT - 1) / TT, (K + TT - 1) / TT, S);
  const long stride = (long)N * K + (with_db ? N : 0);
  float* dst = S == 1 ? dw_db : partials;
  const bool aligned = (N & 3) == 0 && (K & 3) == 0 && (stride & 3) == 0 && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dst) & 15) == 0;
  if (aligned) hipLaunchKernelGGL(gemm_tn_bf16x3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, dy, (long)N, x, (long)K, dst, R, N, K, per, with_db, stride);
  else hipLaunchKernelGGL(gemm_tn_bf16x3_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, dy, (long)N, x, (long)K, dst, R, N, K, per, with_db, stride);
  if (S > 1) hipLaunchKernelGGL(gemm_rows_sum_kernel, dim3((unsigned)((stride + 63) / 64)), dim3(256), 0, (hipStream_t)stream, partials, S, stride, dw_db);
  return hpfg_launch_status("gemm_tn_bf16x3_kernel");
}

extern "C" int hpfg_col_sum_splits(long R) {
  const long s = (R + 511) / 512;
  return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

/* out[c] = sum_r x[r][c] in two deterministic stages; scratch: hpfg_col_sum_splits(R) * M floats */
extern "C" int hpfg_col_sum2(const float* x, long R, int M, long ldx, float* out, float* scratch, void* stream) {
  HPFG_ARG_CHECK(x && out && scratch && R > 0 && M > 0 && ldx >= M, "col_sum2: bad args");
  const int S = hpfg_col_sum_splits(R);
  const int per = (int)((R + S - 1) / S);
  hipLaunchKernelGGL(col_sum_part_kernel, dim3((M + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, x, R, M, ldx, per, S == 1 ? out : scratch);
  if (S > 1) hipLaunchKernelGGL(gemm_rows_sum_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, (hipStream_t)stream, scratch, S, (long)M, out);
  return hpfg_launch_status("col_sum_part_kernel");
}
