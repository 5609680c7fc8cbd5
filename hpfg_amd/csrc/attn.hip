// Attention core of the SegFormer branch on the matrix cores (reference model/segformer.py:92-128, Attention.forward after the q / kv
// projections and the spatial reduction):   out = softmax(scale * q k^T) v   per (image, head), head dim 32, at most 64 keys.
//
// Every product is split-bf16 ("bf16x3": hi*hi + hi*lo + lo*hi, fp32 accumulate) on v_mfma_f32_16x16x32_bf16, like the convolutions.
// Head dim 32 is exactly one MFMA k-step, so a 16 x 16 score tile costs three MFMAs.  Two facts keep the kernels free of LDS transposes
// of the probabilities:
//   * Q, K, V and dO fragments all have the same lane layout (row or column = lane & 15, 8 consecutive head-dim elements of k-group
//     lane >> 4), so the SAME registers give S^T = K Q^T (keys on the accumulator rows, the query on the lane) or S = Q K^T (queries on
//     the rows, the key on the lane) just by swapping the MFMA arguments.
//   * An accumulator tile feeds the next MFMA as its B operand directly (MI355X guide, "accumulator tile as the next MFMA's operand"):
//     the 8 contraction positions of a lane are its 4 accumulator rows of tile 2s and its 4 rows of tile 2s + 1; the OTHER operand (V^T,
//     K^T, dO^T, Q^T from LDS) is gathered in that same permuted order, which costs two 8-byte LDS reads instead of one 16-byte read.
// Forward and dQ use the S^T orientation (softmax statistics are then a 16-value reduction in registers plus two cross-lane steps);
// dK / dV use the S orientation and accumulate over all queries of a workgroup in registers; per-workgroup partials are summed in a fixed
// order by attn_dkv_sum_kernel (deterministic, no atomics).
//   q [B,N,heads,32], kv [B,M,2,heads,32] (the kv Linear's output layout), out / dq like q, dkv like kv.
#include "common.h"

namespace {

typedef __bf16 a_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 a_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 a_bf16x2 __attribute__((ext_vector_type(2)));
typedef float a_f32x2 __attribute__((ext_vector_type(2)));

constexpr int D = 32, MK = 64;                 // head dim, maximum keys
constexpr int KROW = 80;                       // bytes per row of a [64 keys][32 d] bf16 image (64 B + 16 B pad)
constexpr int TROW = 144;                      // bytes per row of a [32 d][64 keys] bf16 image (128 B + 16 B pad)
constexpr int KPLANE = MK * KROW, TPL = D * TROW;
constexpr float NEG = -3.0e38f;

__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hw, uint32_t& lw) {
  const a_bf16x2 h = __builtin_convertvector(a_f32x2{x0, x1}, a_bf16x2);
  hw = __builtin_bit_cast(uint32_t, h);
  const a_f32x2 hf = {__builtin_bit_cast(float, hw << 16), __builtin_bit_cast(float, hw & 0xFFFF0000u)};
  lw = __builtin_bit_cast(uint32_t, __builtin_convertvector(a_f32x2{x0, x1} - hf, a_bf16x2));
}
__device__ __forceinline__ void split8v(const float (&v)[8], a_bf16x8& hi, a_bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) split2(v[2 * k], v[2 * k + 1], h[k], l[k]);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(a_bf16x8, (u32x4{h[0], h[1], h[2], h[3]}));
  lo = __builtin_bit_cast(a_bf16x8, (u32x4{l[0], l[1], l[2], l[3]}));
}

#define ATT_MFMA3(ACC, AH, AL, BH, BL)                                \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, BH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL, BH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, BL, ACC, 0, 0, 0);

// rows [key][32 d] of k or v (which = 0 / 1) of one (image, head) -> natural image [64][32] (hi, lo) and / or transposed image [32][64]
__device__ __forceinline__ void stage_kv(const float* __restrict__ kv, int b, int h, int M, int C, int which, unsigned char* nat, unsigned char* trn, int tid) {
  const int key = tid >> 2, d0 = (tid & 3) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (key < M) {
    const float* p = kv + (((long)b * M + key) * 2 + which) * C + h * D + d0;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
  }
  a_bf16x8 hi, lo;
  split8v(v, hi, lo);
  if (nat) {
    *reinterpret_cast<a_bf16x8*>(nat + key * KROW + d0 * 2) = hi;
    *reinterpret_cast<a_bf16x8*>(nat + key * KROW + d0 * 2 + KPLANE) = lo;
  }
  if (trn) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      *reinterpret_cast<__bf16*>(trn + (d0 + j) * TROW + key * 2) = hi[j];
      *reinterpret_cast<__bf16*>(trn + (d0 + j) * TROW + key * 2 + TPL) = lo[j];
    }
  }
}

// fragment of a natural [rows][32 d] image: row = row0 + (lane & 15), 8 consecutive d of k-group lane >> 4
__device__ __forceinline__ a_bf16x8 nat_frag(const unsigned char* plane, int row0, int lane) {
  return *reinterpret_cast<const a_bf16x8*>(plane + (row0 + (lane & 15)) * KROW + (lane >> 4) * 16);
}
// fragment of a transposed [32 d][64 pos] image for contraction step s over 32 positions, in the accumulator-operand order: this lane's
// positions are 32 s + 4 g + {0..3} and 32 s + 16 + 4 g + {0..3} (g = lane >> 4), row = d0 + (lane & 15)
__device__ __forceinline__ a_bf16x8 trn_frag(const unsigned char* plane, int rowbytes, int d0, int s, int lane) {
  const unsigned char* p = plane + (d0 + (lane & 15)) * rowbytes + (32 * s + 4 * (lane >> 4)) * 2;
  const a_bf16x4 a = *reinterpret_cast<const a_bf16x4*>(p), b = *reinterpret_cast<const a_bf16x4*>(p + 32);
  return a_bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// this lane's 8 head-dim values (k-group lane >> 4) of row `row` of a [.., heads, 32] tensor, times `mul`; zeros beyond `nrows`
__device__ __forceinline__ void load_row8(const float* __restrict__ base, long row, long nrows, int C, int h, int lane, float mul, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (row < nrows) {
    const float* p = base + row * C + h * D + (lane >> 4) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0] * mul; v[1] = a[1] * mul; v[2] = a[2] * mul; v[3] = a[3] * mul;
    v[4] = c[0] * mul; v[5] = c[1] * mul; v[6] = c[2] * mul; v[7] = c[3] * mul;
  }
}

// S^T tiles (keys on rows) of 16 queries -> probabilities p[t][r] of key 16 t + 4 g + r for the query on this lane (lane & 15)
__device__ __forceinline__ void softmax_t(const unsigned char* ldsK, const a_bf16x8& qh, const a_bf16x8& ql, int M, int lane, f32x4 (&p)[4]) {
  const int g = lane >> 4;
  float mx = NEG;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    p[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const a_bf16x8 kh = nat_frag(ldsK, 16 * t, lane), kl = nat_frag(ldsK + KPLANE, 16 * t, lane);
    ATT_MFMA3(p[t], kh, kl, qh, ql)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (16 * t + 4 * g + r >= M) p[t][r] = NEG;
      mx = fmaxf(mx, p[t][r]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[t][r] = 16 * t + 4 * g + r < M ? expf(p[t][r] - mx) : 0.f;
      den += p[t][r];
    }
  den += __shfl_xor(den, 16);
  den += __shfl_xor(den, 32);
  const float inv = 1.f / den;
#pragma unroll
  for (int t = 0; t < 4; ++t) p[t] *= inv;
}

// accumulator tiles (2 s, 2 s + 1) -> split-bf16 B operand of contraction step s
__device__ __forceinline__ void acc_operand(const f32x4& a, const f32x4& b, a_bf16x8& hi, a_bf16x8& lo) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  split8v(v, hi, lo);
}

__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out, int N, int M,
                                                            int heads, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * KPLANE + 2 * TPL];      // K hi | K lo | V^T hi | V^T lo
  unsigned char* ldsK = lds;
  unsigned char* ldsVT = lds + 2 * KPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  stage_kv(kv, b, h, M, C, 0, ldsK, nullptr, tid);
  stage_kv(kv, b, h, M, C, 1, nullptr, ldsVT, tid);
  __syncthreads();
  const long qi = (long)blockIdx.x * 64 + wave * 16 + (lane & 15);
  float qv[8];
  load_row8(q + (long)b * N * C, qi, N, C, h, lane, scale, qv);
  a_bf16x8 qh, ql;
  split8v(qv, qh, ql);
  f32x4 p[4];
  softmax_t(ldsK, qh, ql, M, lane, p);
  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_bf16x8 ph, pl;
    acc_operand(p[2 * s], p[2 * s + 1], ph, pl);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const a_bf16x8 vh = trn_frag(ldsVT, TROW, 16 * dt, s, lane), vl = trn_frag(ldsVT + TPL, TROW, 16 * dt, s, lane);
      ATT_MFMA3(o[dt], vh, vl, ph, pl)          // O^T[d][q]
    }
  }
  if (qi < N) {
    float* op = out + ((long)b * N + qi) * C + h * D + (lane >> 4) * 4;
    *reinterpret_cast<f32x4*>(op) = o[0];
    *reinterpret_cast<f32x4*>(op + 16) = o[1];
  }
}

// dq = scale * dS K with dS = P .* (dP - rowsum(P .* dP)), dP = dO V^T   (S^T orientation, one query per lane)
__global__ __launch_bounds__(256) void attn_mfma_dq_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ dout,
                                                           float* __restrict__ dq, int N, int M, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * KPLANE + 2 * TPL];      // K hi|lo, V hi|lo (natural), K^T hi|lo
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + 2 * KPLANE;
  unsigned char* ldsKT = lds + 4 * KPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  stage_kv(kv, b, h, M, C, 0, ldsK, ldsKT, tid);
  stage_kv(kv, b, h, M, C, 1, ldsV, nullptr, tid);
  __syncthreads();
  const long qi = (long)blockIdx.x * 64 + wave * 16 + (lane & 15);
  float qv[8], dv[8];
  load_row8(q + (long)b * N * C, qi, N, C, h, lane, scale, qv);
  load_row8(dout + (long)b * N * C, qi, N, C, h, lane, 1.f, dv);
  a_bf16x8 qh, ql, dh, dl;
  split8v(qv, qh, ql);
  split8v(dv, dh, dl);
  f32x4 p[4], dp[4];
  softmax_t(ldsK, qh, ql, M, lane, p);
  float delta = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const a_bf16x8 vh = nat_frag(ldsV, 16 * t, lane), vl = nat_frag(ldsV + KPLANE, 16 * t, lane);
    ATT_MFMA3(dp[t], vh, vl, dh, dl)            // dP^T[key][q]
#pragma unroll
    for (int r = 0; r < 4; ++r) delta += p[t][r] * dp[t][r];
  }
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dp[t][r] = p[t][r] * (dp[t][r] - delta);          // dS^T
  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_bf16x8 sh, sl;
    acc_operand(dp[2 * s], dp[2 * s + 1], sh, sl);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const a_bf16x8 kh = trn_frag(ldsKT, TROW, 16 * dt, s, lane), kl = trn_frag(ldsKT + TPL, TROW, 16 * dt, s, lane);
      ATT_MFMA3(o[dt], kh, kl, sh, sl)          // dQ^T[d][q] / scale
    }
  }
  if (qi < N) {
    float* op = dq + ((long)b * N + qi) * C + h * D + g * 4;
    *reinterpret_cast<f32x4*>(op) = o[0] * scale;
    *reinterpret_cast<f32x4*>(op + 16) = o[1] * scale;
  }
}

// dV = P^T dO, dK = scale * dS^T Q over the queries [q0, q1) of this workgroup (S orientation: 4 queries per lane, the key on the lane).
// Each wave takes 32 queries per step; the transposed images dO^T / Q^T [32 d][32 q] of those queries live in a per-wave LDS slot.
constexpr int QROW = 80;                        // bytes per row of a per-wave [32 d][32 q] bf16 image (64 B + 16 B pad)
constexpr int QPL = D * QROW;

__global__ __launch_bounds__(256) void attn_mfma_dkv_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ dout,
                                                            float* __restrict__ part, int N, int M, int heads, float scale, int q_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * KPLANE + 4 * 4 * QPL];      // K, V natural (hi|lo each); per wave: dO^T hi|lo, Q^T hi|lo
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + 2 * KPLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  unsigned char* myDO = lds + 4 * KPLANE + wave * 4 * QPL;
  unsigned char* myQ = myDO + 2 * QPL;
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  stage_kv(kv, b, h, M, C, 0, ldsK, nullptr, tid);
  stage_kv(kv, b, h, M, C, 1, ldsV, nullptr, tid);
  __syncthreads();
  f32x4 accV[2][4], accK[2][4];                 // dV^T / dK^T [d tile][key tile]: rows d, column = key on the lane
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      accV[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      accK[dt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // K / V fragments with the key on the lane (column operand): constant over the loop
  a_bf16x8 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kh[t] = nat_frag(ldsK, 16 * t, lane);
    kl[t] = nat_frag(ldsK + KPLANE, 16 * t, lane);
    vh[t] = nat_frag(ldsV, 16 * t, lane);
    vl[t] = nat_frag(ldsV + KPLANE, 16 * t, lane);
  }
  const long q0 = (long)blockIdx.x * q_per_block, q1 = q0 + q_per_block < N ? q0 + q_per_block : N;
  const float* qb = q + (long)b * N * C;
  const float* db = dout + (long)b * N * C;
  for (long base = q0 + wave * 32; base < q1; base += 128) {
    f32x4 s[2][4], dp[2][4];
    a_bf16x8 qh[2], ql[2], dh[2], dl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long qi = base + 16 * u + (lane & 15);
      float qv[8], dv[8];
      load_row8(qb, qi, q1, C, h, lane, scale, qv);
      load_row8(db, qi, q1, C, h, lane, 1.f, dv);
      split8v(qv, qh[u], ql[u]);
      split8v(dv, dh[u], dl[u]);
      // transposed per-wave images for the dV / dK products: element (d = 8 g + j, query position 16 u + (lane & 15))
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int off = (8 * g + j) * QROW + (16 * u + (lane & 15)) * 2;
        *reinterpret_cast<__bf16*>(myDO + off) = dh[u][j];
        *reinterpret_cast<__bf16*>(myDO + off + QPL) = dl[u][j];
        *reinterpret_cast<__bf16*>(myQ + off) = qh[u][j];
        *reinterpret_cast<__bf16*>(myQ + off + QPL) = ql[u][j];
      }
    }
    // S[q][key] and dP[q][key]: queries 16 u + 4 g + r on the rows, key 16 t + (lane & 15) on the lane
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        ATT_MFMA3(s[u][t], qh[u], ql[u], kh[t], kl[t])
        ATT_MFMA3(dp[u][t], dh[u], dl[u], vh[t], vl[t])
      }
    // softmax statistics per query row: 4 tiles x 16 lanes hold a row's 64 scores
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float mx = NEG;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (16 * t + (lane & 15) >= M) s[u][t][r] = NEG;
          mx = fmaxf(mx, s[u][t][r]);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float den = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[u][t][r] = 16 * t + (lane & 15) < M ? expf(s[u][t][r] - mx) : 0.f;
          den += s[u][t][r];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) den += __shfl_xor(den, o);
        const float inv = 1.f / den;
        float delta = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[u][t][r] *= inv;                                 // P
          delta += s[u][t][r] * dp[u][t][r];
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) delta += __shfl_xor(delta, o);
        const bool live = base + 16 * u + 4 * g + r < q1;      // rows beyond the range contribute nothing (their q / dO were zeroed; P is not zero)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          dp[u][t][r] = live ? s[u][t][r] * (dp[u][t][r] - delta) : 0.f;      // dS
          if (!live) s[u][t][r] = 0.f;
        }
      }
    // dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]   (one contraction step over the wave's 32 queries)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a_bf16x8 ph, pl, sh, sl;
      acc_operand(s[0][t], s[1][t], ph, pl);
      acc_operand(dp[0][t], dp[1][t], sh, sl);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const a_bf16x8 oh = trn_frag(myDO, QROW, 16 * dt, 0, lane), ol = trn_frag(myDO + QPL, QROW, 16 * dt, 0, lane);
        const a_bf16x8 th = trn_frag(myQ, QROW, 16 * dt, 0, lane), tl = trn_frag(myQ + QPL, QROW, 16 * dt, 0, lane);
        ATT_MFMA3(accV[dt][t], oh, ol, ph, pl)
        ATT_MFMA3(accK[dt][t], th, tl, sh, sl)
      }
    }
  }
  // reduce the four waves in a fixed order (wave 0 stores, waves 1..3 add in turn) and write this workgroup's partial [2][64 keys][32 d];
  // accumulator rows = d (4 g + r), column = key.  The K / V images are dead by now (their fragments live in registers).
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);                       // [2][64][32] floats = 16 KB
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + (lane & 15), d = 16 * dt + 4 * g + r;
            float* pk = red + (0 * MK + key) * D + d;
            float* pv = red + (1 * MK + key) * D + d;
            *pk = w == 0 ? accK[dt][t][r] : *pk + accK[dt][t][r];
            *pv = w == 0 ? accV[dt][t][r] : *pv + accV[dt][t][r];
          }
    }
    __syncthreads();
  }
  float* o = part + (((long)(b * heads + h) * gridDim.x + blockIdx.x) * 2) * MK * D;
  for (int e = tid; e < 2 * MK * D; e += 256) o[e] = red[e];
}

// dkv[b][key][which][h][d] = sum over the query blocks of part[b][h][blk][which][key][d]  (dK already carries the scale through the scaled q)
__global__ __launch_bounds__(256) void attn_dkv_sum_kernel(const float* __restrict__ part, float* __restrict__ dkv, int nblk, int M, int heads) {
  const int b = blockIdx.z, h = blockIdx.y, C = heads * D;
  for (int e = threadIdx.x; e < 2 * M * D; e += 256) {
    const int which = e / (M * D), key = (e / D) % M, d = e % D;
    const float* p = part + (((long)(b * heads + h) * nblk) * 2 + which) * MK * D + key * D + d;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += p[(long)k * 2 * MK * D];
    dkv[(((long)b * M + key) * 2 + which) * C + h * D + d] = s;
  }
}

}  // namespace

extern "C" int hpfg_attn_mfma_fwd(const float* q, const float* kv, float* out, int B, int N, int M, int heads, float scale, void* stream) {
  HPFG_ARG_CHECK(q && kv && out && B > 0 && N > 0 && M > 0 && M <= MK && heads > 0, "attn_mfma_fwd: bad args (at most %d keys, head dim %d)", MK, D);
  hipLaunchKernelGGL(attn_mfma_fwd_kernel, dim3((N + 63) / 64, heads, B), dim3(256), 0, (hipStream_t)stream, q, kv, out, N, M, heads, scale);
  return hpfg_launch_status("attn_mfma_fwd_kernel");
}

extern "C" int hpfg_attn_mfma_blocks(int N) {
  int per = 512;                                 // queries per dK / dV workgroup (a multiple of 128: four waves x 32)
  return (N + per - 1) / per;
}

/* dq [B,N,heads,32], dkv [B,M,2,heads,32]; scratch: B * heads * hpfg_attn_mfma_blocks(N) * 2 * 64 * 32 floats */
extern "C" int hpfg_attn_mfma_bwd(const float* q, const float* kv, const float* dout, float* dq, float* dkv, float* scratch, int B, int N, int M, int heads,
                                  float scale, void* stream) {
  HPFG_ARG_CHECK(q && kv && dout && dq && dkv && scratch && B > 0 && N > 0 && M > 0 && M <= MK && heads > 0, "attn_mfma_bwd: bad args");
  hipLaunchKernelGGL(attn_mfma_dq_kernel, dim3((N + 63) / 64, heads, B), dim3(256), 0, (hipStream_t)stream, q, kv, dout, dq, N, M, heads, scale);
  const int nblk = hpfg_attn_mfma_blocks(N);
  hipLaunchKernelGGL(attn_mfma_dkv_kernel, dim3(nblk, heads, B), dim3(256), 0, (hipStream_t)stream, q, kv, dout, scratch, N, M, heads, scale, 512);
  hipLaunchKernelGGL(attn_dkv_sum_kernel, dim3(1, heads, B), dim3(256), 0, (hipStream_t)stream, scratch, dkv, nblk, M, heads);
  return hpfg_launch_status("attn_mfma_bwd");
}
