// Token-layout ([B, N, C] == NHWC) kernels of the SegFormer branch (reference model/segformer.py):
//   LayerNorm forward / backward                      nn.LayerNorm(dim), eps 1e-5         (segformer.py:107,174,192,195,232-244)
//   attention core softmax(q k^T * scale) v, fwd/bwd  Attention.forward :122-126; at most 64 keys (49 at 224x224), head dim 32
//   depthwise 3x3 conv + exact GELU, fwd/bwd          DWConv :139-146 + F.gelu in MLP.forward :156
// First version: plain fp32, one pass per op, sized for correctness and coalescing (16-byte accesses along C); the GEMM-shaped parts
// of the branch (q / kv / proj / fc1 / fc2 / head projections) are library GEMMs on the host side.
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-5f;

// ---- LayerNorm: one wave per row, C <= 1024, C % 4 == 0 ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * C;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (k * 64 + lane) * 4;
    v[k] = c < C ? *reinterpret_cast<const f32x4*>(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < C)
#pragma unroll
      for (int j = 0; j < 4; ++j) q += (v[k][j] - mu) * (v[k][j] - mu);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rs = rsqrtf(q / (float)C + LN_EPS);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < C) {
      const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(b + c);
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[k][j] - mu) * rs * gg[j] + bb[j];
      *reinterpret_cast<f32x4*>(y + row * C + c) = o;
    }
  }
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// dx per row; per-workgroup partial sums of dgamma / dbeta in part[blockIdx][2][C] (summed by ln_param_reduce_kernel in a fixed order)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ g,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                     float* __restrict__ part, long rows, int C, int rows_per_wg) {
  __shared__ float red[4][2][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 ag[4], ab[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ag[k] = ab[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long r0 = (long)blockIdx.x * rows_per_wg;
  for (long row = r0 + wave; row < r0 + rows_per_wg && row < rows; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 xh[4], dg[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (k * 64 + lane) * 4;
      xh[k] = dg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + c), dv = *reinterpret_cast<const f32x4*>(dy + row * C + c);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[k][j] = (xv[j] - mu) * rs;
          dg[k][j] = dv[j] * gg[j];
          s1 += dg[k][j];
          s2 += dg[k][j] * xh[k][j];
          ag[k][j] += dv[j] * xh[k][j];
          ab[k][j] += dv[j];
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < C) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (dg[k][j] - m1 - xh[k][j] * m2);
        *reinterpret_cast<f32x4*>(dx + row * C + c) = o;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = (k * 64 + lane) * 4;
    if (c < C)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[wave][0][c + j] = ag[k][j];
        red[wave][1][c + j] = ab[k][j];
      }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, c = o % C;
    part[((long)blockIdx.x * 2 + which) * C + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// C <= 256: G = 8 / 16 / 32 / 64 lanes per row (one float4 each), 64 / G rows per wave -- with C = 32 the one-row-per-wave kernels above
// keep 8 of 64 lanes busy
template <int G>
__global__ __launch_bounds__(256) void ln_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                           float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long rows, int C) {
  constexpr int RW = 64 / G;
  const int lane = threadIdx.x & 63, q = lane % G, c = q * 4;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + lane / G;
  const bool on = row < rows && c < C;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (on) v = *reinterpret_cast<const f32x4*>(x + row * C + c);
  float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float d2 = 0.f;
  if (on)
#pragma unroll
    for (int j = 0; j < 4; ++j) d2 += (v[j] - mu) * (v[j] - mu);
#pragma unroll
  for (int o = G / 2; o >= 1; o >>= 1) d2 += __shfl_xor(d2, o);
  const float rs = rsqrtf(d2 / (float)C + LN_EPS);
  if (on) {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), bb = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = (v[j] - mu) * rs * gg[j] + bb[j];
    *reinterpret_cast<f32x4*>(y + row * C + c) = o4;
    if (q == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

template <int G>
__global__ __launch_bounds__(256) void ln_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ g,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                           float* __restrict__ part, long rows, int C, int rows_per_wg) {
  constexpr int RW = 64 / G;
  __shared__ float red[4][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane % G, c = q * 4;
  const bool chan = c < C;
  f32x4 gg = {0.f, 0.f, 0.f, 0.f};
  if (chan) gg = *reinterpret_cast<const f32x4*>(g + c);
  f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = ag;
  const long r0 = (long)blockIdx.x * rows_per_wg;
  for (long rb = r0 + wave * RW; rb < r0 + rows_per_wg && rb < rows; rb += 4 * RW) {
    const long row = rb + lane / G;
    const bool on = chan && row < rows && row < r0 + rows_per_wg;
    f32x4 xh = {0.f, 0.f, 0.f, 0.f}, dg = xh;
    float s1 = 0.f, s2 = 0.f, rs = 0.f;
    if (on) {
      const float mu = mean[row];
      rs = rstd[row];
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + c), dv = *reinterpret_cast<const f32x4*>(dy + row * C + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xh[j] = (xv[j] - mu) * rs;
        dg[j] = dv[j] * gg[j];
        s1 += dg[j];
        s2 += dg[j] * xh[j];
        ag[j] += dv[j] * xh[j];
        ab[j] += dv[j];
      }
    }
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    if (on) {
      const float m1 = s1 / (float)C, m2 = s2 / (float)C;
      f32x4 o4;
#pragma unroll
      for (int j = 0; j < 4; ++j) o4[j] = rs * (dg[j] - m1 - xh[j] * m2);
      *reinterpret_cast<f32x4*>(dx + row * C + c) = o4;
    }
  }
  // lanes with the same channel quad (q, q + G, ...) combine, then the four waves
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) {
      ag[j] += __shfl_xor(ag[j], o);
      ab[j] += __shfl_xor(ab[j], o);
    }
  if (lane < G && chan)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      red[wave][0][c + j] = ag[j];
      red[wave][1][c + j] = ab[j];
    }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, cc = o % C;
    part[((long)blockIdx.x * 2 + which) * C + cc] = (red[0][which][cc] + red[1][which][cc]) + (red[2][which][cc] + red[3][which][cc]);
  }
}

__global__ __launch_bounds__(64) void col_reduce_kernel(const float* __restrict__ part, int nblk, int stride, float* __restrict__ out) {
  const int c = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) a += (double)part[(long)i * stride + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (threadIdx.x == 0) out[c] = (float)a;
}

// out[o] = sum_s part[s * n + o] for MANY outputs o (weight-gradient partial matrices): lane <-> output (coalesced), the four waves of a
// workgroup take every fourth row with four loads in flight each, fixed combination order
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int S, long n, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long o = (long)blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (o < n) {
    int s = wave;
    for (; s + 12 < S; s += 16) {
      a0 += part[(long)s * n + o];
      a1 += part[(long)(s + 4) * n + o];
      a2 += part[(long)(s + 8) * n + o];
      a3 += part[(long)(s + 12) * n + o];
    }
    for (; s < S; s += 4) a0 += part[(long)s * n + o];
  }
  red[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && o < n) out[o] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- attention core: thread per query, K / V of one (batch, head) in LDS; M <= 64 keys, head dim 32 ---------------------------------
constexpr int AD = 32, AM = 64;
// q [B,N,h,32] (row stride C = h*32), kv [B,M,2,h,32] (the kv Linear's output), out [B,N,h,32]
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, float* __restrict__ out, int N, int M,
                                                       int heads, float scale) {
  __shared__ float ks[AM][AD], vs[AM][AD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * AD;
  for (int e = threadIdx.x; e < M * AD; e += 256) {
    const int j = e / AD, c = e % AD;
    ks[j][c] = kv[(((long)b * M + j) * 2 + 0) * C + h * AD + c];
    vs[j][c] = kv[(((long)b * M + j) * 2 + 1) * C + h * AD + c];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float qv[AD], s[AM];
  const float* qp = q + ((long)b * N + i) * C + h * AD;
#pragma unroll
  for (int c = 0; c < AD; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qp + c);
    qv[c] = t[0]; qv[c + 1] = t[1]; qv[c + 2] = t[2]; qv[c + 3] = t[3];
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < AM; ++j) {
    float d = 0.f;
    if (j < M)
#pragma unroll
      for (int c = 0; c < AD; ++c) d += qv[c] * ks[j][c];
    s[j] = j < M ? d * scale : -3.0e38f;
    mx = fmaxf(mx, s[j]);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < AM; ++j) {
    s[j] = j < M ? expf(s[j] - mx) : 0.f;
    den += s[j];
  }
  const float inv = 1.f / den;
  float o[AD];
#pragma unroll
  for (int c = 0; c < AD; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < AM; ++j)
    if (j < M) {
      const float p = s[j] * inv;
#pragma unroll
      for (int c = 0; c < AD; ++c) o[c] += p * vs[j][c];
    }
  float* op = out + ((long)b * N + i) * C + h * AD;
#pragma unroll
  for (int c = 0; c < AD; c += 4) *reinterpret_cast<f32x4*>(op + c) = f32x4{o[c], o[c + 1], o[c + 2], o[c + 3]};
}

// backward per query: recompute P, then dq (in place of nothing else to reduce) and the two [B,h,N,M] matrices P and dS that the host
// turns into dV = P^T dO and dK = scale * dS^T Q with library GEMMs.
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ dout,
                                                       float* __restrict__ dq, float* __restrict__ P, float* __restrict__ dS, int N, int M, int heads,
                                                       float scale) {
  __shared__ float ks[AM][AD], vs[AM][AD];
  const int b = blockIdx.z, h = blockIdx.y, C = heads * AD;
  for (int e = threadIdx.x; e < M * AD; e += 256) {
    const int j = e / AD, c = e % AD;
    ks[j][c] = kv[(((long)b * M + j) * 2 + 0) * C + h * AD + c];
    vs[j][c] = kv[(((long)b * M + j) * 2 + 1) * C + h * AD + c];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float qv[AD], dov[AD], s[AM];
  const float* qp = q + ((long)b * N + i) * C + h * AD;
  const float* dp = dout + ((long)b * N + i) * C + h * AD;
#pragma unroll
  for (int c = 0; c < AD; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(qp + c), u = *reinterpret_cast<const f32x4*>(dp + c);
    qv[c] = t[0]; qv[c + 1] = t[1]; qv[c + 2] = t[2]; qv[c + 3] = t[3];
    dov[c] = u[0]; dov[c + 1] = u[1]; dov[c + 2] = u[2]; dov[c + 3] = u[3];
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < AM; ++j) {
    float d = 0.f;
    if (j < M)
#pragma unroll
      for (int c = 0; c < AD; ++c) d += qv[c] * ks[j][c];
    s[j] = j < M ? d * scale : -3.0e38f;
    mx = fmaxf(mx, s[j]);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < AM; ++j) {
    s[j] = j < M ? expf(s[j] - mx) : 0.f;
    den += s[j];
  }
  const float inv = 1.f / den;
  float dsum = 0.f, dpj[AM];
#pragma unroll
  for (int j = 0; j < AM; ++j) {
    s[j] *= inv;
    float d = 0.f;
    if (j < M)
#pragma unroll
      for (int c = 0; c < AD; ++c) d += dov[c] * vs[j][c];
    dpj[j] = d;
    dsum += s[j] * d;
  }
  float dqv[AD];
#pragma unroll
  for (int c = 0; c < AD; ++c) dqv[c] = 0.f;
  float* Pr = P + (((long)b * heads + h) * N + i) * M;
  float* Sr = dS + (((long)b * heads + h) * N + i) * M;
#pragma unroll
  for (int j = 0; j < AM; ++j)
    if (j < M) {
      const float ds = s[j] * (dpj[j] - dsum);
      Pr[j] = s[j];
      Sr[j] = ds;
#pragma unroll
      for (int c = 0; c < AD; ++c) dqv[c] += ds * ks[j][c];
    }
  float* qo = dq + ((long)b * N + i) * C + h * AD;
#pragma unroll
  for (int c = 0; c < AD; c += 4) *reinterpret_cast<f32x4*>(qo + c) = f32x4{dqv[c] * scale, dqv[c + 1] * scale, dqv[c + 2] * scale, dqv[c + 3] * scale};
}

// ---- depthwise 3x3 (pad 1) + exact GELU on [B,H,W,C], C % 4 == 0; weights transposed to [9][C] -------------------------------------
__device__ inline float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ inline float gelu_grad(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
}

__device__ inline f32x4 dw_at(const float* __restrict__ x, const float* __restrict__ w9, const float* __restrict__ bias, int b, int y, int xx, int c,
                              int H, int W, int C) {
  f32x4 a = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, x2 = xx + t % 3 - 1;
    if (yy >= 0 && yy < H && x2 >= 0 && x2 < W) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)b * H + yy) * W + x2) * C + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
      a += v * ww;
    }
  }
  return a;
}

__global__ __launch_bounds__(256) void dwgelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w9, const float* __restrict__ bias,
                                                         float* __restrict__ y, int B, int H, int W, int C) {
  const int Q = C / 4;
  const long total = (long)B * H * W * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long p = i / Q;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    const f32x4 u = dw_at(x, w9, bias, b, yy, xx, c, H, W, C);
    *reinterpret_cast<f32x4*>(y + p * C + c) = f32x4{gelu_f(u[0]), gelu_f(u[1]), gelu_f(u[2]), gelu_f(u[3])};
  }
}

// du = dy * gelu'(u) (u recomputed), plus per-workgroup partial sums of dbias and dweight[9] in part[blockIdx][10][C]
constexpr int DWG_QW = 64;
__global__ __launch_bounds__(256) void dwgelu_bwd_du_kernel(const float* __restrict__ x, const float* __restrict__ w9, const float* __restrict__ bias,
                                                            const float* __restrict__ dy, float* __restrict__ du, float* __restrict__ part, int B, int H,
                                                            int W, int C) {
  __shared__ float red[256 * 40];
  const int Q = C / 4;                       // 256 % Q == 0 or Q % 256 == 0 is NOT required: a thread's channel quad changes per item,
  const long total = (long)B * H * W * Q;    // so the partial sums are accumulated per (item % Q) through LDS below
  // each workgroup owns a contiguous range of pixels and loops over channel quads inside: thread t <-> quad (t % Qw), pixel lane t / Qw
  // and blockIdx.y picks a slice of DWG_QW channel quads, so that at least 4 pixels are in flight per workgroup whatever C is (with all of a
  // wide stage's quads in one workgroup a thread walked its 32 pixels one after the other: 70 - 160 us of pure load latency per launch)
  const int Qw = Q < DWG_QW ? Q : DWG_QW;    // quads handled concurrently
  const int PL = 256 / Qw;                   // pixel lanes
  const long npix = (long)B * H * W;
  const long per = (npix + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
  const int ql = threadIdx.x % Qw, pl = threadIdx.x / Qw;
  (void)total;
  {
    const int qb = blockIdx.y * Qw;
    const int c = (qb + ql) * 4;
    f32x4 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pl < PL && qb + ql < Q) {
      // the thread's channel quad is fixed: its 9 tap weights and bias stay in registers, and the 9 input taps of a pixel are loaded ONCE
      // for both the recomputed pre-activation and the weight-gradient sums (29 -> 10 loads per pixel)
      f32x4 wr[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[t] = *reinterpret_cast<const f32x4*>(w9 + t * C + c);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c);
      for (long p = p0 + pl; p < p1; p += PL) {
        const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((long)W * H));
        f32x4 xs[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
          xs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) xs[t] = *reinterpret_cast<const f32x4*>(x + (((long)b * H + y2) * W + x2) * C + c);
        }
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + p * C + c);
        f32x4 u = b4;
#pragma unroll
        for (int t = 0; t < 9; ++t) u += xs[t] * wr[t];
        const f32x4 g = f32x4{d[0] * gelu_grad(u[0]), d[1] * gelu_grad(u[1]), d[2] * gelu_grad(u[2]), d[3] * gelu_grad(u[3])};
        *reinterpret_cast<f32x4*>(du + p * C + c) = g;
        acc[9] += g;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] += g * xs[t];
      }
    }
    // all ten sums of every channel quad in one LDS round: red[thread][t][j]
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(threadIdx.x * 10 + t) * 4 + j] = acc[t][j];
    __syncthreads();
    for (int o = threadIdx.x; o < 10 * Qw * 4; o += 256) {
      const int t = o / (Qw * 4), cc = o % (Qw * 4), qq = cc >> 2, j = cc & 3;
      if (qb * 4 + cc < C) {
        float s = 0.f;
        for (int l = 0; l < PL; ++l) s += red[((l * Qw + qq) * 10 + t) * 4 + j];
        part[((long)blockIdx.x * 10 + t) * C + qb * 4 + cc] = s;
      }
    }
  }
}

// dx = transpose of the depthwise conv applied to du
__global__ __launch_bounds__(256) void dw_bwd_dx_kernel(const float* __restrict__ du, const float* __restrict__ w9, float* __restrict__ dx, int B, int H,
                                                        int W, int C) {
  const int Q = C / 4;
  const long total = (long)B * H * W * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long p = i / Q;
    const int xx = (int)(p % W), yy = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int y2 = yy - (t / 3 - 1), x2 = xx - (t % 3 - 1);      // output pixel that read this input through tap t
      if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W)
        a += *reinterpret_cast<const f32x4*>(du + (((long)b * H + y2) * W + x2) * C + c) * *reinterpret_cast<const f32x4*>(w9 + t * C + c);
    }
    *reinterpret_cast<f32x4*>(dx + p * C + c) = a;
  }
}

// ---- bilinear resize, align_corners=False (F.interpolate in SegFormerHead.forward :314,319), NHWC, C % 4 == 0 ------------------------
// source coordinate of output o: max(0, (o + 0.5) * in/out - 0.5); taps i0 = floor, i1 = min(i0 + 1, in - 1), weight of i1 = frac
__device__ inline void rs_src(int o, float scale, int n_in, int& i0, int& i1, float& w1) {
  float s = ((float)o + 0.5f) * scale - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 > n_in - 1 ? n_in - 1 : i0;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h, int w, int H, int W, int C,
                                                         float sy, float sx) {
  const int Q = C / 4;
  const long total = (long)B * H * W * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long p = i / Q;
    const int X = (int)(p % W), Y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    int y0, y1, x0, x1;
    float wy, wx;
    rs_src(Y, sy, h, y0, y1, wy);
    rs_src(X, sx, w, x0, x1, wx);
    const float* base = x + (long)b * h * w * C + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + ((long)y0 * w + x0) * C), bb = *reinterpret_cast<const f32x4*>(base + ((long)y0 * w + x1) * C);
    const f32x4 cc = *reinterpret_cast<const f32x4*>(base + ((long)y1 * w + x0) * C), d = *reinterpret_cast<const f32x4*>(base + ((long)y1 * w + x1) * C);
    const f32x4 top = a + (bb - a) * wx, bot = cc + (d - cc) * wx;
    *reinterpret_cast<f32x4*>(y + p * C + c) = top + (bot - top) * wy;
  }
}

// y = base + sum_k resize(x_k): the SegFormer head's fused feature map as one pass (model/segformer.py head: the stage outputs, already through
// their slice of linear_fuse, are brought to the 1/4-resolution grid and added) -- one read of base, one write of y, the low-res maps from L2.
struct ResizeSrc {
  const float* x;
  int h, w;
  float sy, sx;
};
__global__ __launch_bounds__(256) void resize_sum_fwd_kernel(const float* __restrict__ base, ResizeSrc s0, ResizeSrc s1, ResizeSrc s2, int nsrc,
                                                             float* __restrict__ y, int B, int H, int W, int C) {
  const int Q = C / 4;
  const long total = (long)B * H * W * Q;
  const ResizeSrc src[3] = {s0, s1, s2};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long p = i / Q;
    const int X = (int)(p % W), Y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    f32x4 acc = *reinterpret_cast<const f32x4*>(base + p * C + c);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (k >= nsrc) break;
      const int h = src[k].h, w = src[k].w;
      int y0, y1, x0, x1;
      float wy, wx;
      rs_src(Y, src[k].sy, h, y0, y1, wy);
      rs_src(X, src[k].sx, w, x0, x1, wx);
      const float* bp = src[k].x + (long)b * h * w * C + c;
      const f32x4 a = *reinterpret_cast<const f32x4*>(bp + ((long)y0 * w + x0) * C), bb = *reinterpret_cast<const f32x4*>(bp + ((long)y0 * w + x1) * C);
      const f32x4 cc = *reinterpret_cast<const f32x4*>(bp + ((long)y1 * w + x0) * C), d = *reinterpret_cast<const f32x4*>(bp + ((long)y1 * w + x1) * C);
      const f32x4 top = a + (bb - a) * wx, bot = cc + (d - cc) * wx;
      acc += top + (bot - top) * wy;
    }
    *reinterpret_cast<f32x4*>(y + p * C + c) = acc;
  }
}

// backward as a gather (no atomics: reproducible): a source pixel sums, over the outputs whose two taps per axis include it, the
// matching weights.
__global__ __launch_bounds__(256) void resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int h, int w, int H, int W, int C,
                                                         float sy, float sx) {
  const int Q = C / 4;
  const long total = (long)B * h * w * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long p = i / Q;
    const int xi = (int)(p % w), yi = (int)((p / w) % h), b = (int)(p / ((long)w * h));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // outputs whose lower tap is i have source coordinate in [i, i + 1), those whose upper tap is i in [i - 1, i): together the outputs
    // [(i - 0.5) / s - 0.5, (i + 1.5) / s - 0.5) -- 2 * out/in candidates per axis (one more on either side for rounding; the tests below
    // decide), not the 3 * out/in of the whole-pixel bounds
    const float qy = 1.f / sy, qx = 1.f / sx;
    const int Y0 = max(0, (int)floorf((yi - 0.5f) * qy - 0.5f) - 1), Y1 = min(H, (int)ceilf((yi + 1.5f) * qy - 0.5f) + 1);
    const int X0 = max(0, (int)floorf((xi - 0.5f) * qx - 0.5f) - 1), X1 = min(W, (int)ceilf((xi + 1.5f) * qx - 0.5f) + 1);
    for (int Y = Y0; Y < Y1; ++Y) {
      int y0, y1;
      float wy;
      rs_src(Y, sy, h, y0, y1, wy);
      const float fy = (y0 == yi ? 1.f - wy : 0.f) + (y1 == yi ? wy : 0.f);
      if (fy == 0.f) continue;
      for (int X = X0; X < X1; ++X) {
        int x0, x1;
        float wx;
        rs_src(X, sx, w, x0, x1, wx);
        const float fx = (x0 == xi ? 1.f - wx : 0.f) + (x1 == xi ? wx : 0.f);
        if (fx != 0.f) acc += *reinterpret_cast<const f32x4*>(dy + (((long)b * H + Y) * W + X) * C + c) * (fy * fx);
      }
    }
    *reinterpret_cast<f32x4*>(dx + p * C + c) = acc;
  }
}

// ---- BatchNorm (train) + ReLU + channel dropout over tokens [R, C] (ConvModule + Dropout2d of the head, segformer.py:288-296,307,318) ----
// thread t <-> channel quad t % Q and row lane t / Q; part[blockIdx][2][C] partial sums, reduced in a fixed order by col_reduce_kernel
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, long R, int C, float* __restrict__ part, int rows_per_wg) {
  __shared__ float red[256 * 8];
  const int Q = C >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, PL = 256 / Q, c = q * 4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  const long r0 = (long)blockIdx.x * rows_per_wg;
  if (pl < PL)
    for (long r = r0 + pl; r < r0 + rows_per_wg && r < R; r += PL) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
      a += v;
      b += v * v;
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = a[j];
    red[threadIdx.x * 8 + 4 + j] = b[j];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, cc = o % C, qq = cc >> 2, j = cc & 3;
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[(l * Q + qq) * 8 + which * 4 + j];
    part[((long)blockIdx.x * 2 + which) * C + cc] = acc;
  }
}

// y = relu((x - mean) * rstd * gamma + beta) * mask[row / rows_per_image][c] * inv_keep
__global__ __launch_bounds__(256) void bnrelu_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mask, float inv_keep, long rows_per_image, float* __restrict__ y,
                                                           long R, int C) {
  const int Q = C >> 2;
  const long total = R * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long r = i / Q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 m = {inv_keep, inv_keep, inv_keep, inv_keep};
    if (mask) m = *reinterpret_cast<const f32x4*>(mask + (r / rows_per_image) * C + c) * inv_keep;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf((v[j] - mu[j]) * rs[j] * g[j] + be[j], 0.f) * (mask ? m[j] : 1.f);
    *reinterpret_cast<f32x4*>(y + r * C + c) = o;
  }
}

// backward sums: g = dy * mask * inv_keep * [bn(x) > 0];  part[blk][0] = sum g, part[blk][1] = sum g * xhat
__global__ __launch_bounds__(256) void bnrelu_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ mask, float inv_keep,
                                                               long rows_per_image, long R, int C, float* __restrict__ part, int rows_per_wg) {
  __shared__ float red[256 * 8];
  const int Q = C >> 2, q = threadIdx.x % Q, pl = threadIdx.x / Q, PL = 256 / Q, c = q * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  const long r0 = (long)blockIdx.x * rows_per_wg;
  if (pl < PL)
    for (long r = r0 + pl; r < r0 + rows_per_wg && r < R; r += PL) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c), d = *reinterpret_cast<const f32x4*>(dy + r * C + c);
      f32x4 m = {1.f, 1.f, 1.f, 1.f};
      if (mask) m = *reinterpret_cast<const f32x4*>(mask + (r / rows_per_image) * C + c) * inv_keep;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xh = (v[j] - mu[j]) * rs[j];
        const float g = xh * ga[j] + be[j] > 0.f ? d[j] * m[j] : 0.f;
        a[j] += g;
        b[j] += g * xh;
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = a[j];
    red[threadIdx.x * 8 + 4 + j] = b[j];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int which = o / C, cc = o % C, qq = cc >> 2, j = cc & 3;
    float acc = 0.f;
    for (int l = 0; l < PL; ++l) acc += red[(l * Q + qq) * 8 + which * 4 + j];
    part[((long)blockIdx.x * 2 + which) * C + cc] = acc;
  }
}

// dx = gamma * rstd * (g - sum_g / R - xhat * sum_gx / R)
__global__ __launch_bounds__(256) void bnrelu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ mask, float inv_keep,
                                                               long rows_per_image, const float* __restrict__ sums /* [2][C] */, float* __restrict__ dx,
                                                               long R, int C) {
  const int Q = C >> 2;
  const long total = R * Q;
  const float invR = 1.f / (float)R;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Q) * 4;
    const long r = i / Q;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c), d = *reinterpret_cast<const f32x4*>(dy + r * C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c), s2 = *reinterpret_cast<const f32x4*>(sums + C + c);
    f32x4 m = {1.f, 1.f, 1.f, 1.f};
    if (mask) m = *reinterpret_cast<const f32x4*>(mask + (r / rows_per_image) * C + c) * inv_keep;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (v[j] - mu[j]) * rs[j];
      const float g = xh * ga[j] + be[j] > 0.f ? d[j] * m[j] : 0.f;
      o[j] = ga[j] * rs[j] * (g - s1[j] * invR - xh * s2[j] * invR);
    }
    *reinterpret_cast<f32x4*>(dx + r * C + c) = o;
  }
}

// ---- weight gradient of a Linear over MANY tokens: dW[n][k] = sum_r dY[r][n] * X[r][k], R >> N, K ---------------------------------------
// (the library GEMM picks a single-pass kernel for this tall-skinny shape: ~300 us for R = 100 352, N x K = 32 x 128.)  Split the rows
// over blockIdx.y, 64 x 64 output tile per blockIdx.x, 4 x 4 outputs per thread from LDS-staged 64-row panels; the per-split partial
// matrices are summed in a fixed order by col_reduce_kernel.
template <int TN, int TK>
__global__ __launch_bounds__(256) HPFG_NO_PK_F32 void linear_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part, long R,
                                                           int N, int K, int rows_per_split) {
  constexpr int MN = TN / 16, MK = TK / 16;          // outputs per thread: MN x MK (16 x 16 threads)
  __shared__ float sa[64][TN + 4], sb[64][TK + 4];
  const int tiles_k = (K + TK - 1) / TK;
  const int n0 = (blockIdx.x / tiles_k) * TN, k0 = (blockIdx.x % tiles_k) * TK;
  const int tn = (threadIdx.x / 16) * MN, tk = (threadIdx.x % 16) * MK;
  float acc[MN][MK];
#pragma unroll
  for (int i = 0; i < MN; ++i)
#pragma unroll
    for (int j = 0; j < MK; ++j) acc[i][j] = 0.f;
  const long r0 = (long)blockIdx.y * rows_per_split, r1 = r0 + rows_per_split < R ? r0 + rows_per_split : R;
  for (long rb = r0; rb < r1; rb += 64) {
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * (TN / 4); e += 256) {
      const int rr = e / (TN / 4), c4 = (e % (TN / 4)) * 4;
      const long r = rb + rr;
      f32x4 va = {0.f, 0.f, 0.f, 0.f};
      if (r < r1 && n0 + c4 < N) va = *reinterpret_cast<const f32x4*>(dy + r * N + n0 + c4);
      *reinterpret_cast<f32x4*>(&sa[rr][c4]) = va;
    }
    for (int e = threadIdx.x; e < 64 * (TK / 4); e += 256) {
      const int rr = e / (TK / 4), c4 = (e % (TK / 4)) * 4;
      const long r = rb + rr;
      f32x4 vb = {0.f, 0.f, 0.f, 0.f};
      if (r < r1 && k0 + c4 < K) vb = *reinterpret_cast<const f32x4*>(x + r * K + k0 + c4);
      *reinterpret_cast<f32x4*>(&sb[rr][c4]) = vb;
    }
    __syncthreads();
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) {
      float a[MN], b[MK];
#pragma unroll
      for (int i = 0; i < MN; ++i) a[i] = sa[rr][tn + i];
#pragma unroll
      for (int j = 0; j < MK; ++j) b[j] = sb[rr][tk + j];
#pragma unroll
      for (int i = 0; i < MN; ++i)
#pragma unroll
        for (int j = 0; j < MK; ++j) acc[i][j] += a[i] * b[j];
    }
  }
  float* o = part + (long)blockIdx.y * N * K;
#pragma unroll
  for (int i = 0; i < MN; ++i)
#pragma unroll
    for (int j = 0; j < MK; ++j)
      if (n0 + tn + i < N && k0 + tk + j < K) o[(long)(n0 + tn + i) * K + k0 + tk + j] = acc[i][j];
}

// ---- im2col / col2im of the overlap patch embeddings (PatchEmbed.proj, segformer.py:172: kernel k, stride s, padding k/2), NHWC ----------
// cols[b][oy][ox][(u, v, c)] = x[b][oy*s + u - p][ox*s + v - p][c] (0 outside); VEC = 4 channels per thread when C % 4 == 0
template <int VEC>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W, int C, int k, int s,
                                                     int p, int Ho, int Wo) {
  const int Cq = C / VEC;
  const long total = (long)B * Ho * Wo * k * k * Cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cq) * VEC;
    long r = i / Cq;
    const int v = (int)(r % k);
    r /= k;
    const int u = (int)(r % k);
    r /= k;
    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
    const int y = oy * s + u - p, xx = ox * s + v - p;
    const bool in = y >= 0 && y < H && xx >= 0 && xx < W;
    const long src = (((long)b * H + (in ? y : 0)) * W + (in ? xx : 0)) * C + c;
    if (VEC == 4) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (in) val = *reinterpret_cast<const f32x4*>(x + src);
      *reinterpret_cast<f32x4*>(cols + i * 4) = val;
    } else {
      cols[i] = in ? x[src] : 0.f;
    }
  }
}

// dx[b][y][x][c] = sum over the (oy, ox, u, v) whose patch element is this pixel (gather: no atomics)
template <int VEC>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int B, int H, int W, int C, int k, int s,
                                                     int p, int Ho, int Wo) {
  const int Cq = C / VEC;
  const long total = (long)B * H * W * Cq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cq) * VEC;
    const long px = i / Cq;
    const int xx = (int)(px % W), y = (int)((px / W) % H), b = (int)(px / ((long)W * H));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < k; ++u) {
      const int ty = y + p - u;
      if (ty < 0 || ty % s != 0 || ty / s >= Ho) continue;
      for (int v = 0; v < k; ++v) {
        const int tx = xx + p - v;
        if (tx < 0 || tx % s != 0 || tx / s >= Wo) continue;
        const long src = (((((long)b * Ho + ty / s) * Wo + tx / s) * k + u) * k + v) * C + c;
        if (VEC == 4) acc += *reinterpret_cast<const f32x4*>(dcols + src);
        else acc[0] += dcols[src];
      }
    }
    if (VEC == 4) *reinterpret_cast<f32x4*>(dx + px * C + c) = acc;
    else dx[px * C + c] = acc[0];
  }
}

// ---- residual branch with stochastic depth: out = x + y * s[b] (Block.forward :197-198 with DropPath :23-30; s = 1 or 0 or 1/keep) ----------
__global__ __launch_bounds__(256) void residual_scale_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ s,
                                                             float* __restrict__ out, long per_sample4, long total4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const float sc = s ? s[i / per_sample4] : 1.f;
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[i], b = reinterpret_cast<const f32x4*>(y)[i];
    reinterpret_cast<f32x4*>(out)[i] = a + b * sc;
  }
}

// dy_branch = dout * s[b]
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ d, const float* __restrict__ s, float* __restrict__ out,
                                                         long per_sample4, long total4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256)
    reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(d)[i] * s[i / per_sample4];
}

inline int grid_cap(long total, int cap) {
  long b = (total + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" int hpfg_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int C, void* stream) {
  HPFG_ARG_CHECK(x && gamma && beta && y && mean && rstd && rows > 0 && C % 4 == 0 && C >= 4 && C <= 1024, "ln_fwd: bad args (C %% 4 == 0, C <= 1024)");
  const int Q = C / 4;
  if (Q <= 64) {
#define LN_FWD_SMALL(G)                                                                                                                              \
  hipLaunchKernelGGL(ln_fwd_small_kernel<G>, dim3((unsigned)((rows + 4 * (64 / G) - 1) / (4 * (64 / G)))), dim3(256), 0, (hipStream_t)stream, x, gamma, \
                     beta, y, mean, rstd, rows, C)
    if (Q <= 8) LN_FWD_SMALL(8);
    else if (Q <= 16) LN_FWD_SMALL(16);
    else if (Q <= 32) LN_FWD_SMALL(32);
    else LN_FWD_SMALL(64);
#undef LN_FWD_SMALL
    return hpfg_launch_status("ln_fwd_small_kernel");
  }
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, rows, C);
  return hpfg_launch_status("ln_fwd_kernel");
}

extern "C" int hpfg_ln_bwd_blocks(long rows) {
  long b = (rows + 63) / 64;
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

extern "C" int hpfg_ln_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma,
                           float* dbeta, float* partials, long rows, int C, void* stream) {
  HPFG_ARG_CHECK(x && dy && gamma && mean && rstd && dx && dgamma && dbeta && partials && rows > 0 && C % 4 == 0 && C >= 4 && C <= 1024,
                 "ln_bwd: bad args");
  const int nblk = hpfg_ln_bwd_blocks(rows);
  const int per = (int)((rows + nblk - 1) / nblk);
  const int Q = C / 4;
  if (Q <= 64) {
    const int G = Q <= 8 ? 8 : (Q <= 16 ? 16 : (Q <= 32 ? 32 : 64));
    const int step = 4 * (64 / G);                        // rows one workgroup sweep covers
    const int per_s = (per + step - 1) / step * step;     // whole sweeps per workgroup
    const int nb = (int)((rows + per_s - 1) / per_s);     // <= nblk; the unused partial rows are zeroed by the extra workgroups below
    (void)nb;
#define LN_BWD_SMALL(GG) \
  hipLaunchKernelGGL(ln_bwd_small_kernel<GG>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, mean, rstd, dx, partials, rows, C, per_s)
    if (G == 8) LN_BWD_SMALL(8);
    else if (G == 16) LN_BWD_SMALL(16);
    else if (G == 32) LN_BWD_SMALL(32);
    else LN_BWD_SMALL(64);
#undef LN_BWD_SMALL
  } else {
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, dy, gamma, mean, rstd, dx, partials, rows, C, per);
  }
  // dgamma and dbeta are adjacent halves of ONE [2][C] output: a single reduction launch
  HPFG_ARG_CHECK(dbeta == dgamma + C, "ln_bwd: dgamma and dbeta must be the two halves of one [2][C] buffer");
  hipLaunchKernelGGL(col_reduce_kernel, dim3(2 * C), dim3(64), 0, (hipStream_t)stream, partials, nblk, 2 * C, dgamma);
  return hpfg_launch_status("ln_bwd_kernel");
}

extern "C" int hpfg_attn_fwd(const float* q, const float* kv, float* out, int B, int N, int M, int heads, float scale, void* stream) {
  HPFG_ARG_CHECK(q && kv && out && B > 0 && N > 0 && M > 0 && M <= AM && heads > 0, "attn_fwd: bad args (at most %d keys, head dim %d)", AM, AD);
  hipLaunchKernelGGL(attn_fwd_kernel, dim3((N + 255) / 256, heads, B), dim3(256), 0, (hipStream_t)stream, q, kv, out, N, M, heads, scale);
  return hpfg_launch_status("attn_fwd_kernel");
}

extern "C" int hpfg_attn_bwd(const float* q, const float* kv, const float* dout, float* dq, float* P, float* dS, int B, int N, int M, int heads,
                             float scale, void* stream) {
  HPFG_ARG_CHECK(q && kv && dout && dq && P && dS && B > 0 && N > 0 && M > 0 && M <= AM && heads > 0, "attn_bwd: bad args");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3((N + 255) / 256, heads, B), dim3(256), 0, (hipStream_t)stream, q, kv, dout, dq, P, dS, N, M, heads, scale);
  return hpfg_launch_status("attn_bwd_kernel");
}

extern "C" int hpfg_dwgelu_fwd(const float* x, const float* w9, const float* bias, float* y, int B, int H, int W, int C, void* stream) {
  HPFG_ARG_CHECK(x && w9 && bias && y && B > 0 && H > 0 && W > 0 && C % 4 == 0 && C >= 4, "dwgelu_fwd: bad args");
  hipLaunchKernelGGL(dwgelu_fwd_kernel, dim3(grid_cap((long)B * H * W * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x, w9, bias, y, B, H, W, C);
  return hpfg_launch_status("dwgelu_fwd_kernel");
}

extern "C" int hpfg_dwgelu_bwd_blocks(int B, int H, int W) {
  long b = ((long)B * H * W + 31) / 32;            // >= 8 workgroups per CU at the 56x56 stage: the pass is latency bound with fewer
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int hpfg_dwgelu_bwd(const float* x, const float* w9, const float* bias, const float* dy, float* du, float* dx, float* dw9, float* dbias,
                               float* partials, int B, int H, int W, int C, void* stream) {
  HPFG_ARG_CHECK(x && w9 && bias && dy && du && dx && dw9 && dbias && partials && B > 0 && H > 0 && W > 0 && C % 4 == 0 && C >= 4, "dwgelu_bwd: bad args");
  const int nblk = hpfg_dwgelu_bwd_blocks(B, H, W);
  const int Q = C / 4, Qw = Q < DWG_QW ? Q : DWG_QW;
  hipLaunchKernelGGL(dwgelu_bwd_du_kernel, dim3(nblk, (Q + Qw - 1) / Qw), dim3(256), 0, (hipStream_t)stream, x, w9, bias, dy, du, partials, B, H, W, C);
  hipLaunchKernelGGL(dw_bwd_dx_kernel, dim3(grid_cap((long)B * H * W * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, du, w9, dx, B, H, W, C);
  HPFG_ARG_CHECK(dbias == dw9 + 9 * C, "dwgelu_bwd: dw9 and dbias must be one [10][C] buffer");
  hipLaunchKernelGGL(col_reduce_kernel, dim3(10 * C), dim3(64), 0, (hipStream_t)stream, partials, nblk, 10 * C, dw9);
  return hpfg_launch_status("dwgelu_bwd_kernel");
}

extern "C" int hpfg_resize_bilinear_fwd(const float* x, float* y, int B, int h, int w, int H, int W, int C, void* stream) {
  HPFG_ARG_CHECK(x && y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C % 4 == 0 && C >= 4, "resize_bilinear_fwd: bad args");
  hipLaunchKernelGGL(resize_fwd_kernel, dim3(grid_cap((long)B * H * W * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x, y, B, h, w, H, W, C,
                     (float)h / (float)H, (float)w / (float)W);
  return hpfg_launch_status("resize_fwd_kernel");
}

extern "C" int hpfg_resize_sum_fwd(const float* base, const float* const* xs, const int* hs, const int* ws, int nsrc, float* y, int B, int H, int W, int C,
                                   void* stream) {
  HPFG_ARG_CHECK(base && xs && hs && ws && y && nsrc >= 0 && nsrc <= 3 && B > 0 && H > 0 && W > 0 && C % 4 == 0 && C >= 4, "resize_sum_fwd: bad args");
  ResizeSrc s[3] = {{nullptr, 1, 1, 1.f, 1.f}, {nullptr, 1, 1, 1.f, 1.f}, {nullptr, 1, 1, 1.f, 1.f}};
  for (int k = 0; k < nsrc; ++k) {
    HPFG_ARG_CHECK(xs[k] && hs[k] > 0 && ws[k] > 0, "resize_sum_fwd: bad source %d", k);
    s[k] = ResizeSrc{xs[k], hs[k], ws[k], (float)hs[k] / (float)H, (float)ws[k] / (float)W};
  }
  hipLaunchKernelGGL(resize_sum_fwd_kernel, dim3(grid_cap((long)B * H * W * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, base, s[0], s[1], s[2], nsrc,
                     y, B, H, W, C);
  return hpfg_launch_status("resize_sum_fwd_kernel");
}

extern "C" int hpfg_resize_bilinear_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, int C, void* stream) {
  HPFG_ARG_CHECK(dy && dx && B > 0 && h > 0 && w > 0 && H >= h && W >= w && C % 4 == 0 && C >= 4, "resize_bilinear_bwd: bad args (upsampling only)");
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_cap((long)B * h * w * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, h, w, H, W, C,
                     (float)h / (float)H, (float)w / (float)W);
  return hpfg_launch_status("resize_bwd_kernel");
}

extern "C" int hpfg_tok_stat_blocks(long R) {
  long b = (R + 63) / 64;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

static int tok_c_ok(int C) { return C % 4 == 0 && C >= 4 && C <= 1024 && 256 % (C / 4) == 0; }

/* sums[0][C] = sum_r x, sums[1][C] = sum_r x^2 */
extern "C" int hpfg_tok_col_stats(const float* x, long R, int C, float* partials, float* sums, void* stream) {
  HPFG_ARG_CHECK(x && partials && sums && R > 0 && tok_c_ok(C), "tok_col_stats: bad args (C/4 must divide 256)");
  const int nblk = hpfg_tok_stat_blocks(R);
  hipLaunchKernelGGL(col_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, R, C, partials, (int)((R + nblk - 1) / nblk));
  hipLaunchKernelGGL(col_reduce_kernel, dim3(2 * C), dim3(64), 0, (hipStream_t)stream, partials, nblk, 2 * C, sums);
  return hpfg_launch_status("col_stats_kernel");
}

extern "C" int hpfg_bnrelu_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* mask,
                                 float inv_keep, long rows_per_image, float* y, long R, int C, void* stream) {
  HPFG_ARG_CHECK(x && mean && rstd && gamma && beta && y && R > 0 && C % 4 == 0 && rows_per_image > 0, "bnrelu_apply: bad args");
  hipLaunchKernelGGL(bnrelu_apply_kernel, dim3(grid_cap(R * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, mask,
                     inv_keep, rows_per_image, y, R, C);
  return hpfg_launch_status("bnrelu_apply_kernel");
}

extern "C" int hpfg_bnrelu_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const float* mask, float inv_keep, long rows_per_image, float* dx, float* partials, float* sums, long R, int C,
                               void* stream) {
  HPFG_ARG_CHECK(x && dy && mean && rstd && gamma && beta && dx && partials && sums && R > 0 && tok_c_ok(C) && rows_per_image > 0, "bnrelu_bwd: bad args");
  const int nblk = hpfg_tok_stat_blocks(R);
  hipLaunchKernelGGL(bnrelu_bwd_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, dy, mean, rstd, gamma, beta, mask, inv_keep,
                     rows_per_image, R, C, partials, (int)((R + nblk - 1) / nblk));
  hipLaunchKernelGGL(col_reduce_kernel, dim3(2 * C), dim3(64), 0, (hipStream_t)stream, partials, nblk, 2 * C, sums);
  hipLaunchKernelGGL(bnrelu_bwd_apply_kernel, dim3(grid_cap(R * (C / 4), 8192)), dim3(256), 0, (hipStream_t)stream, x, dy, mean, rstd, gamma, beta, mask,
                     inv_keep, rows_per_image, sums, dx, R, C);
  return hpfg_launch_status("bnrelu_bwd_kernel");
}

static void lw_tile(int N, int K, int& TN, int& TK) {
  TN = N <= 32 ? 32 : 64;
  TK = K <= 32 ? 32 : 64;
}

extern "C" int hpfg_linear_wgrad_splits(long R, int N, int K) {
  int TN, TK;
  lw_tile(N, K, TN, TK);
  const long tiles = (long)((N + TN - 1) / TN) * ((K + TK - 1) / TK);
  long s = 2048 / tiles;                       // ~8 workgroups per CU in total
  const long by_rows = (R + 255) / 256;        // at least 256 rows per split
  if (s > by_rows) s = by_rows;
  return (int)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

/* dW[N][K] = dY^T X over R rows; partials [hpfg_linear_wgrad_splits()][N][K] */
extern "C" int hpfg_linear_wgrad(const float* dy, const float* x, float* dw, float* partials, long R, int N, int K, void* stream) {
  HPFG_ARG_CHECK(dy && x && dw && partials && R > 0 && N % 4 == 0 && K % 4 == 0 && N >= 4 && K >= 4, "linear_wgrad: bad args (N, K multiples of 4)");
  const int S = hpfg_linear_wgrad_splits(R, N, K);
  int per = (int)((R + S - 1) / S);
  per = (per + 63) / 64 * 64;
  int TN, TK;
  lw_tile(N, K, TN, TK);
  dim3 grid(((N + TN - 1) / TN) * ((K + TK - 1) / TK), S);
  if (TN == 32 && TK == 32) hipLaunchKernelGGL((linear_wgrad_kernel<32, 32>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, partials, R, N, K, per);
  else if (TN == 32) hipLaunchKernelGGL((linear_wgrad_kernel<32, 64>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, partials, R, N, K, per);
  else if (TK == 32) hipLaunchKernelGGL((linear_wgrad_kernel<64, 32>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, partials, R, N, K, per);
  else hipLaunchKernelGGL((linear_wgrad_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, dy, x, partials, R, N, K, per);
  const long NK = (long)N * K;
  hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((NK + 63) / 64)), dim3(256), 0, (hipStream_t)stream, partials, S, NK, dw);
  return hpfg_launch_status("linear_wgrad_kernel");
}

extern "C" int hpfg_im2col_nhwc(const float* x, float* cols, int B, int H, int W, int C, int k, int s, void* stream) {
  HPFG_ARG_CHECK(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0, "im2col_nhwc: bad args");
  const int p = k / 2, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  const long total = (long)B * Ho * Wo * k * k * (C % 4 == 0 ? C / 4 : C);
  if (C % 4 == 0) hipLaunchKernelGGL(im2col_kernel<4>, dim3(grid_cap(total, 16384)), dim3(256), 0, (hipStream_t)stream, x, cols, B, H, W, C, k, s, p, Ho, Wo);
  else hipLaunchKernelGGL(im2col_kernel<1>, dim3(grid_cap(total, 16384)), dim3(256), 0, (hipStream_t)stream, x, cols, B, H, W, C, k, s, p, Ho, Wo);
  return hpfg_launch_status("im2col_kernel");
}

extern "C" int hpfg_col2im_nhwc(const float* dcols, float* dx, int B, int H, int W, int C, int k, int s, void* stream) {
  HPFG_ARG_CHECK(dcols && dx && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && s > 0, "col2im_nhwc: bad args");
  const int p = k / 2, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  const long total = (long)B * H * W * (C % 4 == 0 ? C / 4 : C);
  if (C % 4 == 0) hipLaunchKernelGGL(col2im_kernel<4>, dim3(grid_cap(total, 16384)), dim3(256), 0, (hipStream_t)stream, dcols, dx, B, H, W, C, k, s, p, Ho, Wo);
  else hipLaunchKernelGGL(col2im_kernel<1>, dim3(grid_cap(total, 16384)), dim3(256), 0, (hipStream_t)stream, dcols, dx, B, H, W, C, k, s, p, Ho, Wo);
  return hpfg_launch_status("col2im_kernel");
}

extern "C" int hpfg_residual_scale(const float* x, const float* y, const float* scale, float* out, int B, long per_sample, void* stream) {
  HPFG_ARG_CHECK(x && y && out && B > 0 && per_sample > 0 && per_sample % 4 == 0, "residual_scale: bad args");
  const long total4 = (long)B * per_sample / 4;
  hipLaunchKernelGGL(residual_scale_kernel, dim3(grid_cap(total4, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, scale, out, per_sample / 4, total4);
  return hpfg_launch_status("residual_scale_kernel");
}

extern "C" int hpfg_scale_rows(const float* d, const float* scale, float* out, int B, long per_sample, void* stream) {
  HPFG_ARG_CHECK(d && scale && out && B > 0 && per_sample > 0 && per_sample % 4 == 0, "scale_rows: bad args");
  const long total4 = (long)B * per_sample / 4;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_cap(total4, 8192)), dim3(256), 0, (hipStream_t)stream, d, scale, out, per_sample / 4, total4);
  return hpfg_launch_status("scale_rows_kernel");
}
