// BatchNorm2d (train mode) statistics, forward finalize and backward reduction for the fused conv pipeline.
// Reference: nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1, affine, running stats with UNBIASED variance) at
// model/unet.py:19,23; backward = torch's native_batch_norm_backward (train) composed with LeakyReLU / Dropout backward.
//
// Forward:  the conv epilogue leaves per-workgroup partial sums  S1 = sum z, S2 = sum z^2  per channel.
//           finalize (one workgroup per channel, fp64 accumulation) turns them into the table rows
//           mean, rstd, scale = gamma*rstd, shift = beta - mean*scale  and updates running_mean / running_var.
// Backward: with g = dL/dy (after LeakyReLU'/Dropout), xhat = (z-mean)*rstd, M = N*H*W:
//           dz = gamma*rstd * (g - mean(g) - xhat*mean(g*xhat));  dgamma = sum g*xhat;  dbeta = sum g.
//           The reduce kernel produces partial (sum g, sum g*xhat); finalize writes dgamma/dbeta and the rows
//           k1 = gamma*rstd, k2 = -gamma*rstd^2*m2, k3 = gamma*rstd*(mean*rstd*m2 - m1)  so that dz = k1*g + k2*z + k3,
//           which the dgrad / wgrad loaders evaluate on the fly (HPFG_ACT_DZ).
// Data parallel: hpfg_reduce_partials gives fp64 [2][C] sums to all-reduce; finalize then takes `sums` instead of partials.
#include <string.h>
#include "common.h"
#include "peer.h"

namespace {

__device__ inline double block_sum(double v, double* sh) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  __syncthreads();
  return t;
}

// Both per-channel sums of a finalize kernel in ONE reduction round (these kernels are pure latency: a handful of dependent
// steps on the critical path of every BatchNorm layer, so the loads are issued together and there is a single barrier).
__device__ inline void block_sum2(double& a, double& b, double* sh8) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if ((tid & 63) == 0) {
    sh8[(tid >> 6) * 2] = a;
    sh8[(tid >> 6) * 2 + 1] = b;
  }
  __syncthreads();
  a = (sh8[0] + sh8[2]) + (sh8[4] + sh8[6]);
  b = (sh8[1] + sh8[3]) + (sh8[5] + sh8[7]);
}

// Partial rows [nblk][2][C] -> the two fp64 sums of channel c; up to 4 rows per thread are requested before any is consumed.
__device__ inline void sum_partials(const float* __restrict__ partials, int nblk, int C, int c, double& a, double& b, double* sh8) {
  a = 0.0;
  b = 0.0;
  for (int i0 = threadIdx.x; i0 < nblk; i0 += 1024) {
    float x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * 256;
      const bool ok = i < nblk;
      const long o = (long)(ok ? i : i0) * 2 * C + c;
      x[k] = ok ? partials[o] : 0.f;
      y[k] = ok ? partials[o + C] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a += (double)x[k];
      b += (double)y[k];
    }
  }
  block_sum2(a, b, sh8);
}

// grid = C blocks; partials [nblk][2][C]
struct HpfgBnFinalizeArgs {
  const float* partials;
  const float* gamma;
  const float* beta;
  float* running_mean;  // or NULL: statistics not tracked
  float* running_var;
  float* bn;
};
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(HpfgBnFinalizeArgs q, int nblk, const double* __restrict__ sums, double count, float momentum,
                                                              float eps, int C, HpfgPeerX px) {
  __shared__ double sh[8];
  const float* __restrict__ partials = q.partials;
  const float* __restrict__ gamma = q.gamma;
  const float* __restrict__ beta = q.beta;
  float* running_mean = q.running_mean;
  float* running_var = q.running_var;
  float* __restrict__ bn = q.bn;
  const int c = blockIdx.x;
  const float ga = gamma[c], be = beta[c];                       // requested up front: off the dependent chain below
  const float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
  double s1, s2;
  if (sums) {
    s1 = sums[c];
    s2 = sums[C + c];
  } else {
    sum_partials(partials, nblk, C, c, s1, s2, sh);
  }
  if (threadIdx.x == 0) {
    if (px.world > 1) {          // data parallel, global-batch mode: add the ranks' sums of this channel (peer mailbox, rank order)
      const int idx[2] = {c, C + c};
      double v[2] = {s1, s2};
      hpfg_peer_allreduce<2>(px, idx, v);
      s1 = v[0];
      s2 = v[1];
    }
    const HpfgBnCoef q = hpfg_bn_coef(s1, s2, count, eps, ga, be);      // (the definition the accumulator path's consumers share: same bits)
    const double mean = s1 / count, var = q.var;
    bn[HPFG_BN_MEAN * C + c] = q.mean;
    bn[HPFG_BN_RSTD * C + c] = q.rstd;
    bn[HPFG_BN_SCALE * C + c] = q.scale;
    bn[HPFG_BN_SHIFT * C + c] = q.shift;
    if (running_mean) {
      double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)rm + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * (double)rv + momentum * unb);
    }
  }
}

// Every BatchNorm layer of a forward pass in ONE launch, from the layers' sum accumulators (HpfgConvArgs.stat_acc): table rows mean / rstd /
// scale / shift for the backward kernels and the running statistics.  The forward consumers derived the same scale / shift themselves
// (hpfg_bn_coef, the shared definition), so this launch sits at the END of the forward, off every conv -> conv dependency.
__global__ __launch_bounds__(256) void bn_acc_finalize_kernel(const HpfgBnAccDesc* __restrict__ table, float momentum, float eps) {
  const HpfgBnAccDesc d = table[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.C) return;
  const double count = (double)d.count;
  const float ga = d.gamma[c], be = d.beta[c];
  double s1, s2;
  hpfg_acc_read2(d.acc, d.C, d.shards, c, s1, s2, ga + be);
  const HpfgBnCoef q = hpfg_bn_coef(s1, s2, count, eps, ga, be);
  d.bn[HPFG_BN_MEAN * d.C + c] = q.mean;
  d.bn[HPFG_BN_RSTD * d.C + c] = q.rstd;
  d.bn[HPFG_BN_SCALE * d.C + c] = q.scale;
  d.bn[HPFG_BN_SHIFT * d.C + c] = q.shift;
  if (d.running_mean) {
    const double mean = s1 / count;
    const double unb = count > 1.0 ? q.var * count / (count - 1.0) : q.var;
    d.running_mean[c] = (float)((1.0 - momentum) * (double)d.running_mean[c] + momentum * mean);
    d.running_var[c] = (float)((1.0 - momentum) * (double)d.running_var[c] + momentum * unb);
  }
  // every consumer of this forward has run (the launch follows the last conv): leave the accumulator zeroed for the next forward
  long long* acc = const_cast<long long*>(d.acc);
  for (int s = 0; s < d.shards; ++s)
    for (int w = 0; w < 2; ++w) {
      long long* b = acc + ((long)((s * 2 + w) * d.C + c)) * 2;
      b[0] = 0;
      b[1] = 0;
    }
}

// Backward counterpart, at the END of a backward pass (or of its decoder half): dgamma / dbeta of every listed layer from its backward sum
// accumulator (+ the k1 .. k3 table rows, for readers of the table), after which the accumulator is ZEROED for the next pass -- every dZ
// consumer of the pass has run by then (the caller joins its side stream first).
__global__ __launch_bounds__(256) void bn_acc_bwd_finalize_kernel(const HpfgBnAccBwdDesc* __restrict__ table) {
  const HpfgBnAccBwdDesc d = table[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.C) return;
  double sg, sgx;
  hpfg_acc_read2(d.acc, d.C, d.shards, c, sg, sgx);
  const HpfgBnBwdCoef q = hpfg_bn_bwd_coef(sg, sgx, (double)d.count, (double)d.bn[HPFG_BN_MEAN * d.C + c], (double)d.bn[HPFG_BN_RSTD * d.C + c], (double)d.gamma[c]);
  d.bn[HPFG_BN_K1 * d.C + c] = q.k1;
  d.bn[HPFG_BN_K2 * d.C + c] = q.k2;
  d.bn[HPFG_BN_K3 * d.C + c] = q.k3;
  if (d.dgamma) d.dgamma[c] = (float)sgx;
  if (d.dbeta) d.dbeta[c] = (float)sg;
  for (int s = 0; s < d.shards; ++s)
    for (int w = 0; w < 2; ++w) {
      long long* b = d.acc + ((long)((s * 2 + w) * d.C + c)) * 2;
      b[0] = 0;
      b[1] = 0;
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int nblk, int C, double* __restrict__ sums) {
  __shared__ double sh[4];
  const int c = blockIdx.x, which = blockIdx.y;
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += (double)partials[((long)i * 2 + which) * C + c];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) sums[which * C + c] = a;
}

// Backward reduction: streaming pass over (dA, z).  Thread t owns channel quad t % Q (its BatchNorm table lives in registers)
// and pixel lane t / Q; a workgroup strides over the pixels with four independent pixel loads in flight per thread.
constexpr int BWD_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(HpfgAct s, long npix, float* __restrict__ partials, long long* __restrict__ acc, int shards) {
  __shared__ float red[256 * 8];
  const int C = s.C, Q = C >> 2, tid = threadIdx.x;
  const ActCtx cx = make_ctx(s);
  const int q = tid % Q, pl = tid / Q, PL = 256 / Q;
  const int c = q * 4;
  const float* t = s.bn + s.bn_coff + c;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(t + HPFG_BN_MEAN * s.bn_stride);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(t + HPFG_BN_RSTD * s.bn_stride);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SCALE * s.bn_stride);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SHIFT * s.bn_stride);
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  const long stride = (long)gridDim.x * PL;
  for (long p0 = (long)blockIdx.x * PL + pl; p0 < npix; p0 += 4 * stride) {
    f32x4 z[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + u * stride;
      if (pix < npix) {
        z[u] = *reinterpret_cast<const f32x4*>(s.z + pix * s.pstride + c);
        g[u] = *reinterpret_cast<const f32x4*>(s.aux + pix * s.aux_pstride + c);
      } else {
        z[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        g[u] = z[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long pix = p0 + u * stride;
      uint32_t km = 0xFu;
      if (s.drop_p > 0.f && pix < npix) km = keep4(s, cx, (uint32_t)(pix * C + c));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = (km >> j) & 1u ? g[u][j] * cx.inv_keep : 0.f;
        gg = z[u][j] * sc[j] + sh[j] > 0.f ? gg : HPFG_LEAKY * gg;
        a[j] += gg;
        b[j] += gg * ((z[u][j] - mu[j]) * rs[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[tid * 8 + j] = a[j];
    red[tid * 8 + 4 + j] = b[j];
  }
  __syncthreads();
  for (int o = tid; o < 2 * C; o += 256) {
    const int which = o / C, cc = o % C, qq = cc >> 2, j = cc & 3;
    float t = 0.f;
    for (int l = 0; l < PL; ++l) t += red[(l * Q + qq) * 8 + which * 4 + j];
    if (acc) hpfg_acc_add(acc, C, (int)blockIdx.x & (shards - 1), which, cc, t);      // (the layer's backward accumulator: HpfgAct.bn_acc of its dZ consumers)
    if (partials) partials[((long)blockIdx.x * 2 + which) * C + cc] = t;
  }
}

// The same reduction for a layer whose output also feeds a MaxPool2d(2) (encoder block outputs, model/unet.py:37): its gradient is
// dA (the skip path, already in place) + the max-pool backward of dP.  One pass does both: a thread takes a 2x2 window and a channel
// quad, recomputes the arg-max from the raw output (first maximum in row-major window order, like the forward kernel's max and
// torch's max_pool2d), adds dP there, writes the completed dA back for the dgrad / wgrad loaders, and accumulates the BatchNorm
// sums from the registers.  Replaces pool_scatter_add + bn_bwd_reduce (one read of z and dA and one launch less).
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(HpfgAct s, const float* __restrict__ dP, int dp_ps, long npool, int Hp, int Wp,
                                                                 float* __restrict__ partials, long long* __restrict__ acc, int shards) {
  __shared__ float red[256 * 8];
  const int C = s.C, Q = C >> 2, tid = threadIdx.x;
  const int q = tid % Q, pl = tid / Q, PL = 256 / Q;
  const int c = q * 4;
  const float* t = s.bn + s.bn_coff + c;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(t + HPFG_BN_MEAN * s.bn_stride);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(t + HPFG_BN_RSTD * s.bn_stride);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SCALE * s.bn_stride);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SHIFT * s.bn_stride);
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  float* dA = const_cast<float*>(s.aux);
  // Two windows per trip with every load of both requested before the first is used: a thread's windows are independent, but the compiler
  // must assume that the dA stores of one alias the dA loads of the next and would otherwise wait out each trip's stores (the launch was a
  // chain of exposed round trips: 22 us for 19 MB at the 28 x 28 level).
  const long stride = (long)gridDim.x * PL;
  for (long pp0 = (long)blockIdx.x * PL + pl; pp0 < npool; pp0 += 2 * stride) {
    f32x4 z[2][4], g[2][4], gp[2];
    long pos[2][4];
    bool on[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long pq = pp0 + u * stride;
      on[u] = pq < npool;
      const long pp = on[u] ? pq : pp0;
      const int xp = (int)(pp % Wp), yp = (int)((pp / Wp) % Hp), n = (int)(pp / ((long)Wp * Hp));
      const long p00 = (long)(n * s.Hs + 2 * yp) * s.Ws + 2 * xp;
      pos[u][0] = p00;
      pos[u][1] = p00 + 1;
      pos[u][2] = p00 + s.Ws;
      pos[u][3] = p00 + s.Ws + 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        z[u][k] = *reinterpret_cast<const f32x4*>(s.z + pos[u][k] * s.pstride + c);
        g[u][k] = *reinterpret_cast<const f32x4*>(dA + pos[u][k] * s.aux_pstride + c);
      }
      gp[u] = *reinterpret_cast<const f32x4*>(dP + pp * dp_ps + c);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!on[u]) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = z[u][k][j] * sc[j] + sh[j];
        float best = lrelu(y[0]);
        int bi = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float v = lrelu(y[k]);
          if (v > best) {
            best = v;
            bi = k;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          g[u][k][j] += bi == k ? gp[u][j] : 0.f;
          const float gg = y[k] > 0.f ? g[u][k][j] : HPFG_LEAKY * g[u][k][j];
          a[j] += gg;
          b[j] += gg * ((z[u][k][j] - mu[j]) * rs[j]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dA + pos[u][k] * s.aux_pstride + c) = g[u][k];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[tid * 8 + j] = a[j];
    red[tid * 8 + 4 + j] = b[j];
  }
  __syncthreads();
  for (int o = tid; o < 2 * C; o += 256) {
    const int which = o / C, cc = o % C, qq = cc >> 2, j = cc & 3;
    float t = 0.f;
    for (int l = 0; l < PL; ++l) t += red[(l * Q + qq) * 8 + which * 4 + j];
    if (acc) hpfg_acc_add(acc, C, (int)blockIdx.x & (shards - 1), which, cc, t);      // (the layer's backward accumulator: HpfgAct.bn_acc of its dZ consumers)
    if (partials) partials[((long)blockIdx.x * 2 + which) * C + cc] = t;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, const double* __restrict__ sums,
                                                              double count, const float* __restrict__ gamma, float* __restrict__ bn,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int C, float pscale, HpfgPeerX px) {
  __shared__ double sh[8];
  const int c = blockIdx.x;
  const double mean = bn[HPFG_BN_MEAN * C + c], rstd = bn[HPFG_BN_RSTD * C + c], ga = gamma[c];
  double sg, sgx;
  if (sums) {
    sg = sums[c];
    sgx = sums[C + c];
  } else {
    sum_partials(partials, nblk, C, c, sg, sgx, sh);
  }
  if (threadIdx.x == 0) {
    if (px.world > 1) {
      const int idx[2] = {c, C + c};
      double v[2] = {sg, sgx};
      hpfg_peer_allreduce<2>(px, idx, v);
      sg = v[0];
      sgx = v[1];
    }
    const HpfgBnBwdCoef q = hpfg_bn_bwd_coef(sg, sgx, count, mean, rstd, ga);      // (the definition the accumulator path's consumers share)
    bn[HPFG_BN_K1 * C + c] = q.k1;
    bn[HPFG_BN_K2 * C + c] = q.k2;
    bn[HPFG_BN_K3 * C + c] = q.k3;
    if (dgamma) dgamma[c] = (float)(sgx * pscale);
    if (dbeta) dbeta[c] = (float)(sg * pscale);
  }
}

__global__ __launch_bounds__(256) void bn_eval_table_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                                            float* __restrict__ bn, int C) {
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float rstd = 1.f / sqrtf(rv[c] + eps);
  float scale = gamma[c] * rstd;
  bn[HPFG_BN_MEAN * C + c] = rm[c];
  bn[HPFG_BN_RSTD * C + c] = rstd;
  bn[HPFG_BN_SCALE * C + c] = scale;
  bn[HPFG_BN_SHIFT * C + c] = beta[c] - rm[c] * scale;
}

}  // namespace

extern "C" int hpfg_bn_eval_table(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                  float* bn, int C, void* stream) {
  HPFG_ARG_CHECK(gamma && beta && running_mean && running_var && bn && C > 0, "bn_eval_table: bad args");
  hipLaunchKernelGGL(bn_eval_table_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps,
                     bn, C);
  return hpfg_launch_status("bn_eval_table_kernel");
}

extern "C" int hpfg_bn_fwd_finalize(const float* partials, int nblk, const double* sums, double count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* bn, int C, void* stream) {
  HPFG_ARG_CHECK((partials && nblk > 0) || sums, "bn_fwd_finalize: need partials or sums");
  HPFG_ARG_CHECK(gamma && beta && bn && C > 0 && count > 0, "bn_fwd_finalize: bad args");
  const HpfgBnFinalizeArgs q{partials, gamma, beta, running_mean, running_var, bn};
  HpfgPeerX none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, q, nblk, sums, count, momentum, eps, C, none);
  return hpfg_launch_status("bn_fwd_finalize_kernel");
}

static int check_px(const HpfgPeerX* px, int C, const char* who) {
  HPFG_ARG_CHECK(px, "%s: null exchange descriptor", who);
  if (px->world <= 1) return 0;
  HPFG_ARG_CHECK(px->world <= HPFG_PEER_MAX_RANKS && px->rank >= 0 && px->rank < px->world && px->epoch && px->slot >= 0 && 2 * C <= px->cap &&
                     px->slot_bytes == hpfg_peer_slot_bytes(px->world, px->cap),
                 "%s: bad exchange descriptor (world %d rank %d cap %d for 2 x %d values)", who, px->world, px->rank, px->cap, C);
  for (int r = 0; r < px->world; ++r) HPFG_ARG_CHECK(px->mbox[r], "%s: mailbox of rank %d not mapped", who, r);
  return 0;
}

extern "C" long hpfg_peer_slot_bytes(int world, int cap) { return (long)2 * world * cap * (long)(sizeof(double) + sizeof(uint32_t)); }

extern "C" int hpfg_bn_fwd_finalize_x(const float* partials, int nblk, const HpfgPeerX* px, double count, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* bn, int C, void* stream) {
  HPFG_ARG_CHECK(partials && nblk > 0 && gamma && beta && bn && C > 0 && count > 0, "bn_fwd_finalize_x: bad args");
  if (int rc = check_px(px, C, "bn_fwd_finalize_x")) return rc;
  const HpfgBnFinalizeArgs q{partials, gamma, beta, running_mean, running_var, bn};
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, q, nblk, nullptr, count, momentum, eps, C, *px);
  return hpfg_launch_status("bn_fwd_finalize_kernel");
}

extern "C" int hpfg_bn_acc_finalize(const HpfgBnAccDesc* table_dev, const HpfgBnAccDesc* table_host, int nlayers, float momentum, float eps, void* stream) {
  HPFG_ARG_CHECK(table_dev && table_host && nlayers > 0 && nlayers < 65536, "bn_acc_finalize: bad args");
  int maxc = 0;
  for (int i = 0; i < nlayers; ++i) {
    const HpfgBnAccDesc& d = table_host[i];
    HPFG_ARG_CHECK(d.acc && d.gamma && d.beta && d.bn && d.C > 0 && d.count >= 1.f && (d.running_mean == nullptr) == (d.running_var == nullptr) &&
                       d.shards >= 1 && d.shards <= HPFG_ACC_MAX_SHARDS && (d.shards & (d.shards - 1)) == 0,
                   "bn_acc_finalize: bad descriptor %d", i);
    HPFG_ARG_CHECK(table_host[i].count <= 16777216.f, "bn_acc_finalize: count of descriptor %d above 2^24 (a float holds N*H*W exactly up to there)", i);
    if (d.C > maxc) maxc = d.C;
  }
  hipLaunchKernelGGL(bn_acc_finalize_kernel, dim3((maxc + 255) / 256, nlayers), dim3(256), 0, (hipStream_t)stream, table_dev, momentum, eps);
  return hpfg_launch_status("bn_acc_finalize_kernel");
}

extern "C" int hpfg_bn_acc_bwd_finalize(const HpfgBnAccBwdDesc* table_dev, const HpfgBnAccBwdDesc* table_host, int nlayers, void* stream) {
  HPFG_ARG_CHECK(table_dev && table_host && nlayers > 0 && nlayers < 65536, "bn_acc_bwd_finalize: bad args");
  int maxc = 0;
  for (int i = 0; i < nlayers; ++i) {
    const HpfgBnAccBwdDesc& d = table_host[i];
    HPFG_ARG_CHECK(d.acc && d.gamma && d.bn && d.C > 0 && d.count >= 1.f && d.shards >= 1 && d.shards <= HPFG_ACC_MAX_SHARDS && (d.shards & (d.shards - 1)) == 0,
                   "bn_acc_bwd_finalize: bad descriptor %d", i);
    HPFG_ARG_CHECK(table_host[i].count <= 16777216.f, "bn_acc_bwd_finalize: count of descriptor %d above 2^24", i);
    if (d.C > maxc) maxc = d.C;
  }
  hipLaunchKernelGGL(bn_acc_bwd_finalize_kernel, dim3((maxc + 255) / 256, nlayers), dim3(256), 0, (hipStream_t)stream, table_dev);
  return hpfg_launch_status("bn_acc_bwd_finalize_kernel");
}

extern "C" int hpfg_reduce_partials(const float* partials, int nblk, int C, double* sums, void* stream) {
  HPFG_ARG_CHECK(partials && sums && nblk > 0 && C > 0, "reduce_partials: bad args");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(C, 2), dim3(256), 0, (hipStream_t)stream, partials, nblk, C, sums);
  return hpfg_launch_status("reduce_partials_kernel");
}

extern "C" int hpfg_bn_bwd_blocks(int N, int H, int W, int C) {
  const long npix = (long)N * H * W;
  const int PL = 256 / (C / 4);                       // pixels a workgroup covers per sweep
  long want = (npix + (long)PL * 16 - 1) / ((long)PL * 16);   // ~16 pixels per thread
  if (want < 1) want = 1;
  return (int)(want > BWD_MAX_BLOCKS ? BWD_MAX_BLOCKS : want);
}

static int bn_bwd_reduce_impl(const HpfgAct* g, int N, int H, int W, float* partials, long long* acc, int shards, void* stream);
extern "C" int hpfg_bn_bwd_reduce(const HpfgAct* g, int N, int H, int W, float* partials, void* stream) {
  return bn_bwd_reduce_impl(g, N, H, W, partials, nullptr, 1, stream);
}
extern "C" int hpfg_bn_bwd_reduce_acc(const HpfgAct* g, int N, int H, int W, long long* acc, int shards, void* stream) {
  HPFG_ARG_CHECK(acc && shards >= 1 && shards <= HPFG_ACC_MAX_SHARDS && (shards & (shards - 1)) == 0, "bn_bwd_reduce_acc: bad accumulator / shards");
  return bn_bwd_reduce_impl(g, N, H, W, nullptr, acc, shards, stream);
}
static int bn_bwd_reduce_impl(const HpfgAct* g, int N, int H, int W, float* partials, long long* acc, int shards, void* stream) {
  HPFG_ARG_CHECK(g && (partials || acc) && g->mode == HPFG_ACT_DZ && g->z && g->aux && g->bn, "bn_bwd_reduce: needs a DZ source");
  HPFG_ARG_CHECK(g->C % 4 == 0 && g->C >= 4 && g->C <= 1024 && 256 % (g->C / 4) == 0, "bn_bwd_reduce: unsupported C=%d", g->C);
  HPFG_ARG_CHECK(g->Hs == H && g->Ws == W, "bn_bwd_reduce: source size mismatch");
  long npix = (long)N * H * W;
  int nblk = hpfg_bn_bwd_blocks(N, H, W, g->C);
  if (acc) HPFG_ACC_CHECK(nblk, shards, "bn_bwd_reduce_acc");
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *g, npix, partials, acc, shards);
  return hpfg_launch_status("bn_bwd_reduce_kernel");
}

extern "C" int hpfg_bn_bwd_pool_blocks(int N, int Hp, int Wp, int C) {
  const long npool = (long)N * Hp * Wp;
  const int PL = 256 / (C / 4);
  long want = (npool + (long)PL * 2 - 1) / ((long)PL * 2);      // 2 windows = 8 pixels per thread: one trip of the kernel's loop
  if (want < 1) want = 1;
  return (int)(want > 2 * BWD_MAX_BLOCKS ? 2 * BWD_MAX_BLOCKS : want);      // (224 -> 112 at 16 images: 1568 -- one full trip each, not one and a half)
}

static int bn_bwd_reduce_pool_impl(const HpfgAct* g, const float* dP, int dp_pstride, int N, int Hp, int Wp, float* partials, long long* acc, int shards,
                                   void* stream);
extern "C" int hpfg_bn_bwd_reduce_pool(const HpfgAct* g, const float* dP, int dp_pstride, int N, int Hp, int Wp, float* partials, void* stream) {
  return bn_bwd_reduce_pool_impl(g, dP, dp_pstride, N, Hp, Wp, partials, nullptr, 1, stream);
}
extern "C" int hpfg_bn_bwd_reduce_pool_acc(const HpfgAct* g, const float* dP, int dp_pstride, int N, int Hp, int Wp, long long* acc, int shards, void* stream) {
  HPFG_ARG_CHECK(acc && shards >= 1 && shards <= HPFG_ACC_MAX_SHARDS && (shards & (shards - 1)) == 0, "bn_bwd_reduce_pool_acc: bad accumulator / shards");
  return bn_bwd_reduce_pool_impl(g, dP, dp_pstride, N, Hp, Wp, nullptr, acc, shards, stream);
}
static int bn_bwd_reduce_pool_impl(const HpfgAct* g, const float* dP, int dp_pstride, int N, int Hp, int Wp, float* partials, long long* acc, int shards,
                                   void* stream) {
  HPFG_ARG_CHECK(g && dP && (partials || acc) && g->mode == HPFG_ACT_DZ && g->z && g->aux && g->bn, "bn_bwd_reduce_pool: needs a DZ source and dP");
  HPFG_ARG_CHECK(g->C % 4 == 0 && g->C >= 4 && g->C <= 1024 && 256 % (g->C / 4) == 0, "bn_bwd_reduce_pool: unsupported C=%d", g->C);
  HPFG_ARG_CHECK(g->Hs == 2 * Hp && g->Ws == 2 * Wp && N > 0, "bn_bwd_reduce_pool: the source must be exactly twice the pooled size");
  HPFG_ARG_CHECK(g->drop_p == 0.f, "bn_bwd_reduce_pool: a pooled block output has no dropout behind it");
  HPFG_ARG_CHECK(dp_pstride % 4 == 0 && g->aux_pstride % 4 == 0 && g->pstride % 4 == 0, "bn_bwd_reduce_pool: pixel strides must be multiples of 4");
  if (acc) HPFG_ACC_CHECK(hpfg_bn_bwd_pool_blocks(N, Hp, Wp, g->C), shards, "bn_bwd_reduce_pool_acc");
  hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, dim3(hpfg_bn_bwd_pool_blocks(N, Hp, Wp, g->C)), dim3(256), 0, (hipStream_t)stream, *g, dP, dp_pstride,
                     (long)N * Hp * Wp, Hp, Wp, partials, acc, shards);
  return hpfg_launch_status("bn_bwd_reduce_pool_kernel");
}

extern "C" int hpfg_bn_bwd_finalize(const float* partials, int nblk, const double* sums, double count, const float* gamma, float* bn,
                                    float* dgamma, float* dbeta, int C, float param_grad_scale, void* stream) {
  HPFG_ARG_CHECK((partials && nblk > 0) || sums, "bn_bwd_finalize: need partials or sums");
  HPFG_ARG_CHECK(gamma && bn && C > 0 && count > 0, "bn_bwd_finalize: bad args");
  HpfgPeerX none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, nblk, sums, count, gamma, bn, dgamma, dbeta, C,
                     param_grad_scale, none);
  return hpfg_launch_status("bn_bwd_finalize_kernel");
}

extern "C" int hpfg_bn_bwd_finalize_x(const float* partials, int nblk, const HpfgPeerX* px, double count, const float* gamma, float* bn, float* dgamma,
                                      float* dbeta, int C, float param_grad_scale, void* stream) {
  HPFG_ARG_CHECK(partials && nblk > 0 && gamma && bn && C > 0 && count > 0, "bn_bwd_finalize_x: bad args");
  if (int rc = check_px(px, C, "bn_bwd_finalize_x")) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, partials, nblk, nullptr, count, gamma, bn, dgamma, dbeta, C,
                     param_grad_scale, *px);
  return hpfg_launch_status("bn_bwd_finalize_kernel");
}
