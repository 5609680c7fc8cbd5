// Projection-neck pooling and the NT-Xent pieces of Dense_Loss (reference model/unet.py:120-152, utils/loss/dense_loss.py:17-40).
// The dense products in between run on hpfg_gemm_f32 (gemm.hip).  Everything here is a few hundred KB per step: the kernels are
// written for a small launch count and deterministic sums, not for bandwidth.
#include "common.h"

namespace {

// torch AdaptiveAvgPool2d bin of output index i over an extent of L: [floor(i*L/S), ceil((i+1)*L/S))
__device__ __forceinline__ void bin_range(int i, int L, int S, int& b0, int& b1) {
  b0 = (i * L) / S;
  b1 = ((i + 1) * L + S - 1) / S;
}

// x: NHWC [N,H,W,C] (pixel stride ps).  grid (S*S + 1, N): block b < S*S writes pool[n, b, :] (bin by = b / S, bx = b % S), block S*S
// writes gap[n, :] (unet.py:141-142, :146).  Threads own one channel each in 256 / C pixel slots; slots are summed in a fixed order.
__global__ __launch_bounds__(256) void neck_pool_fwd_kernel(const float* __restrict__ x, int ps, int H, int W, int C, int S, float* __restrict__ gap,
                                                            float* __restrict__ pool) {
  __shared__ float sh[256];
  const int b = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  int y0 = 0, y1 = H, x0 = 0, x1 = W;
  if (b < S * S) {
    bin_range(b / S, H, S, y0, y1);
    bin_range(b % S, W, S, x0, x1);
  }
  const int slots = 256 / C;            // C <= 256
  const int c = tid % C, slot = tid / C;
  const int bw = x1 - x0, npix = (y1 - y0) * bw;
  float s = 0.f;
  if (slot < slots) {          // four independent loads in flight per thread (a bin of the 224 x 224 logits is 49 pixels per slot: latency, not bytes)
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = slot;
    for (; i + 3 * slots < npix; i += 4 * slots) {
      const int i1 = i + slots, i2 = i + 2 * slots, i3 = i + 3 * slots;
      const float v0 = x[(((long)n * H + y0 + i / bw) * W + x0 + i % bw) * ps + c];
      const float v1 = x[(((long)n * H + y0 + i1 / bw) * W + x0 + i1 % bw) * ps + c];
      const float v2 = x[(((long)n * H + y0 + i2 / bw) * W + x0 + i2 % bw) * ps + c];
      const float v3 = x[(((long)n * H + y0 + i3 / bw) * W + x0 + i3 % bw) * ps + c];
      s += v0;
      s1 += v1;
      s2 += v2;
      s3 += v3;
    }
    for (; i < npix; i += slots) s += x[(((long)n * H + y0 + i / bw) * W + x0 + i % bw) * ps + c];
    s = (s + s1) + (s2 + s3);
  }
  sh[tid] = s;
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int k = 0; k < slots; ++k) t += sh[k * C + tid];
    t /= (float)npix;
    if (b < S * S) pool[((long)n * S * S + b) * C + tid] = t;
    else gap[(long)n * C + tid] = t;
  }
}

// bins that tile the image exactly (H % S == 0, W % S == 0): the global mean is the mean of the S*S bin means -- spares the one
// workgroup per image that would otherwise walk the whole image alone (224 x 224 logits: 50 us)
__global__ void gap_from_pool_kernel(const float* __restrict__ pool, int N, int C, int SS, float* __restrict__ gap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  float t = 0.f;
  for (int b = 0; b < SS; ++b) t += pool[((long)n * SS + b) * C + c];
  gap[i] = t / (float)SS;
}

// dx[n,y,x,c] = dgap[n,c] / (H*W) + sum over the bins containing (y,x) of dpool[n,bin,c] / binsize      (dx: NHWC contiguous)
__global__ __launch_bounds__(256) void neck_pool_bwd_kernel(const float* __restrict__ dgap, const float* __restrict__ dpool, int N, int H, int W, int C,
                                                            int S, float* __restrict__ dx) {
  const long total = (long)N * H * W * C;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % C);
  const long pix = e / C;
  const int xx = (int)(pix % W), yy = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
  float v = dgap ? dgap[(long)n * C + c] / (float)(H * W) : 0.f;
  if (dpool) {
    for (int by = 0; by < S; ++by) {
      int y0, y1;
      bin_range(by, H, S, y0, y1);
      if (yy < y0 || yy >= y1) continue;
      for (int bx = 0; bx < S; ++bx) {
        int x0, x1;
        bin_range(bx, W, S, x0, x1);
        if (xx < x0 || xx >= x1) continue;
        v += dpool[((long)n * S * S + by * S + bx) * C + c] / (float)((y1 - y0) * (x1 - x0));
      }
    }
  }
  dx[e] = v;
}

// F.normalize(x, dim=1) of x viewed as [G, D, S] (dense_loss.py:18-19): u = x / max(||x||_2 over D, 1e-12), norms [G, S].
// Element (g, d, s) sits at g*F + d*sd + s*ss (F = D*S; either [D][S] or [S][D] order inside a row); u keeps x's layout, so the Gram
// matrix of the rows is the one of the flattened features whichever order they are stored in.
// One WAVE per (g, s) column (a thread per column walked D elements one after the other: 28 us for 64 x 16 columns of 128).
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int G, int D, int S, int sd, int ss, float* __restrict__ u,
                                                         float* __restrict__ norms) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= G * S) return;
  const int g = i / S, s = i % S;
  const long base = (long)g * D * S + (long)s * ss;
  float q = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = x[base + (long)d * sd];
    q += v * v;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float nrm = fmaxf(sqrtf(q), 1e-12f);
  if (lane == 0) norms[i] = nrm;
  for (int d = lane; d < D; d += 64) u[base + (long)d * sd] = x[base + (long)d * sd] / nrm;
}

// dx = scale * (du - u * <u, du>) / norm   per (g, s) column  (the clamp is inactive for non-degenerate features); scale: device scalar or NULL
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ du, const float* __restrict__ u, const float* __restrict__ norms, int G,
                                                         int D, int S, int sd, int ss, const float* __restrict__ scale, float* __restrict__ dx) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= G * S) return;
  const int g = i / S, s = i % S;
  const long base = (long)g * D * S + (long)s * ss;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += u[base + (long)d * sd] * du[base + (long)d * sd];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o);
  const float inv = (scale ? scale[0] : 1.f) / norms[i];
  for (int d = lane; d < D; d += 64) dx[base + (long)d * sd] = (du[base + (long)d * sd] - u[base + (long)d * sd] * dot) * inv;
}

// NT-Xent over the Gram matrix Gm [2n, 2n] of U = [student ; teacher] rows (dense_loss.py:24-36):
//   sim = exp(Gm / T), denom_i = sum_{j != i} sim_ij, pos_i = sim_{i, p(i)}, p(i) = (i + n) mod 2n, loss = mean_i -log(pos_i / denom_i).
// Also writes  Q = dL/dGm + (dL/dGm)^T  for the STUDENT rows [0, n) (what the gradient of the student features needs:
// dL/du_i = sum_j Q_ij u_j), dL/dGm_ij = (1 / (2n T)) * ([j != i] sim_ij / denom_i - [j == p(i)]).
// One workgroup of 1024 threads, 2n <= 256: four threads share a row (one thread per row spent 32 us in 128 serial expf's).
__global__ __launch_bounds__(1024) void ntxent_rows_kernel(const float* __restrict__ Gm, int n, float inv_t, float* __restrict__ loss, float* __restrict__ Q) {
  __shared__ float denom[256];
  __shared__ float lrow[256];
  const int m = 2 * n, i = threadIdx.x >> 2, part = threadIdx.x & 3;
  float dn = 0.f;
  if (i < m) {
    for (int j = part; j < m; j += 4)
      if (j != i) dn += expf(Gm[(long)i * m + j] * inv_t);
  }
  dn += __shfl_xor(dn, 1);      // (the partial sums of a row are combined in a fixed order: deterministic)
  dn += __shfl_xor(dn, 2);
  if (i < m && part == 0) {
    denom[i] = dn;
    const int p = i < n ? i + n : i - n;
    lrow[i] = logf(dn) - Gm[(long)i * m + p] * inv_t;          // -log(pos / denom)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < m; ++r) s += lrow[r];
    loss[0] = s / (float)m;
  }
  if (Q && i < n) {
    const float k = inv_t / (float)m;
    for (int j = part; j < m; j += 4) {
      float q = 0.f;
      if (j != i) {
        const float sim = expf(Gm[(long)i * m + j] * inv_t);     // Gm is symmetric up to rounding; both orientations are read as stored
        const float simt = expf(Gm[(long)j * m + i] * inv_t);
        q = k * (sim / denom[i] + simt / denom[j]);
        if (j == i + n) q -= 2.f * k;                             // p(i) = j and p(j) = i
      }
      Q[(long)i * m + j] = q;
    }
  }
}

}  // namespace

extern "C" int hpfg_neck_pool_fwd(const float* x, int pstride, int N, int H, int W, int C, int S, float* gap, float* pool, void* stream) {
  HPFG_ARG_CHECK(x && gap && pool && N > 0 && H > 0 && W > 0 && C >= 1 && C <= 256 && S >= 1 && S <= H && S <= W && pstride >= C,
                 "neck_pool_fwd: bad args (C=%d S=%d)", C, S);
  const bool tiles = H % S == 0 && W % S == 0;
  hipLaunchKernelGGL(neck_pool_fwd_kernel, dim3(S * S + (tiles ? 0 : 1), N), dim3(256), 0, (hipStream_t)stream, x, pstride, H, W, C, S, gap, pool);
  if (tiles) hipLaunchKernelGGL(gap_from_pool_kernel, dim3((N * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, pool, N, C, S * S, gap);
  return hpfg_launch_status("neck_pool_fwd_kernel");
}

extern "C" int hpfg_neck_pool_bwd(const float* dgap, const float* dpool, int N, int H, int W, int C, int S, float* dx, void* stream) {
  HPFG_ARG_CHECK((dgap || dpool) && dx && N > 0 && H > 0 && W > 0 && C >= 1 && S >= 1, "neck_pool_bwd: bad args");
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(neck_pool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dgap, dpool, N, H, W, C, S, dx);
  return hpfg_launch_status("neck_pool_bwd_kernel");
}

extern "C" int hpfg_l2norm_fwd(const float* x, int G, int D, int S, int sd, int ss, float* u, float* norms, void* stream) {
  HPFG_ARG_CHECK(x && u && norms && G > 0 && D > 0 && S > 0 && ((sd == S && ss == 1) || (sd == 1 && ss == D)), "l2norm_fwd: bad args");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((G * S + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, G, D, S, sd, ss, u, norms);
  return hpfg_launch_status("l2norm_fwd_kernel");
}

extern "C" int hpfg_l2norm_bwd(const float* du, const float* u, const float* norms, int G, int D, int S, int sd, int ss, const float* scale, float* dx,
                               void* stream) {
  HPFG_ARG_CHECK(du && u && norms && dx && G > 0 && D > 0 && S > 0 && ((sd == S && ss == 1) || (sd == 1 && ss == D)), "l2norm_bwd: bad args");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((G * S + 3) / 4), dim3(256), 0, (hipStream_t)stream, du, u, norms, G, D, S, sd, ss, scale, dx);
  return hpfg_launch_status("l2norm_bwd_kernel");
}

extern "C" int hpfg_ntxent_rows(const float* gram, int n, float temperature, float* loss, float* Q, void* stream) {
  HPFG_ARG_CHECK(gram && loss && n >= 1 && 2 * n <= 256 && temperature > 0.f, "ntxent_rows: bad args (n=%d)", n);
  hipLaunchKernelGGL(ntxent_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, gram, n, 1.f / temperature, loss, Q);
  return hpfg_launch_status("ntxent_rows_kernel");
}
