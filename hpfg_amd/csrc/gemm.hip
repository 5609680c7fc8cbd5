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

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
  __shared__ float As[TK * LDT], Bs[TK * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging map: consecutive threads walk the operand's unit-stride index (coalesced for either orientation)
  const bool a_k_fast = p.sak == 1, b_n_fast = p.sbn == 1;
  for (int k0 = 0; k0 < p.K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (a_k_fast) {
        k = tid & 15;
        m = (tid >> 4) + 16 * i;
      } else {
        m = tid & 63;
        k = (tid >> 6) + 4 * i;
      }
      const int gm = m0 + m, gk = k0 + k;
      As[k * LDT + m] = (gm < p.M && gk < p.K) ? p.A[(long)gm * p.sam + (long)gk * p.sak] : 0.f;
      int n, kb;
      if (b_n_fast) {
        n = tid & 63;
        kb = (tid >> 6) + 4 * i;
      } else {
        kb = tid & 15;
        n = (tid >> 4) + 16 * i;
      }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb * LDT + n] = (gn < p.N && gkb < p.K) ? p.B[(long)gkb * p.sbk + (long)gn * p.sbn] : 0.f;
    }
    __syncthreads();
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
    __syncthreads();
  }
  // D layout: row = (lane >> 4) * 4 + r, col = lane & 15
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn + j * 16 + (lane & 15);
      if (gn >= p.N) continue;
      const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + i * 16 + (lane >> 4) * 4 + r;
        if (gm >= p.M) continue;
        float v = acc[i][j][r] + bv;
        float* c = p.C + (long)gm * p.ldc + gn;
        if (p.accumulate) v += *c;
        if (p.relu) v = fmaxf(v, 0.f);
        *c = v;
      }
    }
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
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  return hpfg_launch_status("gemm_f32_kernel");
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
