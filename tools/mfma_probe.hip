// Cycles per v_mfma_f32_16x16x32_bf16 in loops shaped like the conv k-loop (diagnostics; GPU only):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// One wave per SIMD (256 threads per workgroup, one workgroup per CU, 256 workgroups).  Variants:
//   0  one accumulator, in place                         1  four accumulators, round robin
//   2  three dependent MFMAs per accumulator, 4 accumulators (the conv kernel's order: hh, lh, hl of tile m, then tile m+1)
//   5  a dependent chain that ping-pongs between two registers (D != C in every MFMA: what the register allocator emits under pressure)
//   6  three dependent MFMAs per tile as in 2, but the first of each triple writes a different register than it reads (D != C once per triple)
//   3  as 2 + two ds_read_b128 per 3 MFMAs into a ring (fragments actually used)        4  as 3 + two global_load_dwordx4 per 12 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void probe(float* out, const bf16x8* wsrc, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (float)(i & 255);
  __syncthreads();
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.01f * (float)(lane + j)); b[j] = (__bf16)(0.02f * (float)(lane - j)); }
  f32x4 acc[4];
  for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* base = lds + (lane & 15) * 16 + (lane >> 4) * 3584;
  const bf16x8* wp = wsrc + lane;
  bf16x8 fr[10];
  for (int k = 0; k < 10; ++k) fr[k] = a;
  bf16x8 wr[2] = {b, b};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[0], 0, 0, 0);
    } else if (V == 5) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n\ts_nop 1" : "=&v"(acc[1]) : "v"(acc[0]), "v"(b), "v"(a));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n\ts_nop 1" : "=&v"(acc[0]) : "v"(acc[1]), "v"(b), "v"(a));
      }
    } else if (V == 6) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        f32x4 t;
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n\ts_nop 1" : "=&v"(t) : "v"(acc[m]), "v"(b), "v"(a));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %1\n\ts_nop 1" : "=&v"(acc[m]) : "v"(t), "v"(b), "v"(a));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+v"(acc[m]) : "v"(b), "v"(a));
      }
    } else if (V == 7) {
#pragma unroll
      for (int k = 0; k < 12; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+v"(acc[0]) : "v"(b), "v"(a));
    } else if (V == 1) {
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[k & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (V >= 3) {
          fr[(2 * m + 8) % 10] = *reinterpret_cast<const bf16x8*>(base + ((it * 4 + m) & 7) * 288);
          fr[(2 * m + 9) % 10] = *reinterpret_cast<const bf16x8*>(base + ((it * 4 + m) & 7) * 288 + 1792);
        }
        if (V >= 4 && m == 0) {
          wr[0] = wp[((it & 63) * 2) * 64];
          wr[1] = wp[((it & 63) * 2 + 1) * 64];
        }
        const bf16x8 ah = V >= 3 ? fr[(2 * m) % 10] : a, al = V >= 3 ? fr[(2 * m + 1) % 10] : a;
        const bf16x8 bh = V >= 4 ? wr[0] : b, bl = V >= 4 ? wr[1] : b;
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x216);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(float* out, bf16x8* w, unsigned long long* cyc, int grid) {
  const int iters = 2000;
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, w, cyc, iters);
  hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 0, 0, out, w, cyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("variant %d, %d workgroups: %.1f cycles per MFMA (s_memtime ticks)\n", V, grid, s / grid / iters / 12.0);
}

int main() {
  float* out; bf16x8* w; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&w, 64 * 2 * 64 * 16 * 2); hipMalloc(&cyc, 1024 * 8);
  hipMemset(w, 0, 64 * 2 * 64 * 16 * 2);
  for (int grid : {256, 512}) {
    run<0>(out, w, cyc, grid); run<1>(out, w, cyc, grid); run<2>(out, w, cyc, grid); run<3>(out, w, cyc, grid); run<4>(out, w, cyc, grid); run<7>(out, w, cyc, grid); run<5>(out, w, cyc, grid); run<6>(out, w, cyc, grid);
  }
  return 0;
}
