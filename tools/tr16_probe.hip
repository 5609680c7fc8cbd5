// Probe of ds_read_b64_tr_b16 semantics (diagnostics): prints what each lane receives from a [row][16 col] image.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int lane = threadIdx.x;
  int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (g * 4 + q) * 16 + 4 * p));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  int* d;
  hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int j = 0; j < 4; ++j) {
      int g = lane >> 4, i = lane & 15;
      int expect = (g * 4 + j) * 16 + i;   // column i of row j of the group's 4x16 block
      if (h[lane * 4 + j] != expect) ++bad;
    }
  printf("tr16 probe: %d mismatches; lane0=%d %d %d %d lane1=%d %d %d %d lane17=%d %d %d %d\n", bad, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
         h[68], h[69], h[70], h[71]);
  return 0;
}
