#include <hip/hip_runtime.h>
#include <cstdio>
// each lane copies 16 B from a (permuted) global granule into LDS through the DMA path, then LDS is dumped
__global__ void probe(const float* __restrict__ src, float* __restrict__ out, int perm) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  const int g = (lane * perm) % 64;                       // arbitrary per-lane global granule
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + g * 4),
                                   (void __attribute__((address_space(3)))*)(lds + 64 * 4), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
  float *s, *o; float h[256], r[512];
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  hipMalloc(&s, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(o, 0, sizeof(r));
  probe<<<1, 64>>>(s, o, 7);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (r[256 + l * 4 + j] != (float)(((l * 7) % 64) * 4 + j)) ok = 0;
  printf("dma probe %s; lds[256..263] = %g %g %g %g %g %g %g %g\n", ok ? "OK" : "MISMATCH", r[256], r[257], r[258], r[259], r[260], r[261], r[262], r[263]);
  return 0;
}
