// Bandwidth probe for the "register-free tile prefetch" plan: one or two workgroups per CU stream
// 20 KB input tiles into an LDS ring R slots deep with global_load_lds_dwordx4, touch them, and write 16 KB per tile --
// the traffic shape of a thin 16->16-channel conv layer.  Prints GB/s (read + write) for R = 1..4 and 1 / 2 workgroups per CU.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/dma_stream_probe.hip -o /tmp/dsp && /tmp/dsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int IN_GRAN = 1280;             // 16-B granules per input tile (20 KB); 5 DMA instructions per wave
constexpr int OUT_F4 = 1024;              // float4 per output tile (16 KB)
constexpr int DMA_PER_WAVE = IN_GRAN / 64 / 4;

__device__ __forceinline__ void dma16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g, (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

template <int R>
__global__ __launch_bounds__(256) void stream(const float* __restrict__ in, float* __restrict__ out, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // R slots of IN_GRAN * 4 floats
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int G = gridDim.x;
  auto issue = [&](int t, int slot) {
    const float* src = in + (long)t * IN_GRAN * 4;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int blk = i * 4 + wave;
      dma16(src + (blk * 64 + lane) * 4, lds + slot * IN_GRAN * 4 + blk * 256);
    }
  };
  int t = blockIdx.x, k = 0;
  for (int j = 0; j < R && t + j * G < ntiles; ++j) issue(t + j * G, j);
  float4 accum = {0.f, 0.f, 0.f, 0.f};
  for (; t < ntiles; t += G, ++k) {
    const int slot = k % R;
    // wait for the oldest tile: allow the (R-1) younger tiles' DMA ops (and nothing else) to stay in flight
    if (R == 1) __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | 0);
    else if (R == 2) __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | (DMA_PER_WAVE & 15));
    else if (R == 3) __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | ((2 * DMA_PER_WAVE) & 15));
    else __builtin_amdgcn_s_waitcnt(0x0070 | 0x0F00 | ((3 * DMA_PER_WAVE) & 15));
    __syncthreads();
    const float4* s = reinterpret_cast<const float4*>(lds + slot * IN_GRAN * 4);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = s[i * 256 + tid], b = s[(i * 256 + tid + 256) % IN_GRAN];
      v[i] = make_float4(a.x + b.x, a.y * 0.5f, a.z - b.z, a.w);
      accum.x += v[i].x;
    }
    __syncthreads();                       // slot consumed
    if (t + R * G < ntiles) issue(t + R * G, slot);
    float4* o = reinterpret_cast<float4*>(out) + (long)t * OUT_F4;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i * 256 + tid] = v[i];
  }
  if (accum.x == 12345.678f) out[0] = accum.x;
}

template <int R>
void run(const float* in, float* out, int ntiles, int wg_per_cu) {
  const int grid = 256 * wg_per_cu;
  const size_t lds = (size_t)R * IN_GRAN * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream<R>, dim3(grid), dim3(256), lds, 0, in, out, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)ntiles * (IN_GRAN * 16 + OUT_F4 * 16);
  printf("R=%d wg/cu=%d lds=%zu KB: %.1f us  %.0f GB/s\n", R, wg_per_cu, lds >> 10, ms * 1e3, bytes / ms / 1e6);
}

int main() {
  const int ntiles = 3136;                 // 16 images x 196 tiles: the 224^2 layer
  float *in, *out;
  hipMalloc(&in, (size_t)ntiles * IN_GRAN * 16);
  hipMalloc(&out, (size_t)ntiles * OUT_F4 * 16);
  hipMemset(in, 0, (size_t)ntiles * IN_GRAN * 16);
  for (int w = 1; w <= 3; ++w) {
    run<1>(in, out, ntiles, w);
    run<2>(in, out, ntiles, w);
    if (w <= 2) run<3>(in, out, ntiles, w);
    if (w == 1) run<4>(in, out, ntiles, w);
  }
  return 0;
}
