// Weight gradient of the 3x3 / 1x1 convolutions on gfx950 matrix cores (exact fp32, v_mfma_f32_16x16x4_f32).
//
//   dW[co,ci,tap] = sum_{n,y,x} A[n, y+dy(tap), x+dx(tap), ci] * dZ[n,y,x,co]
//
// Replaces autograd's convolution_backward (weight) for nn.Conv2d at model/unet.py:18,22,50,99.  Both operands are virtual:
// A is re-activated from the producer's raw output exactly as in the forward pass, dZ is rebuilt from the upstream gradient
// and the saved raw output via the BatchNorm-backward coefficients (common.h, HPFG_ACT_DZ).
//
// GEMM view: M = input channel (16 per workgroup), N = output channel (16*NJ per workgroup), K = pixels.  A workgroup walks
// a strided list of (image, 16x16 pixel tile) work items, keeps the per-tap accumulators in registers, and finally writes a
// private slab; a second kernel sums the slabs in a fixed order (bitwise reproducible, no float atomics) and emits the
// gradient in PyTorch's [Cout][Cin][k][k] layout.  3x3: three waves, wave w owns kernel row ky = w.  1x1: wave w owns
// n-tile w.
#include <cstdlib>
#include "wgrad_kernel.h"
#include "wgrad_bf16_kernel.h"

namespace {

// dw[co][ci][tap] = sum_s slab[s][tap][ci][co].  Workgroup = 16 elements x 16 slab lanes; fixed summation order (reproducible).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int S, int taps, int Cin,
                                                          int CinPad, int Cout, int CoutPad) {
  __shared__ float red[16][17];
  const long per = (long)taps * CinPad * CoutPad;
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + il;
  float t0 = 0.f, t1 = 0.f;
  if (i < per) {
    int s = sl;
    for (; s + 16 < S; s += 32) {
      t0 += slab[s * per + i];
      t1 += slab[(s + 16) * per + i];
    }
    if (s < S) t0 += slab[s * per + i];
  }
  red[sl][il] = t0 + t1;
  __syncthreads();
  if (sl == 0 && i < per) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][il];
    int co = (int)(i % CoutPad);
    int ci = (int)((i / CoutPad) % CinPad);
    int tap = (int)(i / ((long)CoutPad * CinPad));
    if (co < Cout && ci < Cin) dw[((long)co * Cin + ci) * taps + tap] = t;
  }
}

// all layers of a backward pass in one launch: blockIdx.y = layer, workgroups stride over that layer's 16-element groups
// One launch for every layer (blockIdx.y): a workgroup sums 64 consecutive slab elements (16 threads x float4: 256 contiguous
// bytes per slab row) over the slabs in 16 interleaved lanes, in a fixed order (reproducible), and scatters them to OIHW.
__global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const HpfgSlabDesc* __restrict__ table) {
  __shared__ float red[16][65];
  const HpfgSlabDesc d = table[blockIdx.y];
  const long per = (long)d.taps * d.CinPad * d.CoutPad;
  const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const bool vec = (per & 3) == 0;
  for (long g0 = (long)blockIdx.x * 64; g0 < per; g0 += (long)gridDim.x * 64) {
    const long i = g0 + il * 4;
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
    if (vec && i + 3 < per) {
      int s = sl;
      f32x4 t2 = t0, t3 = t0;
      for (; s + 48 < d.S; s += 64) {                 // four independent rows in flight per thread (the chain is latency bound)
        t0 += *reinterpret_cast<const f32x4*>(d.slab + s * per + i);
        t1 += *reinterpret_cast<const f32x4*>(d.slab + (s + 16) * per + i);
        t2 += *reinterpret_cast<const f32x4*>(d.slab + (s + 32) * per + i);
        t3 += *reinterpret_cast<const f32x4*>(d.slab + (s + 48) * per + i);
      }
      for (; s < d.S; s += 16) t0 += *reinterpret_cast<const f32x4*>(d.slab + s * per + i);
      t0 += t2;
      t1 += t3;
    } else {
      for (int s = sl; s < d.S; s += 16)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (i + j < per) t0[j] += d.slab[s * per + i + j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[sl][il * 4 + j] = t0[j] + t1[j];
    __syncthreads();
    if (threadIdx.x < 64) {
      const long e = g0 + threadIdx.x;
      if (e < per) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
        const int co = (int)(e % d.CoutPad), ci = (int)((e / d.CoutPad) % d.CinPad), tap = (int)(e / ((long)d.CoutPad * d.CinPad));
        if (co < d.Cout && ci < d.Cin) d.dw_oihw[((long)co * d.Cin + ci) * d.taps + tap] = t;
      }
    }
  }
}

// db[c] = sum_p g[p*pstride + c]: two-stage deterministic reduction
__global__ __launch_bounds__(256) void channel_sum_stage1(const float* __restrict__ g, int pstride, long npix, int C, float* __restrict__ scratch) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  if ((C & 3) == 0 && (pstride & 3) == 0) {      // float4 per thread: quad q of pixel lane pl, 4 independent pixel loads in flight
    const int Q = C >> 2, q = tid % Q, pl = tid / Q, PL = 256 / Q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (pl < PL) {
      const long stride = (long)gridDim.x * PL;
      for (long pix = (long)blockIdx.x * PL + pl; pix < npix; pix += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long px = pix + u * stride;
          if (px < npix) a += *reinterpret_cast<const f32x4*>(g + px * pstride + q * 4);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[tid * 4 + j] = a[j];
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      const int qq = tid >> 2, j = tid & 3;
      for (int l = 0; l < PL; ++l) t += red[(l * Q + qq) * 4 + j];
      scratch[(long)blockIdx.x * C + tid] = t;
    }
    return;
  }
  const int c = tid % C, pl = tid / C, PL = 256 / C;
  float a = 0.f;
  if (pl < PL)
    for (long pix = (long)blockIdx.x * PL + pl; pix < npix; pix += (long)gridDim.x * PL) a += g[pix * pstride + c];
  red[tid] = a;
  __syncthreads();
  if (tid < C) {
    float t = 0.f;
    for (int l = 0; l < PL; ++l) t += red[l * C + tid];
    scratch[(long)blockIdx.x * C + tid] = t;
  }
}

__global__ __launch_bounds__(64) void channel_sum_stage2(const float* __restrict__ scratch, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) a += (double)scratch[(long)i * C + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
  if (threadIdx.x == 0) out[c] = (float)a;
}

}  // namespace

extern "C" int hpfg_wgrad_splits(int N, int H, int W, int CinPad, int CoutPad, int taps) {
  (void)taps;
  using namespace hpfg_wg;
  int ni, nj;                                // workgroup shape of the bf16x3 kernel (the f32 kernel takes whatever S it is given)
  hpfg_wg16::pick_shape(CinPad, CoutPad, &ni, &nj);
  long pairs = (long)(CinPad / (16 * ni)) * (CoutPad / (16 * nj));
  long nwork = (long)N * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  const long resident = 512;                // exactly one resident round of workgroups: 2 per CU by registers (3 per CU spills 8 .. 102
  long target = resident / pairs;           // VGPRs and costs the step +8 %, profiles/r04_bn_acc.txt)
  if (target < 1) target = 1;
  if (target > nwork) target = nwork;
  long per_block = (nwork + target - 1) / target;   // work items per workgroup
  return (int)((nwork + per_block - 1) / per_block);
}

extern "C" long hpfg_wgrad_slab_floats(int N, int H, int W, int CinPad, int CoutPad, int taps) {
  return (long)hpfg_wgrad_splits(N, H, W, CinPad, CoutPad, taps) * taps * CinPad * CoutPad;
}

extern "C" int hpfg_wgrad(const HpfgWgradArgs* a, void* stream) {
  HPFG_ARG_CHECK(a && a->slab && a->dw_oihw, "wgrad: null pointer");
  HPFG_ARG_CHECK(a->taps == 9 || a->taps == 1, "wgrad: taps must be 1 or 9");
  HPFG_ARG_CHECK(a->CinPad % 16 == 0 && a->CoutPad % 16 == 0 && a->Cin <= a->CinPad && a->Cout <= a->CoutPad, "wgrad: bad channel padding");
  HPFG_ARG_CHECK(a->S >= 1, "wgrad: S < 1");
  HPFG_ARG_CHECK(a->a1.mode == HPFG_ACT_NONE || a->a0.C % 16 == 0, "wgrad: concat needs a0.C %% 16 == 0");
  hipStream_t st = (hipStream_t)stream;
  const int akind = hpfg_kind_of(a->a0, a->a1);
  HPFG_ARG_CHECK(akind >= 0 && akind != HPFG_KIND_DZ, "wgrad: unsupported input source (a0.mode=%d, a1.mode=%d)", a->a0.mode, a->a1.mode);
  int rc = 1;
  const bool b16 = (a->math & 0xff) == HPFG_MATH_BF16X3 && a->taps == 9;
  if ((a->math & 0xff) == HPFG_MATH_BF16X3 && a->taps == 1 && (a->g.mode == HPFG_ACT_PLAIN || a->g.mode == HPFG_ACT_STRIDED) && a->g.C % 4 == 0)
    rc = hpfg_wgrad16_launch_1x1(*a, akind, st);          // 1 = kind not covered, use the fp32 kernel below
  if (a->g.mode == HPFG_ACT_SPLIT16 || akind == HPFG_KIND_SPLIT) {
    HPFG_ARG_CHECK(b16, "wgrad: SPLIT16 sources are a feature of the 3x3 bf16x3 kernel");
    rc = hpfg_wgrad16_launch_split(*a, akind, st);
  } else if (rc != 1) {
  } else if (b16 && a->g.mode == HPFG_ACT_DZ) rc = hpfg_wgrad16_launch_dz(*a, akind, st);
  else if (b16 && (a->g.mode == HPFG_ACT_PLAIN || a->g.mode == HPFG_ACT_STRIDED)) rc = hpfg_wgrad16_launch_plain(*a, akind, st);
  else if (a->g.mode == HPFG_ACT_DZ) rc = hpfg_wgrad_launch_dz(*a, akind, st);
  else if (a->g.mode == HPFG_ACT_PLAIN || a->g.mode == HPFG_ACT_STRIDED) rc = hpfg_wgrad_launch_plain(*a, akind, st);
  else {
    hpfg_set_error("wgrad: unsupported gradient source mode %d", a->g.mode);
    return -1;
  }
  if (rc || a->defer_reduce) return rc;
  long per = (long)a->taps * a->CinPad * a->CoutPad;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((per + 15) / 16)), dim3(256), 0, st, a->slab, a->dw_oihw, a->S, a->taps, a->Cin, a->CinPad,
                     a->Cout, a->CoutPad);
  return hpfg_launch_status("slab_reduce_kernel");
}

extern "C" int hpfg_slab_reduce_multi(const HpfgSlabDesc* table_dev, const HpfgSlabDesc* table_host, int nlayers, void* stream) {
  HPFG_ARG_CHECK(table_dev && table_host && nlayers > 0 && nlayers < 65536, "slab_reduce_multi: bad args");
  long mx = 0;
  for (int i = 0; i < nlayers; ++i) {
    long per = (long)table_host[i].taps * table_host[i].CinPad * table_host[i].CoutPad;
    if (per > mx) mx = per;
  }
  long gx = (mx + 63) / 64;
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((unsigned)gx, nlayers), dim3(256), 0, (hipStream_t)stream, table_dev);
  return hpfg_launch_status("slab_reduce_multi_kernel");
}

extern "C" int hpfg_channel_sum_blocks(long npix, int C) {
  if (C < 1 || C > 256 || npix < 1) return 0;
  const int PL = 256 / C;
  const long want = (npix + PL - 1) / PL;
  return (int)(want < 512 ? want : 512);
}

extern "C" int hpfg_channel_sum_partials(const float* g, int pstride, long npix, int C, float* partials, void* stream) {
  HPFG_ARG_CHECK(g && partials && C >= 1 && C <= 256 && npix > 0, "channel_sum_partials: bad args");
  hipLaunchKernelGGL(channel_sum_stage1, dim3(hpfg_channel_sum_blocks(npix, C)), dim3(256), 0, (hipStream_t)stream, g, pstride, npix, C, partials);
  return hpfg_launch_status("channel_sum_stage1");
}

extern "C" int hpfg_channel_sum(const float* g, int pstride, long npix, int C, float* out, float* scratch, void* stream) {
  HPFG_ARG_CHECK(g && out && scratch && C >= 1 && C <= 256 && npix > 0, "channel_sum: bad args");
  int PL = 256 / C;
  long want = (npix + PL - 1) / PL;
  int nblk = (int)(want < 512 ? want : 512);   // scratch must hold 512*C floats
  hipLaunchKernelGGL(channel_sum_stage1, dim3(nblk), dim3(256), 0, (hipStream_t)stream, g, pstride, npix, C, scratch);
  hipLaunchKernelGGL(channel_sum_stage2, dim3(C), dim3(64), 0, (hipStream_t)stream, scratch, nblk, C, out);
  return hpfg_launch_status("channel_sum");
}
