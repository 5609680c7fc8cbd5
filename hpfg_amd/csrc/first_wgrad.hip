// Weight gradient of the FIRST conv (network input -> 16 channels, model/unet.py:72 / :18: nn.Conv2d(in_channels, 16, 3, padding=1)), the last
// kernel on the chain of every backward pass (nothing is back-propagated into the network input, so this layer has no dgrad):
//
//   dW[co][ci][tap] = sum_{n,y,x} X[n, ci, y+dy, x+dx] * dZ[n,y,x,co],      dZ = k1*g + k2*z + k3,  g = dA * dropmask/(1-p) * lrelu'(scale*z+shift)
//
// K = 9 taps x 1 input channel (grey-scale slices: ACDC, LIDC patches are RGB and keep the tile kernel) against 16 output channels is no work for the matrix cores; the cost is READING (dA, z) of the largest
// activation of the network (2 x 16 channels x 4 B per pixel: 103 MB at 16 x 224^2) once.  The fused dgrad + wgrad kernel in its
// weight-gradient-only form (fused_bwd_kernel<.., NODG>) staged 18 x 18 halo tiles of dZ through LDS and the transposing MFMA reads for that
// and ran at 1.96 TB/s (54 us: 2.5 % of the Mean-Teacher step, all of it exposed).  This kernel streams instead: a workgroup walks image rows;
// consecutive threads own consecutive 16-byte channel quads of (dA, z) -- every wave instruction reads 1 KB contiguous of each, four such
// pairs in flight per thread -- derive dZ in registers, and correlate it with the 9 (27) input taps of their pixel by exact fp32 FMAs (the
// three input rows a work item touches sit in LDS, zero padded).  Per-thread sums -> wave butterfly over the 16 lanes of a quad -> LDS over the waves -> one slab per
// workgroup, summed in a fixed order by hpfg_slab_reduce_multi like every other weight gradient (reproducible: no float atomics).
#include "common.h"

namespace {

constexpr int NT = 256, XR_W = 514;

template <int CIN>
__global__ __launch_bounds__(NT) void first_wgrad_kernel(HpfgAct g, HpfgAct x, float* __restrict__ slab, int N, int H, int W) {
  constexpr int KT = 9 * CIN;
  __shared__ float tab[5][16];                       // scale, shift, k1, k2, k3 of the 16 output channels (table rows, or derived from the sums)
  __shared__ float red[NT / 64][KT][16];
  __shared__ float xrows[CIN * 3 * XR_W];            // the input rows y - 1, y, y + 1 of the current work item, zero padded: [CIN][3][W + 2], W <= XR_W - 2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hpfg_dz_rows_to_lds(g, &tab[0][0], 16, 16, tid, NT);
  __syncthreads();
  const int q = tid & 3, c0 = 4 * q;                 // this thread's channel quad: the same for every element it visits (the stride is a multiple of 4)
  f32x4 sc, sh, k1, k2, k3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = tab[0][c0 + j];
    sh[j] = tab[1][c0 + j];
    k1[j] = tab[2][c0 + j];
    k2[j] = tab[3][c0 + j];
    k3[j] = tab[4][c0 + j];
  }
  const ActCtx cx = make_ctx(g);
  const uint32_t thr = drop_thresh(g, cx);
  f32x4 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Work item = one image row (n, y): no per-element index divisions, and the three input rows its taps touch are parked in LDS once
  // (zero padded), so a tap is an LDS read.  A row's (dA, z) quads are requested U at a time before the first is consumed.
  // (Measured inside the Mean-Teacher step, where the side stream's last weight gradients still run beside this kernel: tile kernel 54 us;
  // one element per trip with the taps from global memory 48; this form 44; bands of five rows per work item 53 -- more registers, two
  // resident workgroups per CU instead of three.)
  constexpr int U = 4;
  const int rows = N * H, WQ = W * 4, WP = W + 2;
  float* xr = xrows;                                  // [CIN][3][W + 2]
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / H, y = r - n * H;
    __syncthreads();                                  // the previous row's tap reads are done
    for (int e = tid; e < CIN * 3 * WP; e += NT) {
      const int ci = e / (3 * WP), rr = (e - ci * 3 * WP) / WP, xq = e - ci * 3 * WP - rr * WP - 1, yy = y + rr - 1;
      const bool in = yy >= 0 && yy < H && xq >= 0 && xq < W;
      xr[e] = in ? x.z[(long)n * x.sn + (long)ci * x.sc + (long)yy * x.sy + (long)xq * x.sx] : 0.f;
    }
    __syncthreads();
    const long rowpix = (long)r * W;
    for (int b0 = 0; b0 < WQ; b0 += U * NT) {
      f32x4 z[U], ga[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = b0 + u * NT + tid;
        const long pix = rowpix + ((idx < WQ ? idx : tid) >> 2);
        z[u] = *reinterpret_cast<const f32x4*>(g.z + pix * g.pstride + c0);
        ga[u] = *reinterpret_cast<const f32x4*>(g.aux + pix * g.aux_pstride + c0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = b0 + u * NT + tid;
        const bool live = idx < WQ;
        const int xx = (live ? idx : tid) >> 2;
        const long pix = rowpix + xx;
        f32x4 gq = ga[u];
        if (g.drop_p > 0.f) {
          const uint32_t e = (uint32_t)(pix * g.C + c0);
          gq = gq * cx.inv_keep;
          const uint32_t h = drop_word(g, cx, e);
          gq[0] = (h & 0xFFu) >= thr ? gq[0] : 0.f;
          gq[1] = ((h >> 8) & 0xFFu) >= thr ? gq[1] : 0.f;
          gq[2] = ((h >> 16) & 0xFFu) >= thr ? gq[2] : 0.f;
          gq[3] = (h >> 24) >= thr ? gq[3] : 0.f;
        }
        f32x4 dz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float yv = z[u][j] * sc[j] + sh[j];
          const float gg = yv > 0.f ? gq[j] : HPFG_LEAKY * gq[j];
          dz[j] = k1[j] * gg + (k2[j] * z[u][j] + k3[j]);      // the association of the staged loaders (stage.h finish_piece<DZ>)
          dz[j] = live ? dz[j] : 0.f;
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[ci * 9 + t] += hpfg_own_vgpr(xr[(ci * 3 + t / 3) * WP + xx + t % 3]) * dz;      // (common.h: HPFG_NO_PK_F32)
      }
    }
  }
  // lanes with the same (lane & 3) hold the same channel quad: butterfly over the other four lane bits
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[k][j];
#pragma unroll
      for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o);
      acc[k][j] = v;
    }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < KT; ++k) *reinterpret_cast<f32x4*>(&red[wave][k][4 * lane]) = acc[k];
  }
  __syncthreads();
  // slab[block][tap][ci (CinPad = 16)][co (16)]: rows ci >= CIN are never read for a result (hpfg_slab_reduce_multi writes ci < Cin only)
  float* out = slab + (long)blockIdx.x * 9 * 16 * 16;
  for (int e = tid; e < KT * 16; e += NT) {
    const int k = e >> 4, co = e & 15, ci = k / 9, t = k - 9 * ci;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += red[w][k][co];
    out[(t * 16 + ci) * 16 + co] = s;
  }
}

}  // namespace

// workgroups (= slabs) of the launch: one image row per work item, three resident workgroups per CU
int hpfg_first_wgrad_grid(int N, int H, int W) {
  (void)W;
  const long rows = (long)N * H;
  return (int)(rows < 768 ? rows : 768);
}

// g: the DZ source of the first conv's output (16 channels), x: the 1-channel network input (STRIDED); slab: [grid][9][16][16] floats
int hpfg_first_wgrad_launch(const HpfgAct& g, const HpfgAct& x, float* slab, int Cin, int N, int H, int W, hipStream_t st) {
  if (Cin != 1) return -2;
  const int grid = hpfg_first_wgrad_grid(N, H, W);
  if (W + 2 > XR_W) return -2;
  hipLaunchKernelGGL((first_wgrad_kernel<1>), dim3(grid), dim3(NT), 0, st, g, x, slab, N, H, W);
  return hpfg_launch_status("first_wgrad_kernel");
}
