// Small kernels around the conv pipeline: weight packing, virtual-activation materialisation, dropout-mask dump,
// max-pool / bilinear-upsample backward routing, CutMix blend, arg-max pseudo-labels, SGD and EMA over flat buffers.
#include <stdarg.h>
#include <string.h>
#include "stage.h"

static thread_local char g_err[512] = "";
extern "C" void hpfg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* hpfg_last_error(void) { return g_err; }
extern "C" int hpfg_version(void) { return HPFG_VERSION; }

// ---- kernel-form switches (tests and A/B tools; never read from the environment in a launch path) -------------------------------------
static int g_opt[HPFG_OPT_COUNT] = {1, 1, 1, 128};
int hpfg_opt(int which) { return which >= 0 && which < HPFG_OPT_COUNT ? g_opt[which] : 0; }
extern "C" int hpfg_set_option(int which, int value) {
  HPFG_ARG_CHECK(which >= 0 && which < HPFG_OPT_COUNT, "set_option: unknown option %d", which);
  const int old = g_opt[which];
  g_opt[which] = value;
  return old;
}

namespace {

// ---- weight packing: OIHW -> MFMA B-fragment order (see conv.hip) ------------------------------------------------------
// fwd  : wpk[tap][ch][nt][lane][ks] = W[co = nt*16 + (lane&15)][ci = ch*16 + ks*4 + (lane>>4)][tap]
// dgrad: wpk[tap][ch][nt][lane][ks] = W[co = ch*16 + ks*4 + (lane>>4)][ci = nt*16 + (lane&15)][taps-1-tap]
// Two per-forward device counters ride along (one workgroup, before any packing): num_batches_tracked of the network's BatchNorm layers
// (+1 each) and the dropout seed word of the engine (+seed_add) -- no launches of their own in front of every forward.
__global__ __launch_bounds__(256) void pack_weights_kernel(const HpfgPackDesc* __restrict__ table, long long* __restrict__ counters, int n_counters,
                                                           int* __restrict__ seed_word, int seed_add, long long* __restrict__ zero, long n_zero) {
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    for (int i = threadIdx.x; i < n_counters; i += 256) counters[i] += 1;
    if (threadIdx.x == 0 && seed_word && seed_add) *seed_word = (*seed_word + seed_add) & 0x7FFFFFFF;
  }
  // the BatchNorm sum accumulators of the pass that follows (HpfgConvArgs.stat_acc) start from zero
  for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < n_zero; i += (long)gridDim.x * gridDim.y * 256) zero[i] = 0;
  const HpfgPackDesc d = table[blockIdx.y];
  const long total = (long)d.taps * d.CinPad * d.CoutPad;
  const int nt_f = d.CoutPad / 16, nch_f = d.CinPad / 16;
  const int nt_d = d.CinPad / 16, nch_d = d.CoutPad / 16;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int ks = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    if (d.wpk_fwd) {
      int nt = (int)(r % nt_f), ch = (int)((r / nt_f) % nch_f), tap = (int)(r / ((long)nt_f * nch_f));
      int co = nt * 16 + (lane & 15), ci = ch * 16 + ks * 4 + (lane >> 4);
      d.wpk_fwd[i] = (co < d.Cout && ci < d.Cin) ? d.w_oihw[((long)co * d.Cin + ci) * d.taps + tap] : 0.f;
    }
    if (d.wpk_dgrad) {
      int nt = (int)(r % nt_d), ch = (int)((r / nt_d) % nch_d), tap = (int)(r / ((long)nt_d * nch_d));
      int co = ch * 16 + ks * 4 + (lane >> 4), ci = nt * 16 + (lane & 15);
      d.wpk_dgrad[i] = (co < d.Cout && ci < d.Cin) ? d.w_oihw[((long)co * d.Cin + ci) * d.taps + (d.taps - 1 - tap)] : 0.f;
    }
    if (i < d.CoutPad) d.bias_pad[i] = (i < d.Cout && d.b) ? d.b[i] : 0.f;
  }
  // ---- bf16x3 fragments: [ks][ntile][hi|lo][64 lanes][8], k-local = 8*(lane>>4)+j, column = lane&15 (conv_bf16_kernel.h) ----
  const int ksteps = d.taps == 1 ? 1 : (d.kc == 32 ? 9 : 5);
  for (int dir = 0; dir < 2; ++dir) {
    __bf16* out = reinterpret_cast<__bf16*>(dir == 0 ? d.wpk16_fwd : d.wpk16_dgrad);
    if (!out) continue;
    const int Kch = dir == 0 ? d.Cin : d.Cout;                  // contraction channels
    const int ntn = (dir == 0 ? d.CoutPad : d.CinPad) / 16;     // output-channel tiles
    const int nchunks = (Kch + d.kc - 1) / d.kc;
    const long tot = (long)nchunks * ksteps * ntn * 512;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
      const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
      const long r = i >> 9;
      const int nt = (int)(r % ntn), ks = (int)(r / ntn);
      const int chunk = ks / ksteps, s = ks % ksteps;
      const int kl = 8 * (lane >> 4) + j;
      int tap, kch;
      if (d.taps == 1) { tap = 0; kch = chunk * 32 + kl; }
      else if (d.kc == 32) { tap = s; kch = chunk * 32 + kl; }
      else { tap = 2 * s + (kl >> 4); kch = chunk * 16 + (kl & 15); }
      const int nch = nt * 16 + (lane & 15);
      float w = 0.f;
      if (tap < d.taps && kch < Kch) {
        if (dir == 0) { if (nch < d.Cout) w = d.w_oihw[((long)nch * d.Cin + kch) * d.taps + tap]; }
        else { if (nch < d.Cin) w = d.w_oihw[((long)kch * d.Cin + nch) * d.taps + (d.taps - 1 - tap)]; }
      }
      const __bf16 hi = (__bf16)w;
      const __bf16 lo = (__bf16)(w - (float)hi);
      const long base = ((long)ks * ntn + nt) * 2;
      out[(base + 0) * 512 + lane * 8 + j] = hi;
      out[(base + 1) * 512 + lane * 8 + j] = lo;
    }
  }
}

__global__ __launch_bounds__(256) void materialize_kernel(HpfgAct a0, HpfgAct a1, int N, int H, int W, float* __restrict__ out) {
  const int Ct = a0.C + a1.C, Q = (Ct + 3) / 4;
  const ActCtx c0 = make_ctx(a0), c1 = make_ctx(a1);
  const long total = (long)N * H * W * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int q = (int)(i % Q);
    long pix = i / Q;
    int x = (int)(pix % W), y = (int)((pix / W) % H), n = (int)(pix / ((long)W * H));
    f32x4 v = cat_load4(a0, c0, a1, c1, n, y, x, q * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < Ct) out[pix * Ct + q * 4 + j] = v[j];
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out, long n, uint32_t thresh, uint32_t seed0,
                                                           const uint32_t* __restrict__ seed_dev) {
  const uint32_t seed = seed0 + (seed_dev ? *seed_dev : 0u);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = hpfg_keep((uint32_t)i, seed, thresh) ? 1 : 0;
}

// MaxPool2d(2) backward, fused with the skip-gradient accumulation: dA[n, 2yp+dy*, 2xp+dx*, c] += dP[n,yp,xp,c]
__global__ __launch_bounds__(256) void pool_scatter_kernel(HpfgAct s, const float* __restrict__ dP, int dp_ps, float* __restrict__ dA, int da_ps,
                                                           int N, int Hp, int Wp) {
  const int Q = s.C / 4;
  const long total = (long)N * Hp * Wp * Q;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int q = (int)(i % Q);
    long pp = i / Q;
    int xp = (int)(pp % Wp), yp = (int)((pp / Wp) % Hp), n = (int)(pp / ((long)Wp * Hp));
    int c = q * 4;
    f32x4 sc = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SCALE * s.bn_stride + s.bn_coff + c);
    f32x4 sh = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SHIFT * s.bn_stride + s.bn_coff + c);
    long p00 = (long)(n * s.Hs + 2 * yp) * s.Ws + 2 * xp;
    long pos[4] = {p00, p00 + 1, p00 + s.Ws, p00 + s.Ws + 1};
    f32x4 zz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) zz[k] = *reinterpret_cast<const f32x4*>(s.z + pos[k] * s.pstride + c);
    f32x4 g = *reinterpret_cast<const f32x4*>(dP + pp * dp_ps + c);
    // whole-window read-modify-write in float4s (coalesced); a scalar update of only the argmax element touches the same sectors
    f32x4 da[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) da[k] = *reinterpret_cast<const f32x4*>(dA + pos[k] * da_ps + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float best = lrelu(zz[0][j] * sc[j] + sh[j]);
      int bi = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        float v = lrelu(zz[k][j] * sc[j] + sh[j]);
        if (v > best) {
          best = v;
          bi = k;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) da[k][j] += bi == k ? g[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(dA + pos[k] * da_ps + c) = da[k];
  }
}

// transpose of the align_corners=True bilinear x2: gather form, one thread per (low-res pixel, channel quad); taps: hpfg_up_taps (common.h)
// The tap lists depend only on the low-res row / column: every workgroup builds both tables in LDS once (one table entry per
// thread) instead of every thread re-deriving them for its pixel (that was ~170 VALU instructions per output float4).
// (A tiled separable variant -- high-res patch staged once in LDS, vertical then horizontal pass -- was measured 1.4-1.8x SLOWER:
// two 72 KB workgroups per CU keep too few loads in flight; this gather form runs 8 workgroups per CU and its re-reads hit L2.)
// Optionally the per-channel sums of each workgroup's outputs go to csum[blockIdx][C]: the bias gradient of the 1x1 conv that
// produced the low-res tensor (model/unet.py:50) is sum_p dU, and the slab reduction adds the rows in a fixed order.
constexpr int UPB_MAXDIM = 512, UPB_TAPS = HPFG_UPB_TAPS;      // a x2 align-corners map sends at most 5 (on average 4) outputs per axis to one source
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void upsample_bwd_kernel(const float* __restrict__ dUp, int dup_ps, float* __restrict__ dU, int N, int Hl, int Wl,
                                                           int C, float* __restrict__ csum, int xcd_aware) {
  __shared__ short t_idx[UPB_MAXDIM][UPB_TAPS];
  __shared__ float t_w[UPB_MAXDIM][UPB_TAPS];
  __shared__ float red[256 * 4];
  // every row / column gets exactly UPB_TAPS entries: unused ones repeat the last valid index with weight 0, so that the gather below is
  // 25 unconditional loads -- ALL in flight before the first is used.  (With run-time tap counts the loops issued load, wait, multiply 16
  // times in a row: the kernel was bound by sixteen L2 round trips per thread, 2 TB/s on a stream that fits L2.)
  for (int e = threadIdx.x; e < Hl + Wl; e += 256) {
    int idx[8], cnt;
    float wgt[8];
    if (e < Hl) hpfg_up_taps(e, Hl, idx, wgt, cnt);
    else hpfg_up_taps(e - Hl, Wl, idx, wgt, cnt);
    if (cnt > UPB_TAPS) cnt = UPB_TAPS;      // cannot happen for a x2 align-corners map
    for (int k = 0; k < UPB_TAPS; ++k) {
      t_idx[e][k] = (short)(k < cnt ? idx[k] : idx[cnt - 1]);
      t_w[e][k] = k < cnt ? wgt[k] : 0.f;
    }
  }
  __syncthreads();
  const int Q = C / 4, Ho = 2 * Hl, Wo = 2 * Wl;
  const long total = (long)N * Hl * Wl * Q;
  f32x4 csum4 = {0.f, 0.f, 0.f, 0.f};
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Every XCD owns a CONTIGUOUS eighth of the
  // outputs and its workgroups sweep it side by side (grid-stride inside the eighth), so the high-res rows that neighbouring low-res rows
  // both gather from are in flight on one XCD at one time and are fetched once (linear grid-stride over all XCDs fetched 2x the
  // algorithmic bytes; PMC: 199 -> 137 MB per step for a contiguous run per workgroup).
  const int nxcd = (xcd_aware && gridDim.x % 8 == 0) ? 8 : 1;
  const long gx = gridDim.x / nxcd, xj = blockIdx.x / nxcd;
  const long span = ((total + nxcd - 1) / nxcd + 255) / 256 * 256;      // outputs per XCD, whole 256-thread rows
  const long x0 = (long)(blockIdx.x % nxcd) * span;
  const long x1 = x0 + span < total ? x0 + span : total;
  for (long i = x0 + xj * 256 + threadIdx.x; i < x1; i += gx * 256) {      // 256 % Q == 0: a thread keeps its channel quad
    int q = (int)(i % Q);
    long pp = i / Q;
    int xl = (int)(pp % Wl), yl = (int)((pp / Wl) % Hl), n = (int)(pp / ((long)Wl * Hl));
    int iy[UPB_TAPS], ix[UPB_TAPS];
    float wy[UPB_TAPS], wx[UPB_TAPS];
#pragma unroll
    for (int k = 0; k < UPB_TAPS; ++k) {
      iy[k] = t_idx[yl][k];
      wy[k] = hpfg_own_vgpr(t_w[yl][k]);              // (broadcast factors of f32x4 products: common.h, HPFG_NO_PK_F32)
      ix[k] = t_idx[Hl + xl][k] * dup_ps;
      wx[k] = hpfg_own_vgpr(t_w[Hl + xl][k]);
    }
    f32x4 v[UPB_TAPS][UPB_TAPS];
#pragma unroll
    for (int a = 0; a < UPB_TAPS; ++a) {
      const float* row = dUp + ((long)(n * Ho + iy[a]) * Wo) * dup_ps + q * 4;
#pragma unroll
      for (int b = 0; b < UPB_TAPS; ++b) v[a][b] = *reinterpret_cast<const f32x4*>(row + ix[b]);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < UPB_TAPS; ++a) {          // separable: the horizontal taps of a row first, then the row's vertical weight
      f32x4 h = wx[0] * v[a][0];
#pragma unroll
      for (int b = 1; b < UPB_TAPS; ++b) h += wx[b] * v[a][b];
      acc += wy[a] * h;
    }
    *reinterpret_cast<f32x4*>(dU + pp * C + q * 4) = acc;
    csum4 += acc;
  }
  if (csum) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x * 4 + j] = csum4[j];
    __syncthreads();
    if ((int)threadIdx.x < C) {
      const int q = threadIdx.x >> 2, j = threadIdx.x & 3;
      float s = 0.f;
      for (int k = q; k < 256; k += Q) s += red[k * 4 + j];           // fixed order
      csum[(long)blockIdx.x * C + threadIdx.x] = s;
    }
  }
}

__global__ __launch_bounds__(256) void cutmix_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ m,
                                                     float* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a[i] * (1.f - m[i]) + b[i] * m[i];
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, long npix, int C, const uint8_t* __restrict__ mix_labels,
                                                     const float* __restrict__ mix_mask, uint8_t* __restrict__ out) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const float* l = logits + i * C;
    float best = l[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
      if (l[c] > best) {
        best = l[c];
        bi = c;
      }
    if (mix_mask) bi = mix_mask[i] > 0.5f ? bi : (int)mix_labels[i];
    out[i] = (uint8_t)bi;
  }
}

// Training-time augmentation of ACDC slices (datasets/utils.py:73-117 RandomGenerator): optional rot90+flip or integer-angle
// rotation (scipy.ndimage.rotate, order 0, reshape=False), then nearest resize to the network size (scipy.ndimage.zoom, order 0),
// for image and mask alike.  All three are index remaps, so one gather per output pixel does the whole chain: the zoom's
// per-axis source indices come from the host (scipy's own 1-D tables), the rotation follows scipy's affine arithmetic in
// double precision (matrix and offset computed on the host exactly as ndimage.rotate does), rot90 / flip are exact permutations.
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ img_pool, const uint8_t* __restrict__ lab_pool,
                                                      const HpfgAugSample* __restrict__ samples, const int* __restrict__ tabs, int B, int H, int W,
                                                      float* __restrict__ out_img, uint8_t* __restrict__ out_lab) {
  const long total = (long)B * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int X = (int)(i % W), Y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    const HpfgAugSample s = samples[b];
    const int y1 = tabs[s.tab_off + Y], x1 = tabs[s.tab_off + H + X];      // pixel of the (possibly rotated) intermediate image
    int ys = y1, xs = x1;
    bool inside = y1 >= 0 && x1 >= 0;            // -1: scipy's zoom wrote its constant there
    if (!inside) {
    } else if (s.mode == 1) {
      const bool odd = s.k & 1;
      const int h1 = odd ? s.w : s.h, w1 = odd ? s.h : s.w;                  // shape after rot90
      int bi = y1, bj = x1;                                                  // undo the flip
      if (s.axis == 0) bi = h1 - 1 - y1;
      else bj = w1 - 1 - x1;
      switch (s.k & 3) {                                                     // np.rot90(a, k)[bi, bj]
        case 0: ys = bi; xs = bj; break;
        case 1: ys = bj; xs = s.w - 1 - bi; break;
        case 2: ys = s.h - 1 - bi; xs = s.w - 1 - bj; break;
        default: ys = s.h - 1 - bj; xs = bi; break;
      }
    } else if (s.mode == 2) {
      // icoor = ((0 + y*m00) + x*m01) + shift, no contraction: the operation order of scipy's NI_GeometricTransform
      double cy = __dmul_rn((double)y1, s.m00);
      cy = __dadd_rn(cy, __dmul_rn((double)x1, s.m01));
      cy = __dadd_rn(cy, s.off_y);
      double cx = __dmul_rn((double)y1, s.m10);
      cx = __dadd_rn(cx, __dmul_rn((double)x1, s.m11));
      cx = __dadd_rn(cx, s.off_x);
      inside = !(cy < 0.0 || cy > (double)(s.h - 1) || cx < 0.0 || cx > (double)(s.w - 1));      // mode='constant', cval 0
      ys = (int)floor(cy + 0.5);
      xs = (int)floor(cx + 0.5);
    }
    float v = 0.f;
    uint8_t l = 0;
    if (inside) {
      v = img_pool[s.img_off + (long)ys * s.w + xs];
      l = lab_pool[s.lab_off + (long)ys * s.w + xs];
    }
    out_img[i] = v;
    out_lab[i] = l;
  }
}

// ICT (2022_02_ISBI_ICT-MedSeg_ACDC.py:111-129): per-sample mixes of inputs and of teacher probabilities
__global__ __launch_bounds__(256) void mix_samples_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ f,
                                                          float* __restrict__ out, long total, long per) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float w = f[i / per];
    out[i] = a[i] * (1.0f - w) + b[i] * w;
  }
}

__global__ __launch_bounds__(256) void softmax_mix_kernel(const float* __restrict__ t0, const float* __restrict__ t1, const float* __restrict__ f,
                                                          float* __restrict__ out, long npix, long pix_per_sample, int C) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const float w = f[i / pix_per_sample];
    const float* x0 = t0 + i * C;
    const float* x1 = t1 + i * C;
    float m0 = x0[0], m1 = x1[0];
    for (int c = 1; c < C; ++c) {
      m0 = fmaxf(m0, x0[c]);
      m1 = fmaxf(m1, x1[c]);
    }
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < C; ++c) {
      s0 += expf(x0[c] - m0);
      s1 += expf(x1[c] - m1);
    }
    for (int c = 0; c < C; ++c) out[i * C + c] = expf(x0[c] - m0) / s0 * (1.0f - w) + expf(x1[c] - m1) / s1 * w;
  }
}

// diagnostics: the 100 MHz chip-wide clock, written when the stream reaches this point (works inside a captured graph)
__global__ void timestamp_kernel(unsigned long long* slot) {
  if (threadIdx.x == 0) *slot = __builtin_amdgcn_s_memrealtime();
}

// UAMT (2019_07_MICCAI_Uncertainty_Aware_ACDC.py:130-147): input noise for the teacher passes, and the uncertainty mask
// out[i] = x[i % n_src] + clamp(noise[i] * scale, lo, hi)   (unlabeled.repeat(2,1,1,1) + clamp(randn*0.1, -0.2, 0.2), :130,142)
__global__ __launch_bounds__(256) void noise_add_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                                        long n_src, long n_out, float scale, float lo, float hi) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n_out; i += (long)gridDim.x * 256) {
    const float v = fminf(fmaxf(noise[i] * scale, lo), hi);
    out[i] = x[i % n_src] + v;
  }
}

// T stochastic teacher predictions of the same S images (image g = t*S + s lives in block g / per_block): mean over t of the
// softmax, entropy  u = -sum_c p log(p + 1e-6)  (:147-151), mask = u < *threshold (:162-163)
__global__ __launch_bounds__(256) void uncertainty_mask_kernel(HpfgPredBlocks pb, int T, int S, long npix_img, int C,
                                                               const float* __restrict__ threshold, float* __restrict__ mask,
                                                               float* __restrict__ uncertainty) {
  const float thr = *threshold;
  const long total = (long)S * npix_img;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int s = (int)(i / npix_img);
    const long px = i - (long)s * npix_img;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int t = 0; t < T; ++t) {
      const int g = t * S + s;
      const float* l = pb.p[g / pb.per_block] + ((long)(g % pb.per_block) * npix_img + px) * C;
      float v[8], m = l[0];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c < C ? l[c] : 0.f;
#pragma unroll
      for (int c = 1; c < 8; ++c) m = c < C ? fmaxf(m, v[c]) : m;
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v[c] = c < C ? expf(v[c] - m) : 0.f;
        se += v[c];
      }
      const float inv = 1.f / se;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] += v[c] * inv;
    }
    float u = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float pm = acc[c] / (float)T;
      u += c < C ? pm * logf(pm + 1e-6f) : 0.f;
    }
    u = -u;
    mask[i] = u < thr ? 1.f : 0.f;
    if (uncertainty) uncertainty[i] = u;
  }
}

// CutMix box masks (utils/utils.py:165-173): mask = invert ? 0 : 1, flipped once per box that covers the pixel
__global__ __launch_bounds__(256) void box_masks_kernel(const int* __restrict__ rects, int n, int nb, int H, int W, int invert,
                                                        float* __restrict__ out) {
  const long total = (long)n * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)((i / W) % H), m = (int)(i / ((long)W * H));
    int flips = 0;
    for (int b = 0; b < nb; ++b) {
      const int* r = rects + ((long)m * nb + b) * 4;      // y0, y1, x0, x1 (already normalised like a Python slice)
      flips += (y >= r[0] && y < r[1] && x >= r[2] && x < r[3]) ? 1 : 0;
    }
    out[i] = (float)((invert ? 0 : 1) ^ (flips & 1));
  }
}

// Evaluation (val.py:376-387): class-confusion counts of a predicted and a true label volume, counts[gt * C + pred] (integer
// atomics: exact and order independent).  A workgroup first counts in LDS.
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt, long n, int C,
                                                        unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[16 * 16];
  for (int i = threadIdx.x; i < C * C; i += 256) h[i] = 0;
  __syncthreads();
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int p = pred[i], g = gt[i];
    if (p < C && g < C) atomicAdd(&h[g * C + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += 256)
    if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long n,
                                                  const float* __restrict__ lr_dev, float momentum, float wd, float gscale) {
#pragma clang fp contract(off)      // separately rounded products, as the reference's optimizer: the fused and the two-launch forms agree bit for bit
  const float lr = *lr_dev;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float w = p[i];
    float d = g[i] * gscale + wd * w;
    float b = momentum * mom[i] + d;
    mom[i] = b;
    p[i] = w - lr * b;
  }
}

// SGD step + EMA teacher update in one pass (the Mean-Teacher family ends every step with both: 2017_03_NIPS_Mean-Teacher_ACDC.py:108-113):
// the fresh weight goes into the teacher's average from a register instead of being read back by a second launch.  Same expressions as
// sgd_kernel / ema_kernel, so the results are bit-identical to the two launches.
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long n,
                                                      const float* __restrict__ lr_dev, float momentum, float wd, float gscale,
                                                      float* __restrict__ t, long n_ema, const float* __restrict__ alpha_dev) {
#pragma clang fp contract(off)      // separately rounded products, as the reference's optimizer: the fused and the two-launch forms agree bit for bit
  const float lr = *lr_dev, a = *alpha_dev;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float w = p[i];
    float d = g[i] * gscale + wd * w;
    float b = momentum * mom[i] + d;
    mom[i] = b;
    const float s = w - lr * b;
    p[i] = s;
    if (i < n_ema) t[i] = t[i] * a + s * (1.f - a);
  }
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected moments) over flat buffers; *step_dev holds the number of steps already taken
// (a one-thread kernel advances it afterwards, so a captured graph counts by itself)
__global__ __launch_bounds__(256) HPFG_NO_PK_F32 void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long n, const float* __restrict__ lr_dev, const float* __restrict__ step_dev, float b1, float b2,
                                                    float eps, float wd, float gscale) {
  const float lr = *lr_dev, t = *step_dev + 1.f;
  const float bc1 = 1.f - powf(b1, t), bc2s = sqrtf(1.f - powf(b2, t));
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float w = p[i] * decay;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - step_size * mi / (sqrtf(vi) / bc2s + eps);
  }
}

__global__ void step_inc_kernel(float* step_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *step_dev += 1.f;
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long n, const float* __restrict__ alpha_dev) {
#pragma clang fp contract(off)      // separately rounded products, as the reference's optimizer: the fused and the two-launch forms agree bit for bit
  const float a = *alpha_dev;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) t[i] = t[i] * a + s[i] * (1.f - a);
}

inline int grid_for(long total, int cap = 4096) {
  long b = (total + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" int hpfg_conv_kc(int H, int W, int taps) { return (taps == 9 && H % 16 == 0 && W % 16 == 0) ? 16 : 32; }

extern "C" long hpfg_wpk16_elems(int Kchannels, int NchannelsPad, int taps, int kc) {
  const int ksteps = taps == 1 ? 1 : (kc == 32 ? 9 : 5);
  return (long)((Kchannels + kc - 1) / kc) * ksteps * (NchannelsPad / 16) * 2 * 64 * 8;
}

extern "C" int hpfg_pack_weights_bump(const HpfgPackDesc* table_dev, const HpfgPackDesc* table_host, int nlayers, long long* counters, int n_counters,
                                      int32_t* seed_word, int seed_add, long long* zero_words, long n_zero, void* stream) {
  HPFG_ARG_CHECK(table_dev && table_host && nlayers > 0 && nlayers < 65536, "pack_weights: bad args");
  HPFG_ARG_CHECK(n_zero >= 0 && (n_zero == 0 || zero_words), "pack_weights: bad zero region");
  HPFG_ARG_CHECK(n_counters >= 0 && (n_counters == 0 || counters) && (seed_add == 0 || seed_word), "pack_weights: bad counter arguments");
  long mx = 0;
  for (int i = 0; i < nlayers; ++i) {
    const HpfgPackDesc& d = table_host[i];
    HPFG_ARG_CHECK(d.CinPad % 16 == 0 && d.CoutPad % 16 == 0 && d.Cin <= d.CinPad && d.Cout <= d.CoutPad && (d.taps == 1 || d.taps == 9),
                   "pack_weights: bad descriptor %d", i);
    HPFG_ARG_CHECK((!d.wpk16_fwd && !d.wpk16_dgrad) || d.kc == 32 || (d.kc == 16 && d.taps == 9), "pack_weights: bad kc in descriptor %d", i);
    long t = (long)d.taps * d.CinPad * d.CoutPad;
    if (t > mx) mx = t;
  }
  hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(mx, 256), nlayers), dim3(256), 0, (hipStream_t)stream, table_dev, counters, n_counters, seed_word,
                     seed_add, zero_words, n_zero);
  return hpfg_launch_status("pack_weights_kernel");
}

extern "C" int hpfg_pack_weights(const HpfgPackDesc* table_dev, const HpfgPackDesc* table_host, int nlayers, void* stream) {
  return hpfg_pack_weights_bump(table_dev, table_host, nlayers, nullptr, 0, nullptr, 0, nullptr, 0, stream);
}

extern "C" int hpfg_act_materialize(const HpfgAct* a0, const HpfgAct* a1, int N, int H, int W, float* out, void* stream) {
  HPFG_ARG_CHECK(a0 && out && N > 0 && H > 0 && W > 0, "act_materialize: bad args");
  HpfgAct none;
  memset(&none, 0, sizeof(none));
  const HpfgAct& b = a1 ? *a1 : none;
  HPFG_ARG_CHECK(b.mode == HPFG_ACT_NONE || a0->C % 4 == 0, "act_materialize: concat needs a0.C %% 4 == 0");
  long total = (long)N * H * W * ((a0->C + b.C + 3) / 4);
  hipLaunchKernelGGL(materialize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *a0, b, N, H, W, out);
  return hpfg_launch_status("materialize_kernel");
}

extern "C" int hpfg_dropout_mask(uint8_t* out, long n_elems, float p, uint32_t seed, const uint32_t* seed_dev, void* stream) {
  HPFG_ARG_CHECK(out && n_elems > 0 && n_elems < (1L << 32) && p >= 0.f && p < 1.f, "dropout_mask: bad args");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n_elems)), dim3(256), 0, (hipStream_t)stream, out, n_elems, hpfg_drop_threshold(p), seed,
                     seed_dev);
  return hpfg_launch_status("dropout_mask_kernel");
}

extern "C" int hpfg_pool_scatter_add(const HpfgAct* src, const float* dP, int dp_pstride, float* dA, int da_pstride, int N, int Hp, int Wp,
                                     void* stream) {
  HPFG_ARG_CHECK(src && dP && dA && src->mode == HPFG_ACT_BNACT && src->C % 4 == 0, "pool_scatter_add: needs a BNACT source with C%%4==0");
  HPFG_ARG_CHECK(src->Hs == 2 * Hp && src->Ws == 2 * Wp, "pool_scatter_add: source must be 2x the pooled size");
  long total = (long)N * Hp * Wp * (src->C / 4);
  hipLaunchKernelGGL(pool_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *src, dP, dp_pstride, dA, da_pstride, N, Hp, Wp);
  return hpfg_launch_status("pool_scatter_kernel");
}

extern "C" int hpfg_upsample2x_bwd_blocks(int N, int Hl, int Wl, int C) {
  if (N < 1 || Hl < 1 || Wl < 1 || C < 4) return 0;
  return grid_for((long)N * Hl * Wl * (C / 4), 1024);      // = rows of channel sums the slab reduction has to add: keep that chain short
}

extern "C" int hpfg_upsample2x_bwd_sums(const float* dUp, int dup_pstride, float* dU, int N, int Hl, int Wl, int C, float* csum_partials,
                                        void* stream) {
  HPFG_ARG_CHECK(dUp && dU && C % 4 == 0 && C >= 4 && N > 0 && Hl > 0 && Wl > 0 && Hl + Wl <= UPB_MAXDIM, "upsample2x_bwd: bad args");
  HPFG_ARG_CHECK(!csum_partials || (C <= 256 && 256 % (C / 4) == 0), "upsample2x_bwd: channel sums need C/4 to divide 256 (C=%d)", C);
  const int xcd_aware = 1;      // (0: contiguous runs dealt to the XCDs round-robin -- an A/B of round 3, DESIGN_HISTORY.md)
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(hpfg_upsample2x_bwd_blocks(N, Hl, Wl, C)), dim3(256), 0, (hipStream_t)stream, dUp, dup_pstride, dU, N,
                     Hl, Wl, C, csum_partials, xcd_aware);
  return hpfg_launch_status("upsample_bwd_kernel");
}

extern "C" int hpfg_upsample2x_bwd(const float* dUp, int dup_pstride, float* dU, int N, int Hl, int Wl, int C, void* stream) {
  return hpfg_upsample2x_bwd_sums(dUp, dup_pstride, dU, N, Hl, Wl, C, nullptr, stream);
}

extern "C" int hpfg_cutmix_blend(const float* a, const float* b, const float* mask, float* out, long n, void* stream) {
  HPFG_ARG_CHECK(a && b && mask && out && n > 0, "cutmix_blend: bad args");
  hipLaunchKernelGGL(cutmix_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, mask, out, n);
  return hpfg_launch_status("cutmix_kernel");
}

extern "C" int hpfg_argmax_labels(const float* logits, int N, int H, int W, int C, const uint8_t* mix_labels, const float* mix_mask, uint8_t* out,
                                  void* stream) {
  HPFG_ARG_CHECK(logits && out && C >= 1 && C <= 255, "argmax_labels: bad args");
  HPFG_ARG_CHECK((mix_labels == nullptr) == (mix_mask == nullptr), "argmax_labels: mix_labels and mix_mask go together");
  long npix = (long)N * H * W;
  hipLaunchKernelGGL(argmax_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, logits, npix, C, mix_labels, mix_mask, out);
  return hpfg_launch_status("argmax_kernel");
}

extern "C" int hpfg_augment_batch(const float* img_pool, const uint8_t* lab_pool, const HpfgAugSample* samples_dev, const int* tabs_dev, int B, int H,
                                  int W, float* out_img, uint8_t* out_lab, void* stream) {
  HPFG_ARG_CHECK(img_pool && lab_pool && samples_dev && tabs_dev && out_img && out_lab && B > 0 && H > 0 && W > 0, "augment_batch: bad args");
  hipLaunchKernelGGL(augment_kernel, dim3(grid_for((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, img_pool, lab_pool, samples_dev, tabs_dev, B,
                     H, W, out_img, out_lab);
  return hpfg_launch_status("augment_kernel");
}

extern "C" int hpfg_mix_samples(const float* a, const float* b, const float* f, float* out, int n, long per_sample, void* stream) {
  HPFG_ARG_CHECK(a && b && f && out && n > 0 && per_sample > 0, "mix_samples: bad args");
  hipLaunchKernelGGL(mix_samples_kernel, dim3(grid_for((long)n * per_sample)), dim3(256), 0, (hipStream_t)stream, a, b, f, out, (long)n * per_sample,
                     per_sample);
  return hpfg_launch_status("mix_samples_kernel");
}

extern "C" int hpfg_softmax_mix(const float* t0, const float* t1, const float* f, float* out_prob, int n, int H, int W, int C, void* stream) {
  HPFG_ARG_CHECK(t0 && t1 && f && out_prob && n > 0 && H > 0 && W > 0 && C >= 1 && C <= 64, "softmax_mix: bad args");
  hipLaunchKernelGGL(softmax_mix_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, (hipStream_t)stream, t0, t1, f, out_prob, (long)n * H * W,
                     (long)H * W, C);
  return hpfg_launch_status("softmax_mix_kernel");
}

extern "C" int hpfg_timestamp(unsigned long long* slot, void* stream) {
  HPFG_ARG_CHECK(slot, "timestamp: null slot");
  hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
  return hpfg_launch_status("timestamp_kernel");
}

extern "C" int hpfg_noise_add(const float* x, const float* noise, float* out, long n_src, long n_out, float scale, float lo, float hi,
                              void* stream) {
  HPFG_ARG_CHECK(x && noise && out && n_src > 0 && n_out > 0 && lo <= hi, "noise_add: bad args");
  hipLaunchKernelGGL(noise_add_kernel, dim3(grid_for(n_out)), dim3(256), 0, (hipStream_t)stream, x, noise, out, n_src, n_out, scale, lo, hi);
  return hpfg_launch_status("noise_add_kernel");
}

extern "C" int hpfg_uncertainty_mask(const HpfgPredBlocks* pb, int T, int S, int H, int W, int C, const float* threshold_dev, float* mask,
                                     float* uncertainty, void* stream) {
  HPFG_ARG_CHECK(pb && threshold_dev && mask && T > 0 && S > 0 && H > 0 && W > 0 && C >= 1 && C <= 8, "uncertainty_mask: bad args (C <= 8)");
  HPFG_ARG_CHECK(pb->n_blocks >= 1 && pb->n_blocks <= 8 && pb->per_block > 0 && (long)pb->n_blocks * pb->per_block == (long)T * S,
                 "uncertainty_mask: %d blocks of %d images do not hold T*S = %d*%d predictions", pb->n_blocks, pb->per_block, T, S);
  for (int i = 0; i < pb->n_blocks; ++i) HPFG_ARG_CHECK(pb->p[i], "uncertainty_mask: block %d is NULL", i);
  hipLaunchKernelGGL(uncertainty_mask_kernel, dim3(grid_for((long)S * H * W)), dim3(256), 0, (hipStream_t)stream, *pb, T, S, (long)H * W, C,
                     threshold_dev, mask, uncertainty);
  return hpfg_launch_status("uncertainty_mask_kernel");
}

extern "C" int hpfg_box_masks(const int* rects, int n, int n_boxes, int H, int W, int invert, float* out, void* stream) {
  HPFG_ARG_CHECK(rects && out && n > 0 && n_boxes > 0 && H > 0 && W > 0, "box_masks: bad args");
  hipLaunchKernelGGL(box_masks_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, (hipStream_t)stream, rects, n, n_boxes, H, W, invert, out);
  return hpfg_launch_status("box_masks_kernel");
}

extern "C" int hpfg_confusion_counts(const uint8_t* pred, const uint8_t* gt, long n, int C, unsigned long long* counts, void* stream) {
  HPFG_ARG_CHECK(pred && gt && counts && n > 0 && C >= 1 && C <= 16, "confusion_counts: bad args (C <= 16)");
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, pred, gt, n, C, counts);
  return hpfg_launch_status("confusion_kernel");
}

extern "C" int hpfg_sgd_step(float* p, const float* g, float* mom, long n, const float* lr_dev, float momentum, float weight_decay, float grad_scale,
                             void* stream) {
  HPFG_ARG_CHECK(p && g && mom && lr_dev && n > 0, "sgd_step: bad args");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, mom, n, lr_dev, momentum, weight_decay, grad_scale);
  return hpfg_launch_status("sgd_kernel");
}

extern "C" int hpfg_sgd_ema_step(float* p, const float* g, float* mom, long n, const float* lr_dev, float momentum, float weight_decay, float grad_scale,
                                 float* t, long n_ema, const float* alpha_dev, void* stream) {
  HPFG_ARG_CHECK(p && g && mom && lr_dev && t && alpha_dev && n > 0 && n_ema >= 0 && n_ema <= n && t != p, "sgd_ema_step: bad args");
  hipLaunchKernelGGL(sgd_ema_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, mom, n, lr_dev, momentum, weight_decay, grad_scale,
                     t, n_ema, alpha_dev);
  return hpfg_launch_status("sgd_ema_kernel");
}

extern "C" int hpfg_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float* step_dev, float beta1, float beta2,
                               float eps, float weight_decay, float grad_scale, void* stream) {
  HPFG_ARG_CHECK(p && g && m && v && lr_dev && step_dev && n > 0, "adamw_step: bad args");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_dev, step_dev, beta1, beta2, eps,
                     weight_decay, grad_scale);
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev);
  return hpfg_launch_status("adamw_kernel");
}

extern "C" int hpfg_ema_update(float* t, const float* s, long n, const float* alpha_dev, void* stream) {
  HPFG_ARG_CHECK(t && s && alpha_dev && n > 0, "ema_update: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, t, s, n, alpha_dev);
  return hpfg_launch_status("ema_kernel");
}
