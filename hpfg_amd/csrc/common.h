// Shared device helpers for the HPFG gfx950 kernels: virtual-activation loaders, dropout RNG, error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/hpfg_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define HPFG_LEAKY 0.01f

extern "C" void hpfg_set_error(const char* fmt, ...);
int hpfg_opt(int which);      // current value of a kernel-form switch (hpfg_set_option, misc.hip)
#define HPFG_ARG_CHECK(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      hpfg_set_error(__VA_ARGS__);           \
      return -1;                             \
    }                                        \
  } while (0)

static inline int hpfg_launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    hpfg_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// ---- counter-based dropout RNG ------------------------------------------------------------------------------
// One 32-bit hash serves four neighbouring elements: h = mix32((i >> 2) * 0x9E3779B1 + seed), r8 = byte (i & 3) of h,
// keep(i) = r8 >= round(p * 256); i = NHWC element index of the activated tensor.  The kept values are scaled by the nominal
// 1/(1-p) of nn.Dropout (the realised keep rate differs from 1-p by < 0.002: 8-bit threshold).
// Stateless, so forward consumers and backward loaders regenerate the same mask without storing it.
__host__ __device__ static inline uint32_t hpfg_hash32(uint32_t i, uint32_t seed) {
  uint32_t h = i * 0x9E3779B1u + seed;
  h ^= h >> 15;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ static inline uint32_t hpfg_drop_threshold(float p) {
  int t = (int)(p * 256.0f + 0.5f);
  return (uint32_t)(t < 0 ? 0 : (t > 255 ? 255 : t));
}
__host__ __device__ static inline bool hpfg_keep(uint32_t i, uint32_t seed, uint32_t thresh8) {
  const uint32_t h = hpfg_hash32(i >> 2, seed);
  return ((h >> (8 * (i & 3u))) & 0xFFu) >= thresh8;
}

__device__ static inline float lrelu(float y) { return fmaxf(y, HPFG_LEAKY * y); }   // == y > 0 ? y : 0.01*y

// ---- BatchNorm sums through integer atomics (HpfgConvArgs.stat_acc / HpfgAct.bn_acc) ----------------------------------------------------
// A producer workgroup adds its per-channel partial sums to the layer accumulator, long long [SHARDS][which][limb][C]; every consumer reads
// the 8 shards of its channels and derives the BatchNorm coefficients itself, so no finalize launch sits between the two kernels.  The
// split t = hi + lo * 2^-52 (hi = rint(t), lo = (t - hi) * 2^52: both exact for a float t above 2^-29; |lo| <= 2^51, so 2^11 partial sums add
// without overflow -- every launcher that takes an accumulator checks its workgroups per shard against that: HPFG_ACC_CHECK) loses nothing an fp64 sum would keep, and integer sums are associative: bit-reproducible totals without a fixed
// workgroup order (a float atomic would make BatchNorm, hence the run, non-deterministic).
// host-side guard of that bound, for every launcher that accepts a sum accumulator: `workgroups` partial sums spread over `shards` shards
#define HPFG_ACC_MAX_ADDS 2048
#define HPFG_ACC_CHECK(workgroups, shards, who)                                                                                     \
  HPFG_ARG_CHECK(((long)(workgroups) + (shards) - 1) / ((shards) > 0 ? (shards) : 1) <= HPFG_ACC_MAX_ADDS,                           \
                 "%s: %ld workgroups on %d accumulator shards exceed the %d exact adds per shard of hpfg_acc_add", who, (long)(workgroups), (int)(shards), \
                 HPFG_ACC_MAX_ADDS)
__device__ __forceinline__ void hpfg_acc_add(long long* acc, int C, int shard, int which, int c, float t) {
  const float r = rintf(t);
  const long long hi = (long long)r, lo = (long long)rintf((t - r) * 4503599627370496.f);
  long long* b = acc + ((long)((shard * 2 + which) * C + c)) * 2;
  __hip_atomic_fetch_add(b, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (agent scope: correct wherever the workgroup runs; the
  __hip_atomic_fetch_add(b + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  //  sharding only spreads the same-address contention)
}
// both sums of channel c: every 16-byte (hi, lo) load of the `shards` shards in flight before the first add.  `with` = a value formed from
// the caller's OTHER loads of this channel (gamma, beta, table rows): the empty asm below makes it an input of the point where the
// accumulator loads are issued... it must not wait for it -- so it is only named as a dependency of the FIRST ADD, which pins those loads
// in front of the accumulator words' consumption (one exposed memory round trip instead of two).
__device__ __forceinline__ void hpfg_acc_read2(const long long* __restrict__ acc, int C, int shards, int c, double& s1, double& s2, float with = 0.f) {
  typedef long long i64x2 __attribute__((ext_vector_type(2)));
  i64x2 v[HPFG_ACC_MAX_SHARDS][2];
#pragma unroll
  for (int s = 0; s < HPFG_ACC_MAX_SHARDS; ++s) {
    const int sc = s < shards ? s : 0;          // (a clamped reload instead of a branch; its value is dropped below)
    v[s][0] = *reinterpret_cast<const i64x2*>(acc + ((long)((sc * 2 + 0) * C + c)) * 2);
    v[s][1] = *reinterpret_cast<const i64x2*>(acc + ((long)((sc * 2 + 1) * C + c)) * 2);
  }
  i64x2 a = {0, 0}, b = {0, 0};
  asm volatile("" ::"v"(with), "v"(v[0][0]));      // `with` and the first accumulator word are both needed HERE: their loads were issued above
#pragma unroll
  for (int s = 0; s < HPFG_ACC_MAX_SHARDS; ++s) {
    if (s < shards) {
      a += v[s][0];
      b += v[s][1];
    }
  }
  s1 = (double)a[0] + (double)a[1] * (1.0 / 4503599627370496.0);
  s2 = (double)b[0] + (double)b[1] * (1.0 / 4503599627370496.0);
}
// The one definition of the forward coefficients that every consumer prologue and hpfg_bn_acc_finalize share (no contraction left to the
// compiler: the values a forward consumer used and the table rows the backward kernels read must be the same bits).
struct HpfgBnCoef {
  float mean, rstd, scale, shift;
  double var;
};
// Everything in fp64, as torch's CPU BatchNorm and hpfg_bn_fwd_finalize always did (E[z^2] - mean^2 cancels; on the 16- to 64-pixel test
// fixtures one ulp of rstd flips an arg-max pseudo-label two steps later) -- but without the ~150 dependent instructions of the fp64
// divide / sqrt / reciprocal sequences in every consumer prologue: 1 / count and 1 / sqrt(var + eps) start from the fp32 hardware
// approximations (1 ulp) and take two Newton steps in fp64 (relative error 1e-7 -> 1e-14 -> below the fp64 rounding).
__device__ __forceinline__ HpfgBnCoef hpfg_bn_coef(double s1, double s2, double count, float eps, float ga, float be) {
#pragma clang fp contract(off)
  double ic = (double)__builtin_amdgcn_rcpf((float)count);
  ic = ic * __builtin_fma(-count, ic, 2.0);
  ic = ic * __builtin_fma(-count, ic, 2.0);
  const double mean = s1 * ic;
  double var = __builtin_fma(-mean, mean, s2 * ic);
  var = var < 0.0 ? 0.0 : var;
  const double ve = var + (double)eps, hv = 0.5 * ve;
  double r = (double)__builtin_amdgcn_rsqf((float)ve);
  r = r * __builtin_fma(-hv * r, r, 1.5);      // Newton: r <- r (3 - ve r^2) / 2
  r = r * __builtin_fma(-hv * r, r, 1.5);
  const double gr = (double)ga * r;
  HpfgBnCoef q;
  q.mean = (float)mean;
  q.rstd = (float)r;
  q.scale = (float)gr;
  q.shift = (float)__builtin_fma(-mean, gr, (double)be);
  q.var = var;
  return q;
}
// Backward coefficients dz = k1 * g + k2 * z + k3 of one channel from the backward sums (sum g, sum g * xhat) -- the definition
// hpfg_bn_bwd_finalize and every dZ consumer's prologue share (HpfgAct.bn_acc on a DZ source = the layer's BACKWARD accumulator).
struct HpfgBnBwdCoef {
  float k1, k2, k3;
};
__device__ __forceinline__ HpfgBnBwdCoef hpfg_bn_bwd_coef(double sg, double sgx, double count, double mean, double rstd, double ga) {
#pragma clang fp contract(off)
  double ic = (double)__builtin_amdgcn_rcpf((float)count);
  ic = ic * __builtin_fma(-count, ic, 2.0);
  ic = ic * __builtin_fma(-count, ic, 2.0);
  const double m1 = sg * ic, m2 = sgx * ic, gr = ga * rstd;
  HpfgBnBwdCoef q;
  q.k1 = (float)gr;
  q.k2 = (float)(-(gr * rstd) * m2);
  q.k3 = (float)(gr * __builtin_fma(mean * rstd, m2, -m1));
  return q;
}
// rows scale, shift, k1, k2, k3 of the C channels of a DZ source -> t[r * cmax + c] (LDS; all threads call, the caller syncs): the table
// rows, or (bn_acc) k1 .. k3 derived here from the layer's backward sums -- no bn_bwd_finalize launch between the kernel that completed the
// sums and this one.  mean / rstd / scale / shift come from the table the forward pass left (hpfg_bn_acc_finalize / hpfg_bn_fwd_finalize).
__device__ __forceinline__ void hpfg_dz_rows_to_lds(const HpfgAct& a, float* t, int cmax, int nch, int tid, int nthr) {
  for (int c = tid; c < nch; c += nthr) {
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < a.C) {
      const float* tb = a.bn + a.bn_coff + c;
      const int st = a.bn_stride;
      v[0] = tb[HPFG_BN_SCALE * st];
      v[1] = tb[HPFG_BN_SHIFT * st];
      if (a.bn_acc) {
        // every load of this channel goes out before the first is consumed: ONE exposed round trip (the compiler otherwise requests
        // gamma / mean / rstd only after the accumulator words have arrived -- a second one, ~2 us per kernel launch)
        const float mean = tb[HPFG_BN_MEAN * st], rstd = tb[HPFG_BN_RSTD * st], ga = a.bn_gamma[a.bn_coff + c];
        double sg, sgx;
        hpfg_acc_read2(a.bn_acc, st, a.bn_shards, a.bn_coff + c, sg, sgx, mean + rstd + ga);
        const HpfgBnBwdCoef q = hpfg_bn_bwd_coef(sg, sgx, (double)a.bn_count, (double)mean, (double)rstd, (double)ga);
        v[2] = q.k1;
        v[3] = q.k2;
        v[4] = q.k3;
      } else {
        v[2] = tb[HPFG_BN_K1 * st];
        v[3] = tb[HPFG_BN_K2 * st];
        v[4] = tb[HPFG_BN_K3 * st];
      }
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) t[r * cmax + c] = v[r];
  }
}
// scale / shift of the C channels of a BatchNorm'd source -> t[0 .. C) and t[cmax .. cmax + C) (LDS; all NTHR threads call, the caller syncs)
__device__ __forceinline__ void hpfg_bn_rows_to_lds(const HpfgAct& a, float* t, int cmax, int tid, int nthr) {
  for (int c = tid; c < a.C; c += nthr) {
    float sc, sh;
    if (a.bn_acc) {
      const int cc = a.bn_coff + c;
      const float ga = a.bn_gamma[cc], be = a.bn_beta[cc];      // (requested WITH the accumulator words: one round trip, see hpfg_dz_rows_to_lds)
      double s1, s2;
      hpfg_acc_read2(a.bn_acc, a.bn_stride, a.bn_shards, cc, s1, s2, ga + be);
      const HpfgBnCoef q = hpfg_bn_coef(s1, s2, (double)a.bn_count, a.bn_eps, ga, be);
      sc = q.scale;
      sh = q.shift;
    } else {
      sc = a.bn[a.bn_coff + HPFG_BN_SCALE * a.bn_stride + c];
      sh = a.bn[a.bn_coff + HPFG_BN_SHIFT * a.bn_stride + c];
    }
    t[c] = sc;
    t[cmax + c] = sh;
  }
}

// Per-block cache of everything a loader needs that does not depend on the pixel.
struct ActCtx {
  uint32_t thresh;
  uint32_t seed;
  float inv_keep;
};
__device__ static inline ActCtx make_ctx(const HpfgAct& s) {
  ActCtx c;
  c.thresh = hpfg_drop_threshold(s.drop_p);
  c.inv_keep = s.drop_p > 0.f ? 1.f / (1.f - s.drop_p) : 1.f;
  c.seed = s.drop_seed + ((s.drop_p > 0.f && s.seed_dev) ? *s.seed_dev : 0u);
  return c;
}

__device__ static inline bool keep_elem(const HpfgAct& s, const ActCtx& cx, uint32_t e) {
  return s.drop_mask ? s.drop_mask[e] != 0 : hpfg_keep(e, cx.seed, cx.thresh);
}

// keep flags of the 4 elements e..e+3 (e % 4 == 0) as a 4-bit mask: one hash, or one 32-bit load of an explicit mask.
__device__ static inline uint32_t keep4(const HpfgAct& s, const ActCtx& cx, uint32_t e) {
  if (s.drop_mask) {
    const uint32_t m = *reinterpret_cast<const uint32_t*>(s.drop_mask + e);
    return ((m & 0xFFu) ? 1u : 0u) | ((m & 0xFF00u) ? 2u : 0u) | ((m & 0xFF0000u) ? 4u : 0u) | ((m & 0xFF000000u) ? 8u : 0u);
  }
  const uint32_t h = hpfg_hash32(e >> 2, cx.seed);
  return ((h & 0xFFu) >= cx.thresh ? 1u : 0u) | (((h >> 8) & 0xFFu) >= cx.thresh ? 2u : 0u) | (((h >> 16) & 0xFFu) >= cx.thresh ? 4u : 0u) |
         ((h >> 24) >= cx.thresh ? 8u : 0u);
}

// Dropout of the 4 elements e..e+3 (e % 4 == 0) in place: v[j] = keep_j ? v[j] / (1 - p) : 0.  One word carries the four
// 8-bit draws (one hash) or the four bytes of an explicit mask; per element that is one byte compare, one select, one multiply.
__device__ static inline uint32_t drop_word(const HpfgAct& s, const ActCtx& cx, uint32_t e) {
  if (s.drop_mask) return *reinterpret_cast<const uint32_t*>(s.drop_mask + e);
  return hpfg_hash32(e >> 2, cx.seed);
}
__device__ static inline void drop_apply4(f32x4& v, uint32_t h, uint32_t thr, float inv_keep) {
  v[0] *= (h & 0xFFu) >= thr ? inv_keep : 0.f;
  v[1] *= ((h >> 8) & 0xFFu) >= thr ? inv_keep : 0.f;
  v[2] *= ((h >> 16) & 0xFFu) >= thr ? inv_keep : 0.f;
  v[3] *= (h >> 24) >= thr ? inv_keep : 0.f;
}
__device__ static inline uint32_t drop_thresh(const HpfgAct& s, const ActCtx& cx) { return s.drop_mask ? 1u : cx.thresh; }

// 4 channels [c, c+4) of the virtual activation `s` at image n, virtual pixel (y, x); caller guarantees the pixel is
// inside the virtual image (H x W).  Channels >= s.C read as zero.
__device__ static inline f32x4 act_load4(const HpfgAct& s, const ActCtx& cx, int n, int y, int x, int c) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (c >= s.C) return v;
  switch (s.mode) {
    case HPFG_ACT_PLAIN: {
      const float* p = s.z + ((long)(n * s.Hs + y) * s.Ws + x) * s.pstride + c;
      v = *reinterpret_cast<const f32x4*>(p);
      break;
    }
    case HPFG_ACT_STRIDED: {
      const float* p = s.z + (long)n * s.sn + (long)y * s.sy + (long)x * s.sx;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < s.C) v[j] = p[(long)(c + j) * s.sc];
      break;
    }
    case HPFG_ACT_BNACT: {
      long pix = (long)(n * s.Hs + y) * s.Ws + x;
      f32x4 z = *reinterpret_cast<const f32x4*>(s.z + pix * s.pstride + c);
      f32x4 sc = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SCALE * s.bn_stride + s.bn_coff + c);
      f32x4 sh = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SHIFT * s.bn_stride + s.bn_coff + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = lrelu(z[j] * sc[j] + sh[j]);
      if (s.drop_p > 0.f) {
        const uint32_t km = keep4(s, cx, (uint32_t)(pix * s.C + c));
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (km >> j) & 1u ? v[j] * cx.inv_keep : 0.f;
      }
      break;
    }
    case HPFG_ACT_BNACT_POOL: {
      f32x4 sc = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SCALE * s.bn_stride + s.bn_coff + c);
      f32x4 sh = *reinterpret_cast<const f32x4*>(s.bn + HPFG_BN_SHIFT * s.bn_stride + s.bn_coff + c);
      const float* base = s.z + ((long)(n * s.Hs + 2 * y) * s.Ws + 2 * x) * s.pstride + c;
      f32x4 z00 = *reinterpret_cast<const f32x4*>(base);
      f32x4 z01 = *reinterpret_cast<const f32x4*>(base + s.pstride);
      f32x4 z10 = *reinterpret_cast<const f32x4*>(base + (long)s.Ws * s.pstride);
      f32x4 z11 = *reinterpret_cast<const f32x4*>(base + (long)s.Ws * s.pstride + s.pstride);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = lrelu(z00[j] * sc[j] + sh[j]), b = lrelu(z01[j] * sc[j] + sh[j]);
        float d = lrelu(z10[j] * sc[j] + sh[j]), e = lrelu(z11[j] * sc[j] + sh[j]);
        v[j] = fmaxf(fmaxf(a, b), fmaxf(d, e));
      }
      break;
    }
    case HPFG_ACT_UP2X: {
      // torch upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1)
      int Ho = 2 * s.Hs, Wo = 2 * s.Ws;
      float ry = Ho > 1 ? (float)(s.Hs - 1) / (float)(Ho - 1) : 0.f;
      float rx = Wo > 1 ? (float)(s.Ws - 1) / (float)(Wo - 1) : 0.f;
      float fy = ry * (float)y, fx = rx * (float)x;
      int y0 = (int)fy, x0 = (int)fx;
      int y1 = y0 + (y0 < s.Hs - 1 ? 1 : 0), x1 = x0 + (x0 < s.Ws - 1 ? 1 : 0);
      float wy1 = fy - (float)y0, wx1 = fx - (float)x0;
      float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
      const float* b = s.z + (long)n * s.Hs * s.Ws * s.pstride + c;
      f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((long)y0 * s.Ws + x0) * s.pstride);
      f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((long)y0 * s.Ws + x1) * s.pstride);
      f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((long)y1 * s.Ws + x0) * s.pstride);
      f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((long)y1 * s.Ws + x1) * s.pstride);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = wy0 * (wx0 * v00[j] + wx1 * v01[j]) + wy1 * (wx0 * v10[j] + wx1 * v11[j]);
      break;
    }
    case HPFG_ACT_DZ: {
      long pix = (long)(n * s.Hs + y) * s.Ws + x;
      f32x4 z = *reinterpret_cast<const f32x4*>(s.z + pix * s.pstride + c);
      f32x4 g = *reinterpret_cast<const f32x4*>(s.aux + pix * s.aux_pstride + c);
      const float* t = s.bn + s.bn_coff + c;
      f32x4 sc = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SCALE * s.bn_stride);
      f32x4 sh = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SHIFT * s.bn_stride);
      f32x4 k1 = *reinterpret_cast<const f32x4*>(t + HPFG_BN_K1 * s.bn_stride);
      f32x4 k2 = *reinterpret_cast<const f32x4*>(t + HPFG_BN_K2 * s.bn_stride);
      f32x4 k3 = *reinterpret_cast<const f32x4*>(t + HPFG_BN_K3 * s.bn_stride);
      const uint32_t km = s.drop_p > 0.f ? keep4(s, cx, (uint32_t)(pix * s.C + c)) : 0xFu;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gg = (km >> j) & 1u ? g[j] * cx.inv_keep : 0.f;
        float yv = z[j] * sc[j] + sh[j];
        gg = yv > 0.f ? gg : HPFG_LEAKY * gg;
        v[j] = k1[j] * gg + (k2[j] * z[j] + k3[j]);      // same association as the staged loaders (stage.h finish_piece)
      }
      break;
    }
    default:
      break;
  }
  return v;
}

// g-only part of the DZ chain (used by the BN-backward reduction): returns g and xhat for 4 channels.
__device__ static inline void dz_load_g_xhat(const HpfgAct& s, const ActCtx& cx, long pix, int c, f32x4& g_out, f32x4& xh_out) {
  f32x4 z = *reinterpret_cast<const f32x4*>(s.z + pix * s.pstride + c);
  f32x4 g = *reinterpret_cast<const f32x4*>(s.aux + pix * s.aux_pstride + c);
  const float* t = s.bn + s.bn_coff + c;
  f32x4 mu = *reinterpret_cast<const f32x4*>(t + HPFG_BN_MEAN * s.bn_stride);
  f32x4 rs = *reinterpret_cast<const f32x4*>(t + HPFG_BN_RSTD * s.bn_stride);
  f32x4 sc = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SCALE * s.bn_stride);
  f32x4 sh = *reinterpret_cast<const f32x4*>(t + HPFG_BN_SHIFT * s.bn_stride);
  const uint32_t km = s.drop_p > 0.f ? keep4(s, cx, (uint32_t)(pix * s.C + c)) : 0xFu;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float gg = (km >> j) & 1u ? g[j] * cx.inv_keep : 0.f;
    float yv = z[j] * sc[j] + sh[j];
    g_out[j] = yv > 0.f ? gg : HPFG_LEAKY * gg;
    xh_out[j] = (z[j] - mu[j]) * rs[j];
  }
}

// Select the source that owns virtual channel c of the concat [a0 | a1] and load 4 channels from it.
__device__ static inline f32x4 cat_load4(const HpfgAct& a0, const ActCtx& c0, const HpfgAct& a1, const ActCtx& c1, int n, int y, int x, int c) {
  if (c < a0.C || a1.mode == HPFG_ACT_NONE) return act_load4(a0, c0, n, y, x, c);
  return act_load4(a1, c1, n, y, x, c - a0.C);
}


// ---- single-mode loaders (compile-time mode): keep the conv kernels' staging code small and register-light -------------
enum { HPFG_KIND_PLAIN = 0, HPFG_KIND_BNACT = 1, HPFG_KIND_POOL = 2, HPFG_KIND_CAT = 3, HPFG_KIND_DZ = 4, HPFG_KIND_SPLIT = 5, HPFG_KIND_UPB = 6 };      // (SPLIT: HPFG_ACT_SPLIT16, weight gradient only; UPB: HPFG_ACT_UPBWD, 1x1 dgrad only)

template <int MODE>
__device__ static inline f32x4 act_load4_mode(const HpfgAct& s, const ActCtx& cx, int n, int y, int x, int c) {
  HpfgAct t = s;
  t.mode = MODE;          // constant-folds the switch in act_load4
  return act_load4(t, cx, n, y, x, c);
}

template <int KIND>
__device__ static inline f32x4 kind_load4(const HpfgAct& a0, const ActCtx& c0, const HpfgAct& a1, const ActCtx& c1, int n, int y, int x, int c) {
  if (KIND == HPFG_KIND_PLAIN) {
    if (a0.mode == HPFG_ACT_STRIDED) return act_load4_mode<HPFG_ACT_STRIDED>(a0, c0, n, y, x, c);
    return act_load4_mode<HPFG_ACT_PLAIN>(a0, c0, n, y, x, c);
  }
  if (KIND == HPFG_KIND_BNACT) return act_load4_mode<HPFG_ACT_BNACT>(a0, c0, n, y, x, c);
  if (KIND == HPFG_KIND_POOL) return act_load4_mode<HPFG_ACT_BNACT_POOL>(a0, c0, n, y, x, c);
  if (KIND == HPFG_KIND_DZ) return act_load4_mode<HPFG_ACT_DZ>(a0, c0, n, y, x, c);
  // CAT: [BNACT skip | bilinear-upsampled plain]
  if (c < a0.C) return act_load4_mode<HPFG_ACT_BNACT>(a0, c0, n, y, x, c);
  return act_load4_mode<HPFG_ACT_UP2X>(a1, c1, n, y, x, c - a0.C);
}

// transpose of the align_corners=True bilinear x2 (nn.Upsample, unet.py:51): the outputs o of a 2L-long axis that read source index `lo`, and
// with which weight -- at most 5 (on average 4); gather form of the backward pass (hpfg_upsample2x_bwd, and the HPFG_ACT_UPBWD loader)
__device__ inline void hpfg_up_taps(int lo, int L, int* idx, float* wgt, int& cnt) {
  const int O = 2 * L;
  const float r = O > 1 ? (float)(L - 1) / (float)(O - 1) : 0.f;
  cnt = 0;
  int b = 2 * lo - 2, e = 2 * lo + 4;
  if (b < 0) b = 0;
  if (e > O - 1) e = O - 1;
  for (int o = b; o <= e; ++o) {
    float f = r * (float)o;
    int i0 = (int)f;
    int i1 = i0 + (i0 < L - 1 ? 1 : 0);
    float w1 = f - (float)i0, w0 = 1.f - w1, w = 0.f;
    if (i0 == lo) w += w0;
    if (i1 == lo) w += w1;
    if ((i0 == lo || i1 == lo) && cnt < 8) {
      idx[cnt] = o;
      wgt[cnt] = w;
      ++cnt;
    }
  }
}
constexpr int HPFG_UPB_TAPS = 5;

// HPFG_NO_PK_F32 (kernel attribute): no packed fp32 VALU instructions in this kernel.  Round 5, first_wgrad_kernel: the broadcast products
// `acc[k] += tap * dz` (f32x4 by a scalar that sits in the HIGH dword of a register pair) compile to `v_pk_fma_f32 ... op_sel:[0,1,0]`, and on
// gfx950 / ROCm 7.2 the LOW half of such an instruction intermittently took the pair's low dword instead -- only with another stream's kernels
// on the chip, never alone: the gradient of the first conv moved by 1e-6 .. 1e-5 from run to run (tests/test_gpu_fullsize.py; the same source
// with volatile tap reads or per-component fmaf -- no cross-half op_sel in the ISA -- is bit-stable over hundreds of runs, explicit
// s_waitcnt lgkmcnt(0) in front of the products is not: profiles/r05_chain_and_prepass.txt).  Every kernel whose ISA contains that operand
// form is cured -- first_wgrad and upsample_bwd keep packed math and launder their broadcast scalars (hpfg_own_vgpr below: the attribute cost
// sup +4.3 %, mt +2.4 %), conv_first, adamw and the fp32 linear weight gradient carry the attribute; tools/pk_opsel_scan.sh lists offenders.
// hpfg_own_vgpr(x): the cheaper cure where it applies -- an empty asm that makes x a 32-bit value of its own, so a packed instruction broadcasts it
// from the LOW dword of a pair (op_sel_hi form) instead of selecting the high dword of the pair an LDS read returned it in.
__device__ __forceinline__ float hpfg_own_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
#if defined(HPFG_NO_PK_F32)
// (an A/B build defines it empty on the command line)
#elif defined(__HIP_DEVICE_COMPILE__)
#define HPFG_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define HPFG_NO_PK_F32
#endif

static inline int hpfg_kind_of(const HpfgAct& a0, const HpfgAct& a1) {
  if (a1.mode == HPFG_ACT_UP2X && a0.mode == HPFG_ACT_BNACT) return HPFG_KIND_CAT;
  if (a1.mode != HPFG_ACT_NONE) return -1;
  switch (a0.mode) {
    case HPFG_ACT_PLAIN: case HPFG_ACT_STRIDED: return HPFG_KIND_PLAIN;
    case HPFG_ACT_BNACT: return HPFG_KIND_BNACT;
    case HPFG_ACT_BNACT_POOL: return HPFG_KIND_POOL;
    case HPFG_ACT_DZ: return HPFG_KIND_DZ;
    case HPFG_ACT_SPLIT16: return HPFG_KIND_SPLIT;
    case HPFG_ACT_UPBWD: return HPFG_KIND_UPB;
    default: return -1;
  }
}
