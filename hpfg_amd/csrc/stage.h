// Staging helpers shared by the split-bf16 conv and wgrad kernels: per-kind raw loads of an 8-channel piece of a virtual
// activation (issue_piece), the producer chain on the loaded values (finish_piece), and the bf16 hi/lo split.
#pragma once
#include "common.h"

namespace hpfg_stage {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// x = hi + lo with hi = bf16(x) (round to nearest even), lo = bf16(x - hi): pairs go through v_cvt_pk_bf16_f32, and the packed
// hi word is widened back with one shift / one mask per pair.
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float x0 = k < 2 ? a[2 * k] : b[2 * k - 4], x1 = k < 2 ? a[2 * k + 1] : b[2 * k - 3];
    const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
    const uint32_t w = __builtin_bit_cast(uint32_t, h);
    const f32x2 hf = {__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
    const bf16x2 l = __builtin_convertvector(f32x2{x0, x1} - hf, bf16x2);
    hw[k] = w;
    lw[k] = __builtin_bit_cast(uint32_t, l);
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(bf16x8, (u32x4{hw[0], hw[1], hw[2], hw[3]}));
  lo = __builtin_bit_cast(bf16x8, (u32x4{lw[0], lw[1], lw[2], lw[3]}));
}

__device__ __forceinline__ f32x4 ld4(const float* base, int off) { return *reinterpret_cast<const f32x4*>(base + off); }

// ---- per-kind staging: issue_piece() puts the raw global loads of one 8-channel piece in flight, finish_piece() applies the
// ---- producer chain.  Raw loads per piece: PLAIN/BNACT 2, DZ 4 (z + dA), POOL 8 (2x2 pixels), CAT 8 (4 bilinear taps).
template <int KIND>
struct RawCount { static constexpr int N = KIND == HPFG_KIND_POOL || KIND == HPFG_KIND_CAT ? 8 : (KIND == HPFG_KIND_DZ ? 4 : 2); };

// Raw (in-flight) data of one staging piece: the float4 loads plus, for the kinds that apply Dropout, the two words of an
// explicit keep mask (HpfgAct.drop_mask, tests only).  The mask words are requested WITH the data loads: a load inside
// finish_piece() would sit behind every younger prefetch in the in-order vmcnt queue and drain the whole ring.
template <int KIND>
struct RawPiece {
  f32x4 v[RawCount<KIND>::N];
  uint32_t dm[2];
};

struct Tab {   // per-chunk per-channel tables of this thread's 8 channels
  f32x4 sc[2], sh[2], k1[2], k2[2], k3[2];
};

template <int KIND>
__device__ __forceinline__ void load_tables(Tab& t, const HpfgAct& a0, int c0, bool chvalid) {
  if (KIND == HPFG_KIND_PLAIN || KIND == HPFG_KIND_SPLIT || !chvalid) return;
  if (KIND == HPFG_KIND_CAT && c0 >= a0.C) return;
  const float* b = a0.bn + a0.bn_coff + c0;
  const int st = a0.bn_stride;
  t.sc[0] = ld4(b, HPFG_BN_SCALE * st);
  t.sc[1] = ld4(b, HPFG_BN_SCALE * st + 4);
  t.sh[0] = ld4(b, HPFG_BN_SHIFT * st);
  t.sh[1] = ld4(b, HPFG_BN_SHIFT * st + 4);
  if (KIND == HPFG_KIND_DZ) {
    t.k1[0] = ld4(b, HPFG_BN_K1 * st);
    t.k1[1] = ld4(b, HPFG_BN_K1 * st + 4);
    t.k2[0] = ld4(b, HPFG_BN_K2 * st);
    t.k2[1] = ld4(b, HPFG_BN_K2 * st + 4);
    t.k3[0] = ld4(b, HPFG_BN_K3 * st);
    t.k3[1] = ld4(b, HPFG_BN_K3 * st + 4);
  }
}

// The forward kinds (BNACT / POOL / concat skip half) read scale / shift of the whole source from LDS rows the kernel prologue filled
// (hpfg_bn_rows_to_lds: from the table, or derived from the producer's sum accumulators).
constexpr int HPFG_BN_CMAX = 256;      // channels of a BatchNorm'd source whose coefficients a consumer keeps in LDS
template <int KIND>
constexpr bool tab_in_lds() { return KIND == HPFG_KIND_BNACT || KIND == HPFG_KIND_POOL || KIND == HPFG_KIND_CAT || KIND == HPFG_KIND_DZ; }
template <int KIND>
constexpr int tab_rows() { return KIND == HPFG_KIND_DZ ? 5 : 2; }      // scale, shift [, k1, k2, k3]
template <int KIND>
__device__ __forceinline__ void load_tables_lds(Tab& t, const float* ldsT, const HpfgAct& a0, int c0) {
  if (KIND == HPFG_KIND_CAT && c0 >= a0.C) return;
  t.sc[0] = ld4(ldsT, c0);
  t.sc[1] = ld4(ldsT, c0 + 4);
  t.sh[0] = ld4(ldsT, HPFG_BN_CMAX + c0);
  t.sh[1] = ld4(ldsT, HPFG_BN_CMAX + c0 + 4);
  if (KIND == HPFG_KIND_DZ) {
    t.k1[0] = ld4(ldsT, 2 * HPFG_BN_CMAX + c0);
    t.k1[1] = ld4(ldsT, 2 * HPFG_BN_CMAX + c0 + 4);
    t.k2[0] = ld4(ldsT, 3 * HPFG_BN_CMAX + c0);
    t.k2[1] = ld4(ldsT, 3 * HPFG_BN_CMAX + c0 + 4);
    t.k3[0] = ld4(ldsT, 4 * HPFG_BN_CMAX + c0);
    t.k3[1] = ld4(ldsT, 4 * HPFG_BN_CMAX + c0 + 4);
  }
}
// the kernel prologue's fill of those rows
template <int KIND>
__device__ __forceinline__ void fill_tables_lds(const HpfgAct& a0, float* ldsT, int tid, int nthr) {
  if (KIND == HPFG_KIND_DZ) hpfg_dz_rows_to_lds(a0, ldsT, HPFG_BN_CMAX, a0.C, tid, nthr);
  else hpfg_bn_rows_to_lds(a0, ldsT, HPFG_BN_CMAX, tid, nthr);
}

// bilinear x2 (align_corners=True) source taps of output coordinate o for a low-res extent L
__device__ __forceinline__ void up_coord(int o, int L, int& i0, int& i1, float& w1) {
  const float r = L > 1 ? (float)(L - 1) / (float)(2 * L - 1) : 0.f;
  const float f = r * (float)o;
  i0 = (int)f;
  i1 = i0 + (i0 < L - 1 ? 1 : 0);
  w1 = f - (float)i0;
}

template <int KIND>
__device__ __forceinline__ void issue_piece(RawPiece<KIND>& rp, const HpfgAct& a, const HpfgAct& u, const ActCtx& cx0, int n, int gy,
                                            int gx, int c0, bool ok) {
  f32x4 (&raw)[RawCount<KIND>::N] = rp.v;
  if ((KIND == HPFG_KIND_BNACT || KIND == HPFG_KIND_DZ) && a.drop_mask && a.drop_p > 0.f) {
    const uint32_t e = (uint32_t)(((n * a.Hs + gy) * a.Ws + gx) * a.C + c0);
    rp.dm[0] = *reinterpret_cast<const uint32_t*>(a.drop_mask + e);
    rp.dm[1] = *reinterpret_cast<const uint32_t*>(a.drop_mask + e + 4);
  }
  // Branch-free on the pixel predicate: an out-of-image (halo) pixel is clamped into the image and loaded anyway, finish_piece()
  // selects zero for it.  Keeps the conv k-loop a single basic block so the compiler can software-pipeline LDS reads and MFMAs.
  if (KIND == HPFG_KIND_CAT) {      // two sources behind a per-thread branch anyway: keep the predicated form
#pragma unroll
    for (int i = 0; i < RawCount<KIND>::N; ++i) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!ok) return;
  }
  if (KIND == HPFG_KIND_SPLIT) {      // 8 channels of the stored (hi | lo) pair: two 16-byte words, carried as bit patterns
    const int off = ((n * a.Hs + gy) * a.Ws + gx) * a.pstride + c0;      // fp32 words == (hi, lo) pairs: 32 contiguous bytes per 8 channels
    raw[0] = ld4(a.z, off);
    raw[1] = ld4(a.z, off + 4);
  } else if (KIND == HPFG_KIND_PLAIN) {
    if (a.mode == HPFG_ACT_STRIDED) {
      raw[0] = act_load4_mode<HPFG_ACT_STRIDED>(a, cx0, n, gy, gx, c0);
      raw[1] = act_load4_mode<HPFG_ACT_STRIDED>(a, cx0, n, gy, gx, c0 + 4);
    } else {
      const int off = ((n * a.Hs + gy) * a.Ws + gx) * a.pstride + c0;
      raw[0] = ld4(a.z, off);
      raw[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c0 + 4 < a.C) raw[1] = ld4(a.z, off + 4);
    }
  } else if (KIND == HPFG_KIND_BNACT || (KIND == HPFG_KIND_CAT && c0 < a.C)) {
    const int off = ((n * a.Hs + gy) * a.Ws + gx) * a.pstride + c0;
    raw[0] = ld4(a.z, off);
    raw[1] = ld4(a.z, off + 4);
  } else if (KIND == HPFG_KIND_POOL) {
    const int off = ((n * a.Hs + 2 * gy) * a.Ws + 2 * gx) * a.pstride + c0;
    const int ro = a.Ws * a.pstride;
    raw[0] = ld4(a.z, off);
    raw[1] = ld4(a.z, off + 4);
    raw[2] = ld4(a.z, off + a.pstride);
    raw[3] = ld4(a.z, off + a.pstride + 4);
    raw[4] = ld4(a.z, off + ro);
    raw[5] = ld4(a.z, off + ro + 4);
    raw[6] = ld4(a.z, off + ro + a.pstride);
    raw[7] = ld4(a.z, off + ro + a.pstride + 4);
  } else if (KIND == HPFG_KIND_CAT) {   // upsampled half
    int y0, y1, x0, x1;
    float wy, wx;
    up_coord(gy, u.Hs, y0, y1, wy);
    up_coord(gx, u.Ws, x0, x1, wx);
    const int cb = n * u.Hs * u.Ws * u.pstride + (c0 - a.C);
    const int o00 = cb + (y0 * u.Ws + x0) * u.pstride, o01 = cb + (y0 * u.Ws + x1) * u.pstride;
    const int o10 = cb + (y1 * u.Ws + x0) * u.pstride, o11 = cb + (y1 * u.Ws + x1) * u.pstride;
    raw[0] = ld4(u.z, o00);
    raw[1] = ld4(u.z, o00 + 4);
    raw[2] = ld4(u.z, o01);
    raw[3] = ld4(u.z, o01 + 4);
    raw[4] = ld4(u.z, o10);
    raw[5] = ld4(u.z, o10 + 4);
    raw[6] = ld4(u.z, o11);
    raw[7] = ld4(u.z, o11 + 4);
  } else {   // DZ
    const int pix = (n * a.Hs + gy) * a.Ws + gx;
    raw[0] = ld4(a.z, pix * a.pstride + c0);
    raw[1] = ld4(a.z, pix * a.pstride + c0 + 4);
    raw[2] = ld4(a.aux, pix * a.aux_pstride + c0);
    raw[3] = ld4(a.aux, pix * a.aux_pstride + c0 + 4);
  }
}

// keep test of the four 8-bit draws (or mask bytes) in h against thr: v[j] = keep_j ? v[j] : 0
__device__ __forceinline__ void keep_select4(f32x4& v, uint32_t h, uint32_t thr) {
  v[0] = (h & 0xFFu) >= thr ? v[0] : 0.f;
  v[1] = ((h >> 8) & 0xFFu) >= thr ? v[1] : 0.f;
  v[2] = ((h >> 16) & 0xFFu) >= thr ? v[2] : 0.f;
  v[3] = (h >> 24) >= thr ? v[3] : 0.f;
}
__device__ __forceinline__ f32x4 lrelu4(const f32x4& y) { return __builtin_elementwise_max(y, y * HPFG_LEAKY); }

// The producer chain on the raw data of one piece.  Written on float4 values so that the multiplies / FMAs pair up into
// v_pk_fma_f32 / v_pk_mul_f32 (the loaders are VALU-issue bound: instruction count is what matters here).  `ok` = the piece is
// inside the image and its channel group exists; a piece that is not yields zeros (its loads were clamped to valid addresses).
template <int KIND>
__device__ __forceinline__ void finish_piece(f32x4& v0, f32x4& v1, const RawPiece<KIND>& rp, const Tab& t, const HpfgAct& a,
                                             const HpfgAct& u, const ActCtx& cx0, int n, int gy, int gx, int c0, bool ok) {
  const f32x4 (&raw)[RawCount<KIND>::N] = rp.v;
  bool need_ok = true;
  v0 = f32x4{0.f, 0.f, 0.f, 0.f};
  v1 = v0;
  if (KIND == HPFG_KIND_PLAIN || KIND == HPFG_KIND_SPLIT) {      // (SPLIT: v0 / v1 carry the hi / lo words; zero bits are bf16 zeros)
    v0 = raw[0];
    v1 = raw[1];
  } else if (KIND == HPFG_KIND_BNACT || (KIND == HPFG_KIND_CAT && c0 < a.C)) {
    v0 = lrelu4(raw[0] * t.sc[0] + t.sh[0]);
    v1 = lrelu4(raw[1] * t.sc[1] + t.sh[1]);
    if (KIND == HPFG_KIND_BNACT && a.drop_p > 0.f) {
      const uint32_t e = (uint32_t)(((n * a.Hs + gy) * a.Ws + gx) * a.C + c0);
      const uint32_t thr = drop_thresh(a, cx0);
      uint32_t h0 = a.drop_mask ? rp.dm[0] : hpfg_hash32(e >> 2, cx0.seed);
      uint32_t h1 = a.drop_mask ? rp.dm[1] : hpfg_hash32((e + 4) >> 2, cx0.seed);
      if (thr >= 1u) {       // a zero draw is always dropped: fold the in-image predicate into the draws
        h0 = ok ? h0 : 0u;
        h1 = ok ? h1 : 0u;
        need_ok = false;
      }
      v0 = v0 * cx0.inv_keep;
      v1 = v1 * cx0.inv_keep;
      keep_select4(v0, h0, thr);
      keep_select4(v1, h1, thr);
    }
  } else if (KIND == HPFG_KIND_POOL) {
    f32x4 m0 = lrelu4(raw[0] * t.sc[0] + t.sh[0]), m1 = lrelu4(raw[1] * t.sc[1] + t.sh[1]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      m0 = __builtin_elementwise_max(m0, lrelu4(raw[2 * q] * t.sc[0] + t.sh[0]));
      m1 = __builtin_elementwise_max(m1, lrelu4(raw[2 * q + 1] * t.sc[1] + t.sh[1]));
    }
    v0 = m0;
    v1 = m1;
  } else if (KIND == HPFG_KIND_CAT) {
    int i0, i1;
    float wy1, wx1;
    up_coord(gy, u.Hs, i0, i1, wy1);
    up_coord(gx, u.Ws, i0, i1, wx1);
    const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    v0 = wy0 * (wx0 * raw[0] + wx1 * raw[2]) + wy1 * (wx0 * raw[4] + wx1 * raw[6]);
    v1 = wy0 * (wx0 * raw[1] + wx1 * raw[3]) + wy1 * (wx0 * raw[5] + wx1 * raw[7]);
  } else {   // DZ: k1*g + k2*z + k3, g = dA * dropmask/(1-p) * lrelu'(scale*z+shift)
    f32x4 ga = raw[2], gb = raw[3];
    if (a.drop_p > 0.f) {
      const uint32_t e = (uint32_t)(((n * a.Hs + gy) * a.Ws + gx) * a.C + c0);
      const uint32_t thr = drop_thresh(a, cx0);
      ga = ga * cx0.inv_keep;
      gb = gb * cx0.inv_keep;
      keep_select4(ga, a.drop_mask ? rp.dm[0] : hpfg_hash32(e >> 2, cx0.seed), thr);
      keep_select4(gb, a.drop_mask ? rp.dm[1] : hpfg_hash32((e + 4) >> 2, cx0.seed), thr);
    }
    const f32x4 ya = raw[0] * t.sc[0] + t.sh[0], yb = raw[1] * t.sc[1] + t.sh[1];
    const f32x4 la = ga * HPFG_LEAKY, lb = gb * HPFG_LEAKY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ga[j] = ya[j] > 0.f ? ga[j] : la[j];
      gb[j] = yb[j] > 0.f ? gb[j] : lb[j];
    }
    v0 = t.k1[0] * ga + (t.k2[0] * raw[0] + t.k3[0]);
    v1 = t.k1[1] * gb + (t.k2[1] * raw[1] + t.k3[1]);
  }
  if (need_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {     // select, not branch: the raw values of a clamped (out-of-image / padding) piece are discarded
      v0[j] = ok ? v0[j] : 0.f;
      v1[j] = ok ? v1[j] : 0.f;
    }
  }
}

// (v0, v1) of finish_piece() -> the hi / lo bf16 fragments a staging store writes
template <int KIND>
__device__ __forceinline__ void split_piece(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& lo) {
  if (KIND == HPFG_KIND_SPLIT) {      // stored already split (HPFG_ACT_SPLIT16): a copy
    hi = __builtin_bit_cast(bf16x8, v0);
    lo = __builtin_bit_cast(bf16x8, v1);
  } else {
    split8(v0, v1, hi, lo);
  }
}

}  // namespace hpfg_stage
