// Staging helpers shared by the split-bf16 conv and wgrad kernels: per-kind raw loads of an 8-channel piece of a virtual
// activation (issue_piece), the producer chain on the loaded values (finish_piece), and the bf16 hi/lo split.
#pragma once
#include "common.h"

namespace hpfg_stage {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h0 = (__bf16)a[j], h1 = (__bf16)b[j];
    hi[j] = h0;
    hi[4 + j] = h1;
    lo[j] = (__bf16)(a[j] - (float)h0);
    lo[4 + j] = (__bf16)(b[j] - (float)h1);
  }
}

__device__ __forceinline__ f32x4 ld4(const float* base, int off) { return *reinterpret_cast<const f32x4*>(base + off); }

// ---- per-kind staging: issue_piece() puts the raw global loads of one 8-channel piece in flight, finish_piece() applies the
// ---- producer chain.  Raw loads per piece: PLAIN/BNACT 2, DZ 4 (z + dA), POOL 8 (2x2 pixels), CAT 8 (4 bilinear taps).
template <int KIND>
struct RawCount { static constexpr int N = KIND == HPFG_KIND_POOL || KIND == HPFG_KIND_CAT ? 8 : (KIND == HPFG_KIND_DZ ? 4 : 2); };

struct Tab {   // per-chunk per-channel tables of this thread's 8 channels
  f32x4 sc[2], sh[2], k1[2], k2[2], k3[2];
};

template <int KIND>
__device__ __forceinline__ void load_tables(Tab& t, const HpfgAct& a0, int c0, bool chvalid) {
  if (KIND == HPFG_KIND_PLAIN || !chvalid) return;
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

// bilinear x2 (align_corners=True) source taps of output coordinate o for a low-res extent L
__device__ __forceinline__ void up_coord(int o, int L, int& i0, int& i1, float& w1) {
  const float r = L > 1 ? (float)(L - 1) / (float)(2 * L - 1) : 0.f;
  const float f = r * (float)o;
  i0 = (int)f;
  i1 = i0 + (i0 < L - 1 ? 1 : 0);
  w1 = f - (float)i0;
}

template <int KIND>
__device__ __forceinline__ void issue_piece(f32x4 (&raw)[RawCount<KIND>::N], const HpfgAct& a, const HpfgAct& u, const ActCtx& cx0, int n, int gy,
                                            int gx, int c0, bool ok) {
  // Branch-free on the pixel predicate: an out-of-image (halo) pixel is clamped into the image and loaded anyway, finish_piece()
  // selects zero for it.  Keeps the conv k-loop a single basic block so the compiler can software-pipeline LDS reads and MFMAs.
  if (KIND == HPFG_KIND_CAT) {      // two sources behind a per-thread branch anyway: keep the predicated form
#pragma unroll
    for (int i = 0; i < RawCount<KIND>::N; ++i) raw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!ok) return;
  }
  if (KIND == HPFG_KIND_PLAIN) {
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

template <int KIND>
__device__ __forceinline__ void finish_piece(f32x4& v0, f32x4& v1, const f32x4 (&raw)[RawCount<KIND>::N], const Tab& t, const HpfgAct& a,
                                             const HpfgAct& u, const ActCtx& cx0, int n, int gy, int gx, int c0, bool ok) {
  v0 = f32x4{0.f, 0.f, 0.f, 0.f};
  v1 = v0;
  if (KIND == HPFG_KIND_PLAIN) {
    v0 = raw[0];
    v1 = raw[1];
  } else if (KIND == HPFG_KIND_BNACT || (KIND == HPFG_KIND_CAT && c0 < a.C)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v0[j] = lrelu(raw[0][j] * t.sc[0][j] + t.sh[0][j]);
      v1[j] = lrelu(raw[1][j] * t.sc[1][j] + t.sh[1][j]);
    }
    if (KIND == HPFG_KIND_BNACT && a.drop_p > 0.f) {
      const uint32_t e = (uint32_t)(((n * a.Hs + gy) * a.Ws + gx) * a.C + c0);
      const uint32_t k0 = keep4(a, cx0, e), k1 = keep4(a, cx0, e + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v0[j] = (k0 >> j) & 1u ? v0[j] * cx0.inv_keep : 0.f;
        v1[j] = (k1 >> j) & 1u ? v1[j] * cx0.inv_keep : 0.f;
      }
    }
  } else if (KIND == HPFG_KIND_POOL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float m0 = lrelu(raw[0][j] * t.sc[0][j] + t.sh[0][j]), m1 = lrelu(raw[1][j] * t.sc[1][j] + t.sh[1][j]);
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        m0 = fmaxf(m0, lrelu(raw[2 * q][j] * t.sc[0][j] + t.sh[0][j]));
        m1 = fmaxf(m1, lrelu(raw[2 * q + 1][j] * t.sc[1][j] + t.sh[1][j]));
      }
      v0[j] = m0;
      v1[j] = m1;
    }
  } else if (KIND == HPFG_KIND_CAT) {
    int i0, i1;
    float wy1, wx1;
    up_coord(gy, u.Hs, i0, i1, wy1);
    up_coord(gx, u.Ws, i0, i1, wx1);
    const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v0[j] = wy0 * (wx0 * raw[0][j] + wx1 * raw[2][j]) + wy1 * (wx0 * raw[4][j] + wx1 * raw[6][j]);
      v1[j] = wy0 * (wx0 * raw[1][j] + wx1 * raw[3][j]) + wy1 * (wx0 * raw[5][j] + wx1 * raw[7][j]);
    }
  } else {   // DZ: k1*g + k2*z + k3, g = dA * dropmask/(1-p) * lrelu'(scale*z+shift)
    uint32_t k0 = 0xFu, k1 = 0xFu;
    if (a.drop_p > 0.f) {
      const uint32_t e = (uint32_t)(((n * a.Hs + gy) * a.Ws + gx) * a.C + c0);
      k0 = keep4(a, cx0, e);
      k1 = keep4(a, cx0, e + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g0 = (k0 >> j) & 1u ? raw[2][j] * cx0.inv_keep : 0.f, g1 = (k1 >> j) & 1u ? raw[3][j] * cx0.inv_keep : 0.f;
      g0 = raw[0][j] * t.sc[0][j] + t.sh[0][j] > 0.f ? g0 : HPFG_LEAKY * g0;
      g1 = raw[1][j] * t.sc[1][j] + t.sh[1][j] > 0.f ? g1 : HPFG_LEAKY * g1;
      v0[j] = t.k1[0][j] * g0 + t.k2[0][j] * raw[0][j] + t.k3[0][j];
      v1[j] = t.k1[1][j] * g1 + t.k2[1][j] * raw[1][j] + t.k3[1][j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {     // select, not branch: the raw values of a clamped (out-of-image / padding) piece are discarded
    v0[j] = ok ? v0[j] : 0.f;
    v1[j] = ok ? v1[j] : 0.f;
  }
}


}  // namespace hpfg_stage
