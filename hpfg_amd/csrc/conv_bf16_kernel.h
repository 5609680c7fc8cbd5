// Implicit-GEMM 3x3 / 1x1 convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3" split precision).
//
// Every fp32 operand is split as x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a product is evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation inside v_mfma_f32_16x16x32_bf16 (the dropped lo*lo term is 2^-16 relative).
// Three bf16 MFMAs replace sixteen fp32-input MFMAs' worth of cycles (MI355X_MICROARCH.md: f32-in MFMA = 1/16 of the bf16 rate).
//
// Structure (3x3): persistent workgroups stream (tile, input-channel chunk) items through a double-buffered LDS tile.
//   * LDS tile: bf16 planes [channel group of 8][hi|lo][pixel slot][8 ch]; one A fragment = one ds_read_b128; plane size is a
//     multiple of 256 B and the 16 pixels of an MFMA tile map to distinct 16-B slots mod 16 -> conflict-free reads.
//   * K packing of one MFMA (K = 32): KC = 32 -> one tap x 32 input channels; KC = 16 -> two taps x 16 input channels
//     (16x16 tiles of the 16/32-channel layers; taps padded 9 -> 10 with zero weights).
//   * The MFMA is issued as D = W^T-fragment x pixel-fragment, so a lane ends up with 4 consecutive output channels of one
//     pixel: the epilogue is one 16-B store per lane per tile pair, 1 KB contiguous per wave instruction.
//   * Staging is the instruction-count hot spot of the thin (16/32-channel, 224^2/112^2) layers -- they are VALU-issue bound,
//     not MFMA bound -- so everything pixel-invariant (piece coordinates, LDS offsets, BatchNorm tables of the chunk) is
//     hoisted into registers, addresses are 32-bit, and the producer chain (BN + LeakyReLU + Dropout | MaxPool | bilinear)
//     is specialised per loader kind at compile time.
#pragma once
#include <stdlib.h>
#include "stage.h"

namespace hpfg_conv16 {

using namespace hpfg_stage;

template <int TH_, int TW_, int WM_, int WN_, int NI_, int TAPS_, int KC_>
struct Cfg {
  static constexpr int TH = TH_, TW = TW_, WM = WM_, WN = WN_, NI = NI_, TAPS = TAPS_, KC = KC_;
  static constexpr int MI = TH * TW / 16 / WM;
  static constexpr int BN = 16 * NI * WN;
  static constexpr int HALO = TAPS == 9 ? 1 : 0;
  static constexpr int HP = TH + 2 * HALO, WP = TW + 2 * HALO;
  static constexpr int RS = TW == 16 ? WP : (TAPS == 9 ? 24 : 8);   // row stride in 16-B slots
  static constexpr int NSLOT = (HP * RS + 15) / 16 * 16;
  static constexpr int NG = KC / 8;
  static constexpr int PLANE = NSLOT * 16;
  static constexpr int BUF_BYTES = NG * 2 * PLANE;
  static constexpr int KSTEPS = TAPS == 1 ? 1 : (KC == 32 ? 9 : 5);
  static constexpr int NPIECE = HP * WP * NG;
  static constexpr int NLD = (NPIECE + 255) / 256;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(256 % NG == 0, "a thread must keep one channel group");
  static_assert(KC == 32 || (KC == 16 && TAPS == 9), "two-tap K packing only for 3x3");
  static_assert(TAPS == 1 || NLD + 2 <= KSTEPS, "stage pipeline must fit into the k-steps of a chunk");
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }   // v_med3_i32

// per-thread, tile-invariant description of staging piece i: local pixel (ly, lx) and LDS byte offset
struct Piece {
  int ly, lx, lds, ok;
};
template <class C>
__device__ __forceinline__ Piece make_piece(int tid, int i) {
  Piece q;
  const int idx = tid + i * 256;
  const int pix = idx / C::NG, g = idx % C::NG;
  q.ok = idx < C::NPIECE;
  q.ly = pix / C::WP - C::HALO;
  q.lx = pix % C::WP - C::HALO;
  q.lds = (g * 2) * C::PLANE + ((pix / C::WP) * C::RS + pix % C::WP) * 16;
  return q;
}

// HpfgConvArgs.stage_out (SIDE_ON: the instantiations that offer it; `on` is workgroup-uniform: output-channel slice 0 only): the staged
// value of an interior, in-image piece (the virtual input of a forward conv, the dZ of a dgrad) also goes to memory -- as the (hi | lo) bf16
// pair the LDS image receives, bf16 [N][H][W][C / 8][hi 8 | lo 8]: the same 32 contiguous bytes per 8-channel piece an fp32 copy took, and the
// layer's weight gradient reads it back as an HPFG_ACT_SPLIT16 source without converting anything.  Halo pieces belong to a neighbouring
// tile's interior.
template <class C, int KIND = HPFG_KIND_PLAIN, bool SIDE_ON = false>
__device__ __forceinline__ void store_piece(unsigned char* buf, const Piece& q, const f32x4& v0, const f32x4& v1, const HpfgConvArgs* p = nullptr,
                                            bool on = false, int n = 0, int gy = 0, int gx = 0, int c0 = 0, bool ok = false) {
  if (!q.ok) return;
  bf16x8 hi, lo;
  split_piece<KIND>(v0, v1, hi, lo);
  *reinterpret_cast<bf16x8*>(buf + q.lds) = hi;
  *reinterpret_cast<bf16x8*>(buf + q.lds + C::PLANE) = lo;
  if constexpr (SIDE_ON) {
    if (on && ok && q.ly >= 0 && q.ly < C::TH && q.lx >= 0 && q.lx < C::TW) {
      const int ct = p->a0.C + p->a1.C;
      __bf16* d = reinterpret_cast<__bf16*>(p->stage_out) + ((((long)n * p->H + gy) * p->W + gx) * ct + c0) * 2;
      *reinterpret_cast<bf16x8*>(d) = hi;
      *reinterpret_cast<bf16x8*>(d + 8) = lo;
    }
  }
}

template <class C>
__device__ __forceinline__ void load_b(bf16x8 (&bh)[C::NI], bf16x8 (&bl)[C::NI], const bf16x8* wpk, int ks, int ntn, int nt0, int lane) {
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const bf16x8* q = wpk + ((ks * ntn + nt0 + j) * 2) * 64 + lane;
    bh[j] = q[0];
    bl[j] = q[64];
  }
}

// Epilogue: acc holds D[cout = (lane>>4)*4 + r][pixel = lane & 15] per (m, j) tile -> one 16-B store per lane.  The BatchNorm
// partial sums are accumulated per lane in (s1, s2) -- across ALL tiles of a persistent workgroup -- and flushed once.
template <class C>
__device__ __forceinline__ void conv16_load_bias(const HpfgConvArgs& p, f32x4 (&bias)[C::NI], int lane, int nt0) {
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const int co = (nt0 + j) * 16 + (lane >> 4) * 4;
    bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && co < p.CoutPad) bias[j] = ld4(p.bias, co);
  }
}

// BWD_OK (dgrad kernels: DZ / PLAIN sources): with p.bwd_stats the tile is the finished gradient w.r.t. the activated output of the
// BatchNorm layer p.bwd_of, and (s1, s2) collect that layer's backward sums  sum(g), sum(g * xhat)  with
// g = out * dropout * LeakyReLU'(bn(z))  -- the separate streaming pass over (dA, z) of hpfg_bn_bwd_reduce, done while the tile is
// still in registers (it costs one read of z instead of a read of both tensors, and a launch).
template <class C, int BWD_OK = 0>      // 0: forward statistics; 1: dgrad with p.bwd_stats == 1; 2: ... == 2 (max-pool backward epilogue)
__device__ __forceinline__ void conv16_store_tile(const HpfgConvArgs& p, f32x4 (&acc)[C::MI][C::NI], f32x4 (&s1)[C::NI], f32x4 (&s2)[C::NI],
                                                  const f32x4 (&bias)[C::NI], int lane, int wm, int nt0, int n, int ty0, int tx0,
                                                  bool reload_bias = false) {
  const int H = p.H, W = p.W;
  const bool vec = (p.Cout & 3) == 0 && (p.out_pstride & 3) == 0 && (p.out2_pstride & 3) == 0;
  const bool bwd = BWD_OK && p.bwd_stats;
  // bwd_stats == 2: the tile is dP, the gradient w.r.t. MaxPool2d(2) of bwd_of's activated output (bwd_of at 2H x 2W, its gradient so far -- the
  // skip path -- in bwd_of.aux).  The epilogue is the max-pool backward: it finds the window's arg-max from the raw output (first maximum in
  // row-major window order, like the forward kernel and torch), adds dP there IN PLACE, and takes the BatchNorm-backward sums of the completed
  // gradient from the registers -- the pass hpfg_bn_bwd_reduce_pool makes in a launch of its own; `out` is not written.
  const bool pool = BWD_OK == 2 && bwd;
  ActCtx bcx;
  if (bwd) bcx = make_ctx(p.bwd_of);
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    const int co = (nt0 + j) * 16 + (lane >> 4) * 4;
    f32x4 b = bias[j];
    if (reload_bias) {      // register-starved kernels keep no copy across the k-loop
      b = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && co < p.CoutPad) b = ld4(p.bias, co);
    }
    f32x4 tsc, tsh;      // only the sign of bn(z) is needed here: sum(g*xhat) = rstd * (sum(g*z) - mean * sum(g)) is finished in the flush
    if (bwd && co < p.Cout) {
      const float* tb = p.bwd_of.bn + p.bwd_of.bn_coff + co;
      tsc = ld4(tb, HPFG_BN_SCALE * p.bwd_of.bn_stride);
      tsh = ld4(tb, HPFG_BN_SHIFT * p.bwd_of.bn_stride);
    }
    // bwd: request the z values of all MI pixel tiles of this channel group before the first one is used (one exposed round trip per
    // channel group instead of one per tile when the scheduler would otherwise pair each load with its use)
    f32x4 zq[C::MI];
    if (bwd && !pool) {
#pragma unroll
      for (int m = 0; m < C::MI; ++m) {
        const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
        const int gy = min(ty0 + pxl / C::TW, H - 1), gx = min(tx0 + pxl % C::TW, W - 1);
        zq[m] = *reinterpret_cast<const f32x4*>(p.bwd_of.z + (((long)n * H + gy) * W + gx) * p.bwd_of.pstride + min(co, p.Cout - 4));
      }
    }
#pragma unroll
    for (int m = 0; m < C::MI; ++m) {
      const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
      const int gy = ty0 + pxl / C::TW, gx = tx0 + pxl % C::TW;
      if (pool) {
        if (gy < H && gx < W && co < p.Cout) {
          const f32x4 v = acc[m][j];
          float* dA = const_cast<float*>(p.bwd_of.aux);
          const long p00 = ((long)n * p.bwd_of.Hs + 2 * gy) * p.bwd_of.Ws + 2 * gx;
          f32x4 z4[4], g4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const long off = p00 + (k >> 1) * p.bwd_of.Ws + (k & 1);
            z4[k] = *reinterpret_cast<const f32x4*>(p.bwd_of.z + off * p.bwd_of.pstride + co);
            g4[k] = *reinterpret_cast<const f32x4*>(dA + off * p.bwd_of.aux_pstride + co);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = z4[k][r] * tsc[r] + tsh[r];
            float best = lrelu(y[0]);
            int bi = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              const float a = lrelu(y[k]);
              if (a > best) {
                best = a;
                bi = k;
              }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              g4[k][r] += bi == k ? v[r] : 0.f;
              const float gg = y[k] > 0.f ? g4[k][r] : HPFG_LEAKY * g4[k][r];
              s1[j][r] += gg;
              s2[j][r] += gg * z4[k][r];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const long off = p00 + (k >> 1) * p.bwd_of.Ws + (k & 1);
            *reinterpret_cast<f32x4*>(dA + off * p.bwd_of.aux_pstride + co) = g4[k];
          }
        }
        continue;
      }
      if (gy < H && gx < W && co < p.Cout) {
        f32x4 v = acc[m][j] + b;
        float* o = (p.out_split && co >= p.out_split) ? p.out2 + ((n * H + gy) * W + gx) * p.out2_pstride + (co - p.out_split)
                                                       : p.out + ((n * H + gy) * W + gx) * p.out_pstride + co;
        if (p.math & 0x100) {
        } else if (vec) {
          *reinterpret_cast<f32x4*>(o) = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (co + r < p.Cout) o[r] = v[r];
            else v[r] = 0.f;
          }
        }
        if (bwd) {
          const long pix = ((long)n * H + gy) * W + gx;
          const f32x4 z = zq[m];
          uint32_t km = 0xFu;
          if (p.bwd_of.drop_p > 0.f) km = keep4(p.bwd_of, bcx, (uint32_t)(pix * p.bwd_of.C + co));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float gg = (km >> r) & 1u ? v[r] * bcx.inv_keep : 0.f;
            gg = z[r] * tsc[r] + tsh[r] > 0.f ? gg : HPFG_LEAKY * gg;
            s1[j][r] += gg;
            s2[j][r] += gg * z[r];
          }
        } else {
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
  }
}

// Reduce the per-lane partial sums over the 16 pixel lanes and the WM waves that share output channels; row `row` of
// stat_partials ([rows][2][CoutPad]) receives this workgroup's sum(z), sum(z^2).
template <class C, int BWD = 0>
__device__ __forceinline__ void conv16_flush_stats(const HpfgConvArgs& p, f32x4 (&s1)[C::NI], f32x4 (&s2)[C::NI], float* ldsf, int tid, int lane, int wm,
                                                   int wn, int cb, int row) {
  if ((!p.stat_partials && !p.stat_acc) || (p.math & 0x1000)) return;
#pragma unroll
  for (int j = 0; j < C::NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = s1[j][r], b = s2[j][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
      }
      s1[j][r] = a;
      s2[j][r] = b;
    }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int j = 0; j < C::NI; ++j) {
      const int cl = (wn * C::NI + j) * 16 + (lane >> 4) * 4;
      *reinterpret_cast<f32x4*>(ldsf + (0 * C::WM + wm) * C::BN + cl) = s1[j];
      *reinterpret_cast<f32x4*>(ldsf + (1 * C::WM + wm) * C::BN + cl) = s2[j];
    }
  }
  __syncthreads();
  if (tid < 2 * C::BN) {
    const int which = tid / C::BN, cl = tid % C::BN;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < C::WM; ++w) t += ldsf[(which * C::WM + w) * C::BN + cl];
    const int co = cb * C::BN + cl;
    if (BWD && p.bwd_stats && which == 1 && co < p.Cout) {       // sum(g*z) -> sum(g*xhat)
      float sg = 0.f;
#pragma unroll
      for (int w = 0; w < C::WM; ++w) sg += ldsf[w * C::BN + cl];
      const float* tb = p.bwd_of.bn + p.bwd_of.bn_coff + co;
      t = tb[HPFG_BN_RSTD * p.bwd_of.bn_stride] * (t - tb[HPFG_BN_MEAN * p.bwd_of.bn_stride] * sg);
    }
    if (p.stat_acc && co < p.Cout)      // integer atomics into the layer accumulator: no finalize launch needs rows
      hpfg_acc_add(p.stat_acc, p.CoutPad, ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x & (p.stat_shards - 1), which, co, t);
    if (p.stat_partials && co < p.CoutPad) p.stat_partials[((long)row * 2 + which) * p.CoutPad + co] = t;
  }
}

#define HPFG16_MFMA3(ACC, AH, AL, BH, BL)                              \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BH, AH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BL, AH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BH, AL, ACC, 0, 0, 0);

// Concat loader (KIND == CAT), upsampled half: the bilinear x2 taps of a tile come from a (TH/2+3) x (TW/2+3) patch of the
// low-resolution 1x1-conv output.  That patch is fetched ONCE per chunk (at most one 8-channel piece per thread, two float4:
// prefetched a chunk ahead like the cheap loaders), parked in LDS as fp32, and every output piece interpolates its four taps
// from LDS -- instead of 8 float4 global loads per output piece (24 per thread and chunk), whose registers forced a
// two-k-step pipeline that exposed one memory latency per piece.  Skip-half chunks are plain BN + LeakyReLU pieces.
template <class C>
struct UpGeo {
  static constexpr int SH = C::TH / 2 + 3, SW = C::TW / 2 + 3;
  static constexpr int NPIECE = SH * SW * C::NG;
  static constexpr int BYTES = SH * SW * C::KC * 4;
  static_assert(NPIECE <= 256, "one source piece per thread");
};
template <int KIND>
constexpr int eff_nr() { return KIND == HPFG_KIND_CAT ? 2 : RawCount<KIND>::N; }   // float4 loads in flight per piece

// first low-res row / column a tile (with its halo) touches: source index of output coordinate max(o0, 0)
__device__ __forceinline__ int up_base(int o0, int L) {
  const float r = L > 1 ? (float)(L - 1) / (float)(2 * L - 1) : 0.f;
  return (int)(r * (float)(o0 < 0 ? 0 : o0));
}

// Resident workgroups per CU the kernel is compiled for.  The thin 16x16-tile layers behind a cheap loader are HBM-bound: three
// workgroups per CU keep more tile loads in flight; everything else needs the registers of a two-per-CU budget.
template <class C, int KIND>
constexpr int wg_per_cu() {
  return (C::KSTEPS == 5 && C::NI == 1 && RawCount<KIND>::N <= 2) ? 3 : 2;      // (the concat kernel would spill at 168 VGPRs)
}

// Diagnostics build only (make TRACE=1 -> libhpfg_hip_trace.so, tools/trace_conv.py): with math bit 0x2000 wave 0 of every
// workgroup writes (id << 56 | s_memtime) stamps to stat_partials + 256 * blockIdx.x (u64), which the tool sizes accordingly.
#ifdef HPFG_TRACE
#define HPFG_TR(ID)                                                                                        \
  if (tr_on && tr_i < 255) {                                                                               \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffffull); \
  }
#define HPFG_TR_REAL(ID)                                                                                   \
  if (tr_on && tr_i < 255) {                                                                               \
    tr_buf[tr_i++] = ((unsigned long long)(ID) << 56) | (__builtin_amdgcn_s_memrealtime() & 0xffffffffffffffull); \
  }
#else
#define HPFG_TR(ID)
#define HPFG_TR_REAL(ID)
#endif

// BWD: the dgrad variant whose epilogue also produces the BatchNorm-backward sums of the layer below (p.bwd_stats); a separate
// instantiation so that the plain kernels keep their register budget.
template <class C, int KIND, int BWD = 0>
__global__ __launch_bounds__(256, (wg_per_cu<C, KIND>())) void conv_bf16x3_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 9, "persistent kernel is the 3x3 path");
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  constexpr int NR = eff_nr<KIND>();
  constexpr bool CATK = KIND == HPFG_KIND_CAT;
  constexpr int UP_BYTES = CATK ? UpGeo<C>::BYTES : 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * C::BUF_BYTES + STAT_BYTES + UP_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds + 2 * C::BUF_BYTES);
  float* ldsU = reinterpret_cast<float*>(lds + 2 * C::BUF_BYTES + STAT_BYTES);      // low-res source patch of an upsampled chunk
  (void)ldsU;
  constexpr bool TABK = tab_in_lds<KIND>();
  __shared__ __attribute__((aligned(16))) float ldsBN[TABK ? tab_rows<KIND>() * HPFG_BN_CMAX : 4];      // scale | shift [| k1 | k2 | k3] of every channel of the source
  (void)ldsBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % C::WM, wn = wave / C::WM;
  const int cb = blockIdx.y;
  const int H = p.H, W = p.W;
  const int ntiles = tiles_x * tiles_y, nwork = ntiles * p.N;
  const ActCtx cx0 = make_ctx(p.a0);
  // HpfgConvArgs.stage_out (the 4 x 16-pixel-tile kernels of the sizes that are not multiples of 16 only: the engine's aligned layers run the
  // fused / thin kernels, and the aligned instantiations have no registers to spare): this workgroup also stores what it stages (slice 0 only)
  constexpr bool SIDE = KIND != HPFG_KIND_PLAIN && C::TH == 4;
  const bool dzw = SIDE && p.stage_out != nullptr && cb == 0;
  (void)dzw;
#ifdef HPFG_TRACE
  const bool tr_on = (p.math & 0x2000) && tid == 0;
  unsigned long long* tr_buf = reinterpret_cast<unsigned long long*>(p.stat_partials) + 256 * ((long)blockIdx.y * gridDim.x + blockIdx.x);
  int tr_i = 0;
#endif
  HPFG_TR_REAL(11)
  HPFG_TR(1)

  f32x4 acc[C::MI][C::NI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kg = lane >> 4;
  const int gl = C::KC == 32 ? kg : (kg & 1);
  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (gl * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }
  // byte offsets of the k-steps' taps for this lane (KC=16 packs taps 2s, 2s+1 into one MFMA; tap 9 re-reads tap 8: zero weights)
  int toff[C::KSTEPS];
#pragma unroll
  for (int s = 0; s < C::KSTEPS; ++s) {
    int tap = C::KC == 32 ? s : 2 * s + (kg >> 1);
    tap = tap > 8 ? 8 : tap;
    toff[s] = ((tap / 3) * C::RS + (tap % 3)) * 16;
  }
  Piece pc[C::NLD];
#pragma unroll
  for (int i = 0; i < C::NLD; ++i) pc[i] = make_piece<C>(tid, i);
  const int g8 = (tid % C::NG) * 8;          // this thread's channel group inside a chunk

  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + C::KC - 1) / C::KC;
  const int ntn = p.CoutPad / 16;
  const int nt0 = (cb * C::WN + wn) * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);

  // XCD-aware work mapping: workgroups are dealt round-robin over the 8 XCDs (b % 8 names the XCD group), each XCD has its own
  // L2.  Give every XCD group a CONTIGUOUS range of tiles (row-major inside an image) so that tiles sharing a halo are read
  // through the same L2; inside a group, workgroup j takes tiles j, j + G, j + 2G, ... of the range.  (Speed only: any mapping
  // is correct.)
  const int nx = gridDim.x >= 8 ? 8 : 1;
  const int xg = (int)blockIdx.x % nx, xj = (int)blockIdx.x / nx;
  const int per_x = (nwork + nx - 1) / nx;                 // tiles per XCD group
  const int wend = (xg + 1) * per_x < nwork ? (xg + 1) * per_x : nwork;
  const int G = ((int)gridDim.x - xg + nx - 1) / nx;       // workgroups in this group
  int w = xg * per_x + xj;
  if (w >= wend) {   // no tile for this workgroup: its BatchNorm partial row must still be defined
    if (p.stat_partials && threadIdx.x < 2 * C::BN) {
      const int co = cb * C::BN + (int)threadIdx.x % C::BN;
      if (co < p.CoutPad) p.stat_partials[((long)blockIdx.x * 2 + (int)threadIdx.x / C::BN) * p.CoutPad + co] = 0.f;
    }
    return;
  }
  int n = w / ntiles, tyi = (w % ntiles) / tiles_x, txi = (w % ntiles) % tiles_x;
  int ty0 = tyi * C::TH, tx0 = txi * C::TW;
  // stepping to this workgroup's next tile (w += G) without divisions: decompose the stride once
  const int gn = G / ntiles, gty = (G % ntiles) / tiles_x, gtx = (G % ntiles) % tiles_x;
  f32x4 s1[C::NI], s2[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
  }
  constexpr bool TIGHT = C::TAPS == 9 && C::KSTEPS == 5 && C::NI >= 2 && eff_nr<KIND>() >= 4;
  f32x4 bias[C::NI];       // per-workgroup constant: fetched once, not per tile in the epilogue
  if (!TIGHT) conv16_load_bias<C>(p, bias, lane, nt0);
  Tab tab;
  HPFG_TR(2)
  {
    const int c0 = g8;
    const bool chv = c0 < cin_total;
    const int c0c = chv ? c0 : 0;
    // the first tile's raw loads go out before the BatchNorm coefficients are formed (sum accumulators -> scale / shift in LDS): one
    // exposed round trip for both
    RawPiece<KIND> raw0[C::NLD];
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const int gy = ty0 + pc[i].ly, gx = tx0 + pc[i].lx;
      const bool ok = pc[i].ok && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
      issue_piece<KIND>(raw0[i], p.a0, p.a1, cx0, n, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), c0c, ok);
    }
    if constexpr (TABK) {
      fill_tables_lds<KIND>(p.a0, ldsBN, tid, 256);
      __syncthreads();
      load_tables_lds<KIND>(tab, ldsBN, p.a0, c0c);
    } else {
      load_tables<KIND>(tab, p.a0, c0, chv);
    }
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const int gy = ty0 + pc[i].ly, gx = tx0 + pc[i].lx;
      const bool ok = pc[i].ok && chv && gy >= 0 && gy < H && gx >= 0 && gx < W;
      f32x4 v0, v1;
      const int gyc = clampi(gy, 0, H - 1), gxc = clampi(gx, 0, W - 1);
      finish_piece<KIND>(v0, v1, raw0[i], tab, p.a0, p.a1, cx0, n, gyc, gxc, c0c, ok);
      store_piece<C, KIND, SIDE>(lds, pc[i], v0, v1, &p, dzw, n, gyc, gxc, c0c, ok);
    }
  }
  // B-fragment ring: the fragments of k-step g + BD are requested while k-step g computes (BD = ring size - 1 k-steps of latency
  // cover; the ring position of a k-step is static because the ring size divides KSTEPS).  Weights do not depend on the tile, so
  // the k-step sequence is periodic with period nchunks * KSTEPS.
  // vmcnt retires in issue order: a B fragment requested AFTER the tile prefetch of the next chunk would make its consumer wait
  // for that whole prefetch.  With a ring as deep as a chunk (NI == 1: 8 VGPRs per k-step) every fragment a chunk consumes was
  // requested during the previous chunk, i.e. before this chunk's prefetch, and the k-loop never waits on it.
  constexpr int BR = C::KSTEPS == 5 ? 5 : (C::NI == 1 ? 9 : 3), BD = BR - 1;
  static_assert(C::KSTEPS % BR == 0, "ring size must divide the k-steps of a chunk");
  const int kperiod = nchunks * C::KSTEPS;
  bf16x8 bh[BR][C::NI], bl[BR][C::NI];
#pragma unroll
  for (int d = 0; d < BD; ++d) load_b<C>(bh[d], bl[d], wpk, d % kperiod, ntn, nt0, lane);
  HPFG_TR(3)
  __syncthreads();
  HPFG_TR(4)
  int item = 0;
  while (true) {
    for (int ch = 0; ch < nchunks; ++ch, ++item) {
      const unsigned char* cur = lds + (item & 1) * C::BUF_BYTES;
      unsigned char* nxt = lds + ((item + 1) & 1) * C::BUF_BYTES;
      const bool last_chunk = ch + 1 == nchunks;
      const int w2 = w + G;
      const bool more = !last_chunk || w2 < wend;
      const int nch = last_chunk ? 0 : ch + 1;
      int nn = n, nty = ty0, ntx = tx0;
      if (last_chunk && more) {
        int x2 = txi + gtx, y2 = tyi + gty;
        nn = n + gn;
        if (x2 >= tiles_x) {
          x2 -= tiles_x;
          ++y2;
        }
        if (y2 >= tiles_y) {
          y2 -= tiles_y;
          ++nn;
        }
        nty = y2 * C::TH;
        ntx = x2 * C::TW;
      }
      const int c0n = nch * C::KC + g8;
      const bool chvn = more && c0n < cin_total && !(p.math & 0x400);
      const int c0c = c0n < cin_total ? c0n : 0;
      // No branches from here to the barrier (except the CAT loader's per-thread source select): the tables are reloaded even when
      // they cannot have changed, so that the whole chunk is one scheduling region with a hand-placed instruction order.
      if constexpr (TABK) load_tables_lds<KIND>(tab, ldsBN, p.a0, c0c);
      else if (!TIGHT || (more && nchunks > 1)) load_tables<KIND>(tab, p.a0, c0c, true);
      // Prefetch depth: kinds with few raw loads per piece (PLAIN/BNACT) put ALL pieces of the next item in flight before the
      // first MFMA and finish them after the last one (a whole item of latency cover); DZ (4 raw float4 + 10 table registers) and
      // POOL/CAT (8 raw float4 per piece) keep the two-k-step ring to stay inside the register budget.
      // The channel-rich layers (KC = 32; not the concat loader, whose upsampled chunks take the LDS-patch path) interleave instead: one wave
      // per SIMD there, and issue -> k-loop -> convert in sequence left the MFMA pipe idle for half of every chunk (in-kernel timeline:
      // 2957 cycles of k-loop + 2249 of conversion per chunk against 1728 cycles of MFMAs).
      constexpr bool DEEP = (NR <= 2 || C::NLD <= 2) && (C::KC != 32 || CATK);
      // concat: a chunk is entirely skip half (BN + LeakyReLU pieces, like BNACT) or entirely upsampled half (a0.C % KC == 0)
      constexpr int SK = CATK ? HPFG_KIND_BNACT : KIND;        // loader kind of the per-pixel pieces held in `raw`
      RawPiece<SK> raw[DEEP ? C::NLD : 2];
      const bool up_next = CATK && nch * C::KC >= p.a0.C;       // workgroup-uniform
      f32x4 rawU[2];
      int sy_base = 0, sx_base = 0;
      if (CATK && up_next) {
        using UG = UpGeo<C>;
        sy_base = up_base(nty - 1, p.a1.Hs);
        sx_base = up_base(ntx - 1, p.a1.Ws);
        const int pix = tid / C::NG;
        const int sy = clampi(sy_base + pix / UG::SW, 0, p.a1.Hs - 1), sx = clampi(sx_base + pix % UG::SW, 0, p.a1.Ws - 1);
        const int cu = (c0n < cin_total ? c0n : p.a0.C) - p.a0.C;                 // channel inside the upsampled tensor
        const int off = ((nn * p.a1.Hs + sy) * p.a1.Ws + sx) * p.a1.pstride + cu;
        rawU[0] = ld4(p.a1.z, off);
        rawU[1] = ld4(p.a1.z, off + 4);
      } else if (DEEP) {
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
          const int gy = nty + pc[i].ly, gx = ntx + pc[i].lx;
          const bool ok = pc[i].ok && chvn && gy >= 0 && gy < H && gx >= 0 && gx < W;
          issue_piece<SK>(raw[DEEP ? i : 0], p.a0, p.a1, cx0, nn, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), c0c, ok);
        }
      }
      HPFG_TR(5)
      // k-loop as a linear sequence of (k-step, pixel-tile) steps q: the A fragments of step q + AD are read from LDS while step q
      // multiplies (ring of AD + 1 fragment pairs), so an LDS read has AD - 1 steps of MFMAs to land.  The scheduler is fenced at
      // every step: left alone it sinks each read down to its use and exposes the LDS latency once per k-step.
      constexpr int Q = C::KSTEPS * C::MI;
      constexpr int AD0 = wg_per_cu<C, KIND>() == 3 ? 2 : (C::NI == 1 ? 4 : (TIGHT ? 0 : 2));   // TIGHT (16x16 tiles x 32 output channels behind a wide loader): no registers for read-ahead
      constexpr int AD = AD0 < Q ? AD0 : Q;
      constexpr int AR = AD + 1;
      constexpr bool FENCE = !TIGHT;   // those three kernels have no registers to spare for a pinned order
      bf16x8 ah[AR], al[AR];
#pragma unroll
      for (int q = 0; q < AD; ++q) {
        ah[q % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q % C::MI] + toff[q / C::MI]);
        al[q % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q % C::MI] + toff[q / C::MI] + C::PLANE);
      }
      if (FENCE) __builtin_amdgcn_sched_barrier(0x216);   // VALU, SALU, VMEM and DS writes may cross; DS reads and MFMAs may not
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int s = q / C::MI, m = q % C::MI;
        if (q + AD < Q) {
          const int q2 = q + AD;
          ah[q2 % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q2 % C::MI] + toff[q2 / C::MI]);
          al[q2 % AR] = *reinterpret_cast<const bf16x8*>(cur + aoff[q2 % C::MI] + toff[q2 / C::MI] + C::PLANE);
        }
        if (m == 0) {
          if (!DEEP) {
            if (s >= 2 && s - 2 < C::NLD) {        // finish the piece issued two k-steps ago and park it in the other LDS buffer
              const int i = s - 2 < C::NLD ? s - 2 : 0;
              const int gy = nty + pc[i].ly, gx = ntx + pc[i].lx;
              const bool ok = pc[i].ok && chvn && gy >= 0 && gy < H && gx >= 0 && gx < W;
              f32x4 v0, v1;       // also on the last item (more == false): the parked piece lands in the unused buffer, no branch
              finish_piece<SK>(v0, v1, raw[i & 1], tab, p.a0, p.a1, cx0, nn, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), c0c, ok);
              store_piece<C, SK, SIDE>(nxt, pc[i], v0, v1, &p, dzw, nn, gy, gx, c0c, ok);
            }
            if (s < C::NLD) {
              const int i = s < C::NLD ? s : 0;
              const int gy = nty + pc[i].ly, gx = ntx + pc[i].lx;
              const bool ok = pc[i].ok && chvn && gy >= 0 && gy < H && gx >= 0 && gx < W;
              issue_piece<SK>(raw[i & 1], p.a0, p.a1, cx0, nn, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), c0c, ok);
            }
          }
          int ksn = ch * C::KSTEPS + s + BD;
          ksn = ksn >= kperiod ? ksn - kperiod : ksn;       // BD < KSTEPS <= kperiod: one subtraction wraps
          load_b<C>(bh[(s + BD) % BR], bl[(s + BD) % BR], wpk, ksn, ntn, nt0, lane);
        }
#pragma unroll
        for (int j = 0; j < C::NI; ++j) { HPFG16_MFMA3(acc[m][j], ah[q % AR], al[q % AR], bh[s % BR][j], bl[s % BR][j]) }
        if (FENCE) __builtin_amdgcn_sched_barrier(0x216);   // VALU, SALU, VMEM and DS writes may cross; DS reads and MFMAs may not
      }
      HPFG_TR(6)
      if (CATK && up_next) {
        using UG = UpGeo<C>;
        // park the low-res patch (fp32, [pixel][KC channels]), then every output piece blends its four taps from LDS
        if (tid < UG::NPIECE) {
          float* d = ldsU + (tid / C::NG) * C::KC + g8;
          *reinterpret_cast<f32x4*>(d) = rawU[0];
          *reinterpret_cast<f32x4*>(d + 4) = rawU[1];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
          const int gy = nty + pc[i].ly, gx = ntx + pc[i].lx;
          const bool ok = pc[i].ok && chvn && gy >= 0 && gy < H && gx >= 0 && gx < W;
          int y0, y1, x0, x1;
          float wy1, wx1;
          up_coord(clampi(gy, 0, H - 1), p.a1.Hs, y0, y1, wy1);
          up_coord(clampi(gx, 0, W - 1), p.a1.Ws, x0, x1, wx1);
          const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
          const float* r0 = ldsU + ((y0 - sy_base) * UG::SW) * C::KC + g8;
          const float* r1 = ldsU + ((y1 - sy_base) * UG::SW) * C::KC + g8;
          const int o0 = (x0 - sx_base) * C::KC, o1 = (x1 - sx_base) * C::KC;
          const f32x4 a00 = *reinterpret_cast<const f32x4*>(r0 + o0), b00 = *reinterpret_cast<const f32x4*>(r0 + o0 + 4);
          const f32x4 a01 = *reinterpret_cast<const f32x4*>(r0 + o1), b01 = *reinterpret_cast<const f32x4*>(r0 + o1 + 4);
          const f32x4 a10 = *reinterpret_cast<const f32x4*>(r1 + o0), b10 = *reinterpret_cast<const f32x4*>(r1 + o0 + 4);
          const f32x4 a11 = *reinterpret_cast<const f32x4*>(r1 + o1), b11 = *reinterpret_cast<const f32x4*>(r1 + o1 + 4);
          f32x4 v0 = wy0 * (wx0 * a00 + wx1 * a01) + wy1 * (wx0 * a10 + wx1 * a11);     // same expression order as finish_piece<CAT>
          f32x4 v1 = wy0 * (wx0 * b00 + wx1 * b01) + wy1 * (wx0 * b10 + wx1 * b11);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v0[j] = ok ? v0[j] : 0.f;
            v1[j] = ok ? v1[j] : 0.f;
          }
          store_piece<C, HPFG_KIND_PLAIN, SIDE>(nxt, pc[i], v0, v1, &p, dzw, nn, gy, gx, c0c, ok);
        }
      } else if (DEEP) {      // also on the last item: the pieces land in the unused buffer
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
          const int gy = nty + pc[i].ly, gx = ntx + pc[i].lx;
          const bool ok = pc[i].ok && chvn && gy >= 0 && gy < H && gx >= 0 && gx < W;
          f32x4 v0, v1;
          finish_piece<SK>(v0, v1, raw[DEEP ? i : 0], tab, p.a0, p.a1, cx0, nn, clampi(gy, 0, H - 1), clampi(gx, 0, W - 1), c0c, ok);
          store_piece<C, SK, SIDE>(nxt, pc[i], v0, v1, &p, dzw, nn, gy, gx, c0c, ok);
        }
      }
      HPFG_TR(7)
      __syncthreads();
      HPFG_TR(8)
    }
    conv16_store_tile<C, BWD>(p, acc, s1, s2, bias, lane, wm, nt0, n, ty0, tx0, TIGHT);
    HPFG_TR(9)
#pragma unroll
    for (int m = 0; m < C::MI; ++m)
#pragma unroll
      for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    w += G;
    if (w >= wend) break;
    txi += gtx;
    tyi += gty;
    n += gn;
    if (txi >= tiles_x) {
      txi -= tiles_x;
      ++tyi;
    }
    if (tyi >= tiles_y) {
      tyi -= tiles_y;
      ++n;
    }
    ty0 = tyi * C::TH;
    tx0 = txi * C::TW;
  }
  conv16_flush_stats<C, BWD>(p, s1, s2, ldsf, tid, lane, wm, wn, cb, (int)blockIdx.x);
  HPFG_TR(10)
  HPFG_TR_REAL(12)
}

// 1x1: one tile per workgroup, K = 32 input channels per MFMA step, no halo.
template <class C, int KIND, int BWD = 0>
__global__ __launch_bounds__(256) void conv1x1_bf16x3_kernel(HpfgConvArgs p, int tiles_x, int tiles_y) {
  static_assert(C::TAPS == 1, "1x1 path");
  constexpr int STAT_BYTES = 2 * 4 * C::BN * 4;
  constexpr int NR = RawCount<KIND>::N;
  __shared__ __attribute__((aligned(16))) unsigned char lds[C::BUF_BYTES + STAT_BYTES];
  float* ldsf = reinterpret_cast<float*>(lds + C::BUF_BYTES);
  constexpr bool TABK = tab_in_lds<KIND>();
  __shared__ __attribute__((aligned(16))) float ldsBN[TABK ? tab_rows<KIND>() * HPFG_BN_CMAX : 4];
  (void)ldsBN;
  // HPFG_ACT_UPBWD source (the dgrad of a decoder block's 1x1 conv, unet.py:50-51): the staged value of a pixel is the transposed bilinear
  // interpolation of the gradient w.r.t. the upsampled tensor -- hpfg_upsample2x_bwd's gather (same taps, same order of additions), made
  // while the tile is staged instead of by a launch of its own in front of this one.  Output-channel slice 0 also stores the gathered
  // tensor (p.stage_out: the dZ of the 1x1 conv's weight gradient) and its per-workgroup channel sums (p.side_sums: the bias gradient).
  constexpr bool UPB = KIND == HPFG_KIND_UPB;
  __shared__ short upI[UPB ? C::TH + C::TW : 1][HPFG_UPB_TAPS];
  __shared__ float upW[UPB ? C::TH + C::TW : 1][HPFG_UPB_TAPS];
  __shared__ __attribute__((aligned(16))) float upS[UPB ? 256 * 8 : 4];
  (void)upI;
  (void)upW;
  (void)upS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % C::WM, wn = wave / C::WM;
  const int tile = blockIdx.x, n = blockIdx.y, cb = blockIdx.z;
  const int ty0 = (tile / tiles_x) * C::TH, tx0 = (tile % tiles_x) * C::TW;
  const int H = p.H, W = p.W;
  const ActCtx cx0 = make_ctx(p.a0);
  f32x4 acc[C::MI][C::NI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m)
#pragma unroll
    for (int j = 0; j < C::NI; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kg = lane >> 4;
  int aoff[C::MI];
#pragma unroll
  for (int m = 0; m < C::MI; ++m) {
    const int pxl = (wm * C::MI + m) * 16 + (lane & 15);
    aoff[m] = (kg * 2) * C::PLANE + ((pxl / C::TW) * C::RS + (pxl % C::TW)) * 16;
  }
  const int g8 = (tid % C::NG) * 8;
  const int cin_total = p.a0.C + p.a1.C;
  const int nchunks = (cin_total + C::KC - 1) / C::KC;
  const int ntn = p.CoutPad / 16;
  const int nt0 = (cb * C::WN + wn) * C::NI;
  const bf16x8* wpk = reinterpret_cast<const bf16x8*>(p.wpk);
  Tab tab;
  if constexpr (TABK) {
    fill_tables_lds<KIND>(p.a0, ldsBN, tid, 256);
    __syncthreads();
  }
  if constexpr (UPB) {      // tap lists of this tile's rows and columns: HPFG_UPB_TAPS entries each, unused ones repeat the last index with weight 0
    if (tid < C::TH + C::TW) {
      int idx[8], cnt;
      float wgt[8];
      const bool isrow = tid < C::TH;
      const int L = isrow ? H : W;
      const int lo = clampi(isrow ? ty0 + tid : tx0 + tid - C::TH, 0, L - 1);
      hpfg_up_taps(lo, L, idx, wgt, cnt);
      cnt = cnt > HPFG_UPB_TAPS ? HPFG_UPB_TAPS : cnt;
      for (int k = 0; k < HPFG_UPB_TAPS; ++k) {
        upI[tid][k] = (short)(k < cnt ? idx[k] : idx[cnt - 1]);
        upW[tid][k] = k < cnt ? wgt[k] : 0.f;
      }
    }
    __syncthreads();
  }
  const bool side = UPB && cb == 0;      // (workgroup-uniform)
  const int wg_row = n * (tiles_x * tiles_y) + tile;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * C::KC + g8;
    const bool chv = c0 < cin_total;
    if constexpr (TABK) load_tables_lds<KIND>(tab, ldsBN, p.a0, chv ? c0 : 0);
    else if constexpr (!UPB) load_tables<KIND>(tab, p.a0, c0, chv);
    __syncthreads();
    if constexpr (UPB) {
      const int c0c = chv ? c0 : 0, ps = p.a0.pstride, Wo = 2 * W;
      const float* src = p.a0.z + (long)n * (2 * H) * Wo * ps + c0c;
      f32x4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0;
#pragma unroll
      for (int i = 0; i < C::NLD; ++i) {
        const Piece q = make_piece<C>(tid, i);
        if (!q.ok) continue;
        const int gy = ty0 + q.ly, gx = tx0 + q.lx;
        const bool ok = chv && gy < H && gx < W;
        int ix[HPFG_UPB_TAPS];
        float wx[HPFG_UPB_TAPS];
#pragma unroll
        for (int b = 0; b < HPFG_UPB_TAPS; ++b) {
          ix[b] = upI[C::TH + q.lx][b] * ps;
          wx[b] = hpfg_own_vgpr(upW[C::TH + q.lx][b]);
        }
        f32x4 v[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {          // the two channel quads of the piece: 25 loads in flight each
          f32x4 t[HPFG_UPB_TAPS][HPFG_UPB_TAPS];
#pragma unroll
          for (int a = 0; a < HPFG_UPB_TAPS; ++a) {
            const float* row = src + (long)upI[q.ly][a] * Wo * ps + 4 * hq;
#pragma unroll
            for (int b = 0; b < HPFG_UPB_TAPS; ++b) t[a][b] = *reinterpret_cast<const f32x4*>(row + ix[b]);
          }
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int a = 0; a < HPFG_UPB_TAPS; ++a) {      // separable, in hpfg_upsample2x_bwd's order: a row's horizontal taps, then its vertical weight
            f32x4 h = wx[0] * t[a][0];
#pragma unroll
            for (int b = 1; b < HPFG_UPB_TAPS; ++b) h += wx[b] * t[a][b];
            acc += hpfg_own_vgpr(upW[q.ly][a]) * h;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) v[hq][j] = ok ? acc[j] : 0.f;
        }
        store_piece<C, HPFG_KIND_PLAIN>(lds, q, v[0], v[1]);
        if (side && ok) {
          if (p.stage_out) {
            float* d = p.stage_out + ((long)(n * H + gy) * W + gx) * p.a0.C + c0c;
            *reinterpret_cast<f32x4*>(d) = v[0];
            *reinterpret_cast<f32x4*>(d + 4) = v[1];
          }
          cs0 += v[0];
          cs1 += v[1];
        }
      }
      if (side && p.side_sums) {
        *reinterpret_cast<f32x4*>(upS + tid * 8) = cs0;
        *reinterpret_cast<f32x4*>(upS + tid * 8 + 4) = cs1;
      }
    } else {
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
      const Piece q = make_piece<C>(tid, i);
      const int gy = ty0 + q.ly, gx = tx0 + q.lx;
      const bool ok = q.ok && chv && gy < H && gx < W;
      RawPiece<KIND> raw;
      f32x4 v0, v1;
      const int gyc = clampi(gy, 0, H - 1), gxc = clampi(gx, 0, W - 1), c0c = chv ? c0 : 0;
      issue_piece<KIND>(raw, p.a0, p.a1, cx0, n, gyc, gxc, c0c, ok);
      finish_piece<KIND>(v0, v1, raw, tab, p.a0, p.a1, cx0, n, gyc, gxc, c0c, ok);
      store_piece<C, KIND>(lds, q, v0, v1);
    }
    }
    bf16x8 bh[C::NI], bl[C::NI];
    load_b<C>(bh, bl, wpk, ch, ntn, nt0, lane);
    __syncthreads();
    if constexpr (UPB) {
      if (side && p.side_sums && tid < C::KC) {      // channel tid of the chunk: the threads of its 8-channel group, in a fixed order
        const int g = tid >> 3, j = tid & 7;
        float tsum = 0.f;
        for (int k = g; k < 256; k += C::NG) tsum += upS[k * 8 + j];
        const int cch = ch * C::KC + tid;
        if (cch < p.a0.C) p.side_sums[(long)wg_row * p.a0.C + cch] = tsum;
      }
    }
#pragma unroll
    for (int m = 0; m < C::MI; ++m) {
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(lds + aoff[m]);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(lds + aoff[m] + C::PLANE);
#pragma unroll
      for (int j = 0; j < C::NI; ++j) { HPFG16_MFMA3(acc[m][j], ah, al, bh[j], bl[j]) }
    }
  }
  f32x4 s1[C::NI], s2[C::NI];
#pragma unroll
  for (int j = 0; j < C::NI; ++j) {
    s1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[j] = s1[j];
  }
  f32x4 bias[C::NI];       // per-workgroup constant: fetched once, not per tile in the epilogue
  conv16_load_bias<C>(p, bias, lane, nt0);
  conv16_store_tile<C, BWD>(p, acc, s1, s2, bias, lane, wm, nt0, n, ty0, tx0);
  __syncthreads();
  conv16_flush_stats<C, BWD>(p, s1, s2, ldsf, tid, lane, wm, wn, cb, n * (tiles_x * tiles_y) + tile);
}

// number of stat_partials rows the kernel for this configuration writes (persistent 3x3: one per workgroup)
template <class C, int KIND>
int persistent_grid(const HpfgConvArgs& a) {
  const int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  const int lds_bytes = 2 * C::BUF_BYTES + 2 * 4 * C::BN * 4;
  int per_cu = 160 * 1024 / lds_bytes;
  const int reg_cap = wg_per_cu<C, KIND>();   // matches __launch_bounds__ (waves per SIMD)
  if (per_cu > reg_cap) per_cu = reg_cap;
  if (per_cu < 1) per_cu = 1;
  // All workgroups (grid.x * grid.y, grid.y = output-channel slices) must be resident at once -- a second round of workgroups
  // would start cold behind the first -- and should carry the same number of tiles: spread the tiles over as many workgroups as
  // ceil(tiles / capacity) rounds need, not over the whole capacity.
  const long nwork = (long)tx * ty * a.N;
  const long gy = a.CoutPad / C::BN;
  long cap = 256L * per_cu / (gy > 0 ? gy : 1);
  if (cap < 1) cap = 1;
  const long rounds = (nwork + cap - 1) / cap;
  return (int)((nwork + rounds - 1) / rounds);
}

template <class C, int KIND>
int launch_cfg(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  int tx = (a.W + C::TW - 1) / C::TW, ty = (a.H + C::TH - 1) / C::TH;
  if (rows_only) {
    if constexpr (C::TAPS == 9) *rows_only = persistent_grid<C, KIND>(a);
    else *rows_only = tx * ty * a.N;
    return 0;
  }
  if constexpr (C::TAPS == 9) {
    dim3 grid((unsigned)persistent_grid<C, KIND>(a), a.CoutPad / C::BN);
    if (a.stat_acc) HPFG_ACC_CHECK((long)grid.x * grid.y, a.stat_shards, "conv_fwd");
    if constexpr (KIND == HPFG_KIND_DZ || KIND == HPFG_KIND_PLAIN) {
      if constexpr (C::TH == 4) {      // (the max-pool-backward epilogue: an instantiation of its own, for the sizes that are not multiples of 16)
        if (a.bwd_stats == 2) {
          hipLaunchKernelGGL((conv_bf16x3_kernel<C, KIND, 2>), grid, dim3(256), 0, st, a, tx, ty);
          return hpfg_launch_status("conv_bf16x3_kernel<bwd stats + pool>");
        }
      }
      if (a.bwd_stats == 2) {
        hpfg_set_error("conv_fwd: bwd_stats == 2 is built for H or W not a multiple of 16");
        return -1;
      }
      if (a.bwd_stats) {
        hipLaunchKernelGGL((conv_bf16x3_kernel<C, KIND, 1>), grid, dim3(256), 0, st, a, tx, ty);
        return hpfg_launch_status("conv_bf16x3_kernel<bwd stats>");
      }
    }
    hipLaunchKernelGGL((conv_bf16x3_kernel<C, KIND>), grid, dim3(256), 0, st, a, tx, ty);
  } else {
    dim3 grid(tx * ty, a.N, a.CoutPad / C::BN);
    if (a.stat_acc) HPFG_ACC_CHECK((long)grid.x * grid.y * grid.z, a.stat_shards, "conv_fwd(1x1)");
    if constexpr (KIND == HPFG_KIND_DZ || KIND == HPFG_KIND_PLAIN || KIND == HPFG_KIND_UPB) {
      if (a.bwd_stats) {
        hipLaunchKernelGGL((conv1x1_bf16x3_kernel<C, KIND, 1>), grid, dim3(256), 0, st, a, tx, ty);
        return hpfg_launch_status("conv1x1_bf16x3_kernel<bwd stats>");
      }
    }
    hipLaunchKernelGGL((conv1x1_bf16x3_kernel<C, KIND>), grid, dim3(256), 0, st, a, tx, ty);
  }
  return hpfg_launch_status("conv_bf16x3_kernel");
}

template <int KIND, int TAPS>
int conv_dispatch_kind(const HpfgConvArgs& a, hipStream_t st, int* rows_only) {
  const bool big = (a.H % 16 == 0) && (a.W % 16 == 0);
  const int cp = a.CoutPad;
  constexpr int KCB = TAPS == 9 ? 16 : 32;
  if (big) {
    if constexpr (TAPS == 1) {
      if (cp % 64 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 4, TAPS, KCB>, KIND>(a, st, rows_only);
    }
    if (cp % 32 == 0) return launch_cfg<Cfg<16, 16, 4, 1, 2, TAPS, KCB>, KIND>(a, st, rows_only);   // 3x3: 32-channel slices (B ring = 80 VGPRs)
    return launch_cfg<Cfg<16, 16, 4, 1, 1, TAPS, KCB>, KIND>(a, st, rows_only);
  }
  if constexpr (TAPS == 9) {
    // 3x3 on sizes that are not multiples of 16 (56, 28, 14, ...): 4x16-pixel tiles.  Same 64 pixels per workgroup as an 8x8 tile
    // but the LDS image needs no row padding (14 KB instead of 31 KB per buffer).  Output-channel slice per workgroup: 64 where
    // it divides (one wave per 16 channels, 4 pixel tiles per wave: every staged input tile and every B fragment feeds 4x the
    // MFMAs of the 32-wide slice, and the B ring can be a whole chunk deep) -- measured 21 vs 29 us on 128->128 @28, 36 vs 50 us
    // on 256->128 @28; a 128-wide slice (two channel tiles per wave, shallow B ring) is slower again.
    // (round 4: 8-wave workgroups for the <= 256-workgroup layers -- two k-groups per staged chunk, or 128-wide slices -- sped those launches up
    // by 10-25 % and the step down by 2 %: such a workgroup owns its CU and the other stream's kernels lose their share; profiles/r04_conv8_lever_b.txt)
    // (round 4: the same 64 x 64 tile as 2 x 2 waves of 2 pixel tiles x 2 channel tiles -- half the LDS reads of A fragments, which the SQ
    // counters make the longest of the kernel's three pipes, but a B ring of 3 instead of 9 k-steps -- measured +7 % on the step,
    // profiles/r04_schedule_experiments.txt)
    // (round 5) a 64-wide slice leaves a launch of <= 128 workgroups on half of the 256 CUs at one wave per SIMD -- the dgrad of down4.c1,
    // 256 -> 128 channels on 16 x 14 x 14 pixels with the max-pool backward in its epilogue, took 44 - 73 us that way: 32-wide slices there
    // (twice the workgroups; the staging is repeated per slice, but such a launch is latency-bound, not issue-bound)
    {
      const long tiles = (long)a.N * ((a.H + 3) / 4) * ((a.W + 15) / 16);
      if (cp % 64 == 0 && tiles * (cp / 64) <= hpfg_opt(HPFG_OPT_NARROW_DEEP))      // (the option's value = the workgroup threshold; 0 = off)
        return launch_cfg<Cfg<4, 16, 2, 2, 1, TAPS, 32>, KIND>(a, st, rows_only);
    }
    if (cp % 64 == 0) return launch_cfg<Cfg<4, 16, 1, 4, 1, TAPS, 32>, KIND>(a, st, rows_only);
    if (cp % 32 == 0) return launch_cfg<Cfg<4, 16, 2, 2, 1, TAPS, 32>, KIND>(a, st, rows_only);
    return launch_cfg<Cfg<4, 16, 4, 1, 1, TAPS, 32>, KIND>(a, st, rows_only);
  }
  // 1x1 on small spatial sizes (8x8 tiles): narrow the output-channel slice per workgroup until there are >= 2 workgroups per CU
  const long nwork = (long)a.N * ((a.H + 7) / 8) * ((a.W + 7) / 8);
  if (cp % 128 == 0 && nwork * (cp / 128) >= 512) return launch_cfg<Cfg<8, 8, 1, 4, 2, TAPS, 32>, KIND>(a, st, rows_only);
  if (cp % 64 == 0 && (nwork * (cp / 64) >= 512 || cp % 32 != 0)) return launch_cfg<Cfg<8, 8, 1, 4, 1, TAPS, 32>, KIND>(a, st, rows_only);
  if (cp % 32 == 0) return launch_cfg<Cfg<8, 8, 2, 2, 1, TAPS, 32>, KIND>(a, st, rows_only);
  return launch_cfg<Cfg<8, 8, 4, 1, 1, TAPS, 32>, KIND>(a, st, rows_only);
}

}  // namespace hpfg_conv16

constexpr int HPFG_THIN_NONE = -100;
int hpfg_conv_thin_try(const HpfgConvArgs& a, hipStream_t st, int* rows_only);      // conv_thin.hip
int hpfg_conv16_launch_plain(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
int hpfg_conv16_launch_bnact(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
int hpfg_conv16_launch_pool(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
int hpfg_conv16_launch_cat(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
int hpfg_conv16_launch_dz(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
int hpfg_conv16_launch_upb(const HpfgConvArgs& a, hipStream_t st, int* rows_only);
