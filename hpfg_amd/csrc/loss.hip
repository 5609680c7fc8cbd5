// Fused per-pixel segmentation losses on NHWC logits: softmax + cross-entropy(ignore 255) + Dice + MSE consistency.
//
// Reference arithmetic:
//   CE    nn.CrossEntropyLoss(ignore_index=255): mean over non-ignored pixels of -log softmax[target]
//         (main.py:92,164,167; utils/loss/medloss.py:49)
//   Dice  utils/loss/diceloss.py:155-191: one-hot by equality (ignored pixels match no class but still count in sum p^2),
//         per class 1 - (2*sum(p*t)+1e-5)/(sum(p*p)+sum(t*t)+1e-5) with sums over the whole batch, mean over classes
//   MSE   mean((softmax(student) - softmax(teacher))^2) over the unlabelled images (main.py:191, Mean-Teacher :104)
//   total = ce0*CE0 + dice0*Dice0 + ce1*CE1 + dice1*Dice1 + mse_w*MSE, two label groups: images [0,n_lab) with labels0,
//           images [n_lab,N) with labels1 (pseudo-labels; CPS :108-110, HPFG main.py:176-180).
// Dice is a ratio of batch-wide sums, so the op is two-phase: (1) partial sums -> sums (all-reduce them across ranks for
// data parallel), finalize -> scalars; (2) backward re-reads the logits and writes dlogits from the global sums.
// Nothing here synchronises with the host; the reference's per-class dice.item() (diceloss.py:189) has no counterpart.
#include <string.h>
#include "common.h"
#include "peer.h"

namespace {

constexpr int NS = HPFG_LOSS_NSUM;
// sums layout: 0 nll0, 1 cnt0, 2 nll1, 3 cnt1, 4 mse (masked: sum mask*d^2), 5 sum mask, 8+c I0, 12+c Z0, 16+c Y0, 20+c I1, 24+c Z1, 28+c Y1   (C <= 4)
constexpr int PIX_PER_BLOCK = 1024;
constexpr float SMOOTH = 1e-5f;

template <int C>
__device__ inline void softmax_c(const float* l, float* p) {
  float m = l[0];
#pragma unroll
  for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    p[c] = expf(l[c] - m);
    s += p[c];
  }
  float inv = 1.f / s;
#pragma unroll
  for (int c = 0; c < C; ++c) p[c] *= inv;
}

template <int C>
__device__ inline void load_px(const float* base, long pix, float* l) {
  if (C == 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(base + pix * 4);
    l[0] = v[0]; l[1] = v[1]; l[2] = v[2]; l[3] = v[3];
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) l[c] = base[pix * C + c];
  }
}

// All accumulators are named scalars / statically indexed arrays (runtime-indexed register arrays would spill to scratch).
template <int C>
__global__ __launch_bounds__(256) void loss_partials_kernel(HpfgLossArgs a, long npix_img) {
  __shared__ float red[4][NS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float nll[2] = {0.f, 0.f}, cnt[2] = {0.f, 0.f}, mse = 0.f, msk = 0.f;
  float I[2][4], Z[2][4], Y[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int c = 0; c < 4; ++c) I[g][c] = Z[g][c] = Y[g][c] = 0.f;
  const long total = (long)a.N * npix_img;
  const long p0 = (long)blockIdx.x * PIX_PER_BLOCK;
  for (long pix = p0 + tid; pix < p0 + PIX_PER_BLOCK && pix < total; pix += 256) {
    const int n = (int)(pix / npix_img);
    float l[C], p[C];
    load_px<C>(a.logits, pix, l);
    if (a.input_is_prob) {
#pragma unroll
      for (int c = 0; c < C; ++c) p[c] = l[c];
    } else {
      softmax_c<C>(l, p);
    }
    const int grp = n < a.n_lab ? 0 : 1;
    const uint8_t* lab = grp == 0 ? a.labels0 : a.labels1;
    if (lab) {
      const int t = grp == 0 ? lab[pix] : lab[pix - (long)a.n_lab * npix_img];
      float m = l[0];
#pragma unroll
      for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
      float se = 0.f, lt = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        se += expf(l[c] - m);
        lt = (t == c) ? l[c] : lt;
      }
      const bool valid = t != 255 && t < C;
      const float nl = valid ? (m + logf(se)) - lt : 0.f;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const bool on = g == grp;
        nll[g] += on ? nl : 0.f;
        cnt[g] += (on && valid) ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float tc = (t == c) ? 1.f : 0.f;
          I[g][c] += on ? p[c] * tc : 0.f;
          Z[g][c] += on ? p[c] * p[c] : 0.f;
          Y[g][c] += on ? tc : 0.f;
        }
      }
    }
    if (grp == 1 && a.t_logits) {
      float q[C], tl[C];
      const long upix = pix - (long)a.n_lab * npix_img;
      load_px<C>(a.t_logits, a.t_unlab_only ? upix : pix, tl);
      if (a.teacher_is_prob) {
#pragma unroll
        for (int c = 0; c < C; ++c) q[c] = tl[c];
      } else {
        softmax_c<C>(tl, q);
      }
      const float w = a.cons_mask ? a.cons_mask[upix] : 1.f;
      float dd = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float d = p[c] - q[c];
        dd += d * d;
      }
      mse += w * dd;
      msk += w;
    }
  }
  float s[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) s[i] = 0.f;
  s[0] = nll[0]; s[1] = cnt[0]; s[2] = nll[1]; s[3] = cnt[1]; s[4] = mse; s[5] = msk;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    s[8 + c] = I[0][c]; s[12 + c] = Z[0][c]; s[16 + c] = Y[0][c];
    s[20 + c] = I[1][c]; s[24 + c] = Z[1][c]; s[28 + c] = Y[1][c];
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    float v = s[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (tid < NS) a.partials[(long)blockIdx.x * NS + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ sums, HpfgPeerX px) {
  __shared__ double sh[4];
  const int i = blockIdx.x;
  double v = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) v += (double)partials[(long)b * NS + i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[1] = {sh[0] + sh[1] + sh[2] + sh[3]};
    if (px.world > 1) {          // data parallel, global-batch mode: the ranks' sums of this term (peer mailbox, rank order)
      const int idx[1] = {i};
      hpfg_peer_allreduce<1>(px, idx, t);
    }
    sums[i] = (float)t[0];
  }
}

__device__ inline float dice_of(const float* s, int base, int C) {
  float d = 0.f;
  for (int c = 0; c < C; ++c) d += 1.f - (2.f * s[base + c] + SMOOTH) / (s[base + 4 + c] + s[base + 8 + c] + SMOOTH);
  return d / (float)C;
}

__global__ void loss_finalize_kernel(HpfgLossArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float* s = a.sums;
  const int C = a.C;
  float ce0 = s[1] > 0.f ? s[0] / s[1] : 0.f;
  float ce1 = s[3] > 0.f ? s[2] / s[3] : 0.f;
  float d0 = a.labels0 && a.n_lab > 0 ? dice_of(s, 8, C) : 0.f;
  float d1 = a.labels1 && a.n_lab < a.N ? dice_of(s, 20, C) : 0.f;
  float cnt = (float)((double)(a.N - a.n_lab) * a.H * a.W * C * a.world);
  float mse = (a.t_logits && cnt > 0.f) ? (a.cons_mask ? s[4] / (2.f * s[5] + 1e-16f) : s[4] / cnt) : 0.f;
  const float* k = a.coef;
  a.out[0] = k[0] * ce0 + k[1] * d0 + k[2] * ce1 + k[3] * d1 + k[4] * mse;
  a.out[1] = ce0;
  a.out[2] = d0;
  a.out[3] = ce1;
  a.out[4] = d1;
  a.out[5] = mse;
  a.out[6] = 0.f;
  a.out[7] = 0.f;
}

template <int C>
__global__ __launch_bounds__(256) void loss_bwd_kernel(HpfgLossArgs a, long npix_img, const float* __restrict__ gscale_dev) {
  const float gscale = gscale_dev ? *gscale_dev : 1.f;
  const float* s = a.sums;
  const float* k = a.coef;
  const long total = (long)a.N * npix_img;
  const float cnt_mse = (float)((double)(a.N - a.n_lab) * a.H * a.W * C * a.world);
  // per-group, per-class Dice derivative coefficients: dDice/dp_c = A_c * t_c + B_c * p_c
  float dA[2][C], dB[2][C], wce[2], cntv[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int base = g == 0 ? 8 : 20;
    const float wd = k[2 * g + 1] / (float)C;
    wce[g] = k[2 * g];
    cntv[g] = s[2 * g + 1];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float den = s[base + 4 + c] + s[base + 8 + c] + SMOOTH;
      float num = 2.f * s[base + c] + SMOOTH;
      dA[g][c] = -2.f * wd / den;
      dB[g][c] = 2.f * wd * num / (den * den);
    }
  }
  const float wm = cnt_mse > 0.f ? k[4] * 2.f / (a.cons_mask ? 2.f * s[5] + 1e-16f : cnt_mse) : 0.f;
  for (long pix = blockIdx.x * 256L + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int n = (int)(pix / npix_img);
    float l[C], p[C], dp[C];
    load_px<C>(a.logits, pix, l);
    if (a.input_is_prob) {
#pragma unroll
      for (int c = 0; c < C; ++c) p[c] = l[c];
    } else {
      softmax_c<C>(l, p);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dp[c] = 0.f;
    const int grp = n < a.n_lab ? 0 : 1;
    const uint8_t* lab = grp == 0 ? a.labels0 : a.labels1;
    int t = 255;
    if (lab) {
      t = grp == 0 ? lab[pix] : lab[pix - (long)a.n_lab * npix_img];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float tc = (t == c) ? 1.f : 0.f;
        float ca = grp == 0 ? dA[0][c] : dA[1][c], cb = grp == 0 ? dB[0][c] : dB[1][c];
        dp[c] += ca * tc + cb * p[c];
      }
    }
    if (grp == 1 && a.t_logits) {
      float q[C], tl[C];
      const long upix = pix - (long)a.n_lab * npix_img;
      load_px<C>(a.t_logits, a.t_unlab_only ? upix : pix, tl);
      if (a.teacher_is_prob) {
#pragma unroll
        for (int c = 0; c < C; ++c) q[c] = tl[c];
      } else {
        softmax_c<C>(tl, q);
      }
      const float w = a.cons_mask ? wm * a.cons_mask[upix] : wm;
#pragma unroll
      for (int c = 0; c < C; ++c) dp[c] += w * (p[c] - q[c]);
    }
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) dot += p[c] * dp[c];
    const float cn = grp == 0 ? cntv[0] : cntv[1];
    const float wc = grp == 0 ? wce[0] : wce[1];
    const bool ce_on = lab && t != 255 && t < C && cn > 0.f;
    float g[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      g[c] = a.input_is_prob ? dp[c] : p[c] * (dp[c] - dot);
      if (ce_on && !a.input_is_prob) g[c] += wc * (p[c] - (t == c ? 1.f : 0.f)) / cn;
      g[c] *= gscale;
    }
    if (C == 4) {
      *reinterpret_cast<f32x4*>(a.dlogits + pix * 4) = f32x4{g[0], g[1], g[2], g[3]};
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) a.dlogits[pix * C + c] = g[c];
    }
  }
}

}  // namespace

extern "C" int hpfg_loss_blocks(int N, int H, int W) { return (int)(((long)N * H * W + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK); }

static int check_loss(const HpfgLossArgs* a) {
  HPFG_ARG_CHECK(a && a->logits && a->coef && a->sums, "seg_loss: null pointer");
  HPFG_ARG_CHECK(a->C >= 2 && a->C <= 4, "seg_loss: C must be 2..4 (got %d)", a->C);
  HPFG_ARG_CHECK(a->N > 0 && a->n_lab >= 0 && a->n_lab <= a->N && a->H > 0 && a->W > 0 && a->world >= 1, "seg_loss: bad sizes");
  HPFG_ARG_CHECK(a->n_lab == 0 || a->labels0, "seg_loss: labels0 missing");
  return 0;
}

static int loss_partials_impl(const HpfgLossArgs* a, const HpfgPeerX* px, void* stream) {
  if (int rc = check_loss(a)) return rc;
  HPFG_ARG_CHECK(a->partials, "seg_loss_partials: workspace missing");
  int nblk = hpfg_loss_blocks(a->N, a->H, a->W);
  if (a->C == 4) hipLaunchKernelGGL(loss_partials_kernel<4>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W);
  else if (a->C == 3) hipLaunchKernelGGL(loss_partials_kernel<3>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W);
  else hipLaunchKernelGGL(loss_partials_kernel<2>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W);
  HpfgPeerX none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(NS), dim3(256), 0, (hipStream_t)stream, a->partials, nblk, a->sums, px ? *px : none);
  return hpfg_launch_status("loss_partials_kernel");
}

extern "C" int hpfg_seg_loss_partials(const HpfgLossArgs* a, void* stream) { return loss_partials_impl(a, nullptr, stream); }

extern "C" int hpfg_seg_loss_partials_x(const HpfgLossArgs* a, const HpfgPeerX* px, void* stream) {
  HPFG_ARG_CHECK(px, "seg_loss_partials_x: null exchange descriptor");
  if (px->world > 1) {
    HPFG_ARG_CHECK(px->world <= HPFG_PEER_MAX_RANKS && px->rank >= 0 && px->rank < px->world && px->epoch && px->slot >= 0 && NS <= px->cap &&
                       px->slot_bytes == hpfg_peer_slot_bytes(px->world, px->cap),
                   "seg_loss_partials_x: bad exchange descriptor");
    for (int r = 0; r < px->world; ++r) HPFG_ARG_CHECK(px->mbox[r], "seg_loss_partials_x: mailbox of rank %d not mapped", r);
  }
  return loss_partials_impl(a, px, stream);
}

extern "C" int hpfg_seg_loss_finalize(const HpfgLossArgs* a, void* stream) {
  if (int rc = check_loss(a)) return rc;
  HPFG_ARG_CHECK(a->out, "seg_loss_finalize: out missing");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);
  return hpfg_launch_status("loss_finalize_kernel");
}

extern "C" int hpfg_seg_loss_bwd(const HpfgLossArgs* a, const float* grad_scale_dev, void* stream) {
  if (int rc = check_loss(a)) return rc;
  HPFG_ARG_CHECK(a->dlogits, "seg_loss_bwd: dlogits missing");
  long total = (long)a->N * a->H * a->W;
  long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (a->C == 4) hipLaunchKernelGGL(loss_bwd_kernel<4>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W, grad_scale_dev);
  else if (a->C == 3) hipLaunchKernelGGL(loss_bwd_kernel<3>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W, grad_scale_dev);
  else hipLaunchKernelGGL(loss_bwd_kernel<2>, dim3((int)b), dim3(256), 0, (hipStream_t)stream, *a, (long)a->H * a->W, grad_scale_dev);
  return hpfg_launch_status("loss_bwd_kernel");
}
