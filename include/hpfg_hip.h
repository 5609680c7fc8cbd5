/*
 * hpfg_hip.h -- C ABI of libhpfg_hip.so: the MI355X (gfx950) kernels behind HPFG's training hot path.
 *
 * The reference (fakerlove1/HPFG) is pure Python/PyTorch: its "FFI" for this path is the set of torch ops its
 * U-Net and losses issue.  Each entry point below replaces one fused family of those ops
 * and cites the reference lines whose arithmetic it reproduces.  Conventions:
 *   - activations are fp32 NHWC on the device; all pointers are device pointers unless marked host;
 *   - every function only enqueues work on `stream` (a hipStream_t passed as void*): no allocation, no
 *     synchronisation, no host<->device copies, safe to capture into a hipGraph;
 *   - return value 0 = ok, <0 = argument error (see hpfg_last_error()), >0 = hipError_t from the launch.
 *
 * "Virtual activations": train-mode BatchNorm makes conv -> BN -> LeakyReLU -> Dropout -> (MaxPool | Upsample+concat)
 * a chain of per-element maps once the batch statistics are known, so consumers apply that chain while they load
 * the producer's raw (pre-BN) conv output instead of reading a materialised activation.  HpfgAct describes such a
 * source; the same descriptor describes the backward-side virtual tensor dZ (gradient w.r.t. a raw conv output).
 */
#ifndef HPFG_HIP_H
#define HPFG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPFG_VERSION 132
enum { HPFG_MATH_F32 = 0, HPFG_MATH_BF16X3 = 1 };

/* rows of a per-layer BatchNorm table `bn` ([HPFG_BN_ROWS][C] floats) */
enum { HPFG_BN_MEAN = 0, HPFG_BN_RSTD = 1, HPFG_BN_SCALE = 2, HPFG_BN_SHIFT = 3,
       HPFG_BN_K1 = 4, HPFG_BN_K2 = 5, HPFG_BN_K3 = 6, HPFG_BN_SPARE = 7, HPFG_BN_ROWS = 8 };

enum {
  HPFG_ACT_NONE = 0,       /* contributes no channels */
  HPFG_ACT_PLAIN = 1,      /* v = z[n,y,x,c]                       (NHWC, float4 path, C%4==0)               */
  HPFG_ACT_STRIDED = 2,    /* v = z[n*sn + c*sc + y*sy + x*sx]     (any layout, scalar path; network input)  */
  HPFG_ACT_BNACT = 3,      /* v = drop(lrelu(z*scale+shift))       unet.py:19-21,23-24 (BN train, LeakyReLU .01, Dropout) */
  HPFG_ACT_BNACT_POOL = 4, /* v = max 2x2 of lrelu(z*scale+shift)  unet.py:37 (MaxPool2d(2)) fused on load   */
  HPFG_ACT_UP2X = 5,       /* v = bilinear x2, align_corners=True  unet.py:51,56 fused on load               */
  HPFG_ACT_DZ = 6,         /* v = k1*g + k2*z + k3, g = aux*dropmask*lrelu'(z*scale+shift): BN/LeakyReLU/Dropout backward */
  HPFG_ACT_SPLIT16 = 7,    /* a side tensor stored by HpfgConvArgs.stage_out (round 5): the value ALREADY split for the bf16x3 matrix-core
                              products, v = hi + lo with hi = bf16(v), lo = bf16(v - hi): z = bf16 [N][Hs][Ws][pstride / 8][hi 8 | lo 8]
                              (pstride = channels per pixel, C % 8 == 0: 32 contiguous bytes per 8 channels, the size of the fp32 tensor).
                              Read by hpfg_wgrad only: its loader becomes a copy into LDS. */
  HPFG_ACT_UPBWD = 8       /* (round 5, 1x1 dgrad only) the gradient w.r.t. a tensor BEFORE nn.Upsample(x2, bilinear, align_corners=True) (unet.py:51),
                              gathered on load from the gradient w.r.t. the upsampled tensor: z = that gradient [N][2*Hs][2*Ws] with pstride
                              floats per pixel, C channels; the value at (n, y, x) is the transposed interpolation -- what
                              hpfg_upsample2x_bwd writes into a tensor of its own */
};

typedef struct HpfgAct {
  const float* z;      /* raw tensor */
  const float* bn;     /* BN table [HPFG_BN_ROWS][bn_stride] for BNACT/BNACT_POOL/DZ, else NULL */
  const float* aux;    /* DZ: upstream gradient w.r.t. the activated output (dA) */
  int32_t mode;
  int32_t C;           /* channels this source provides (virtual channels >= C read as 0) */
  int32_t Hs, Ws;      /* spatial size of z (POOL: 2x the virtual size, UP2X: half of it) */
  int32_t pstride;     /* floats between pixels of z (PLAIN/BNACT/POOL/UP2X/DZ) */
  int32_t aux_pstride; /* floats between pixels of aux (DZ) */
  int32_t bn_stride;   /* row length of bn (= channel count of the producing layer) */
  int32_t bn_coff;     /* first channel of this source inside the bn rows (normally 0) */
  int32_t sn, sc, sy, sx; /* STRIDED element strides */
  float drop_p;        /* dropout probability applied to this activation (BNACT, DZ); 0 = none */
  uint32_t drop_seed;  /* counter-based RNG key; keep(i) = hash(i, seed) >= p*2^32, i = NHWC element index */
  const uint8_t* drop_mask; /* optional explicit keep-mask (bytes, NHWC element order) used instead of the RNG: lets parity
                               tests replay the masks torch drew in the reference run */
  const uint32_t* seed_dev; /* optional device word added to drop_seed at run time (lets a captured hipGraph draw new masks per replay) */
  /* BatchNorm statistics straight from the producer's sum accumulators (round 4; BNACT / BNACT_POOL sources of the bf16x3 forward kernels):
     with bn_acc != NULL the consumer derives scale / shift of its input channels itself in its prologue -- mean = S1 / count,
     var = S2 / count - mean^2, rstd = 1 / sqrt(var + eps), scale = gamma * rstd, shift = beta - mean * scale (fp64, the arithmetic of
     hpfg_bn_fwd_finalize) -- instead of reading the rows of `bn`, so that no finalize launch sits between a conv and its consumer
     (unet.py:18-24: conv -> BatchNorm -> LeakyReLU).  `bn` is then written once per forward by hpfg_bn_acc_finalize for the backward kernels. */
  const long long* bn_acc;  /* accumulators of the producing layer, see HpfgConvArgs.stat_acc; NULL = read the table rows */
  const float* bn_gamma;    /* BatchNorm weight / bias of the producing layer ([bn_stride]) */
  const float* bn_beta;
  float bn_count;           /* N * H * W of the producing layer (elements per channel) */
  float bn_eps;
  int32_t bn_shards;        /* shards of bn_acc (the producer's stat_shards) */
  int32_t reserved1;
} HpfgAct;

typedef struct HpfgConvArgs {
  HpfgAct a0, a1;       /* input = channel concat [a0 | a1]  (torch.cat([skip, up]), unet.py:57); (a0.C+a1.C) padded to 16 */
  const float* wpk;     /* weights in MFMA fragment order, see hpfg_pack_weights */
  const float* bias;    /* [CoutPad] or NULL */
  float* out;           /* raw output, NHWC, out_pstride floats per pixel */
  float* stat_partials; /* NULL or [hpfg_conv_stat_blocks()][2][CoutPad]: per-block sum(z), sum(z*z) for BN (unet.py:19,23) */
  int32_t out_pstride, Cout, CoutPad;
  int32_t N, H, W;      /* output (= virtual input) size */
  int32_t taps;         /* 9 (3x3, pad 1) or 1 (1x1) */
  int32_t math;         /* HPFG_MATH_F32: exact fp32 MFMA, wpk = fp32 fragments; HPFG_MATH_BF16X3: split-bf16 MFMA (hi*hi+hi*lo+lo*hi,
                           fp32 accumulate), wpk = the wpk16_* buffer of hpfg_pack_weights */
  int32_t bwd_stats;    /* dgrad only (BF16X3, DZ or PLAIN source).  2 (3x3): `out` would be the gradient w.r.t. MaxPool2d(2) of bwd_of's activated
                           output (bwd_of at 2H x 2W, its gradient so far in bwd_of.aux, no dropout behind it): the epilogue adds the tile at the
                           arg-max of each 2 x 2 window into bwd_of.aux IN PLACE (unet.py:37 backward), accumulates the backward sums of the
                           completed gradient as below, and does not write `out` -- hpfg_bn_bwd_reduce_pool without its launch.
                           1 = `out` is the COMPLETE gradient w.r.t. the activated output of the
                           BatchNorm layer described by `bwd_of`; stat_partials then receives that layer's backward sums
                           sum(g), sum(g*xhat)  (g = out * LeakyReLU' * dropout) instead of sum(z), sum(z*z) -- what hpfg_bn_bwd_reduce
                           would compute in a pass of its own; feed the rows to hpfg_bn_bwd_finalize */
  HpfgAct bwd_of;       /* that layer: z, its table (mean, rstd, scale, shift rows), dropout fields; C == Cout, Hs x Ws == H x W */
  float* out2;          /* optional second destination: output channels >= out_split go to out2[pixel * out2_pstride + (co - out_split)]
                           instead of `out`.  The dgrad of a decoder block's first conv writes d(skip) and d(upsampled) -- the two halves of the
                           concat gradient, unet.py:57 -- into buffers of their own, so that neither is read through a half-used pixel stride */
  int32_t out_split;    /* multiple of 16; 0 = everything to `out` */
  int32_t out2_pstride;
  long long* stat_acc;  /* optional (bf16x3 kernels), instead of stat_partials: the per-channel sums are ADDED to this layer accumulator,
                           long long [HPFG_ACC_SHARDS][2 (sum z | sum z^2)][CoutPad][2 (limb)], by integer atomics.  A partial sum t is split
                           exactly as t = hi + lo * 2^-52 (hi = rint(t), lo = (t - hi) * 2^52): integer addition is associative, so the
                           totals are bit-reproducible whatever order the workgroups finish in.  Shard = workgroup id % stat_shards: more
                           shards = less same-address contention among the producer's workgroups (1 shard of a 768-workgroup launch: +40 us),
                           fewer = less to read in every consumer workgroup's prologue (32 bytes per channel and shard).
                           The accumulator must be zero when the launch starts (hpfg_pack_weights_bump zeroes a region). */
  int32_t stat_shards;  /* power of two, 1 .. HPFG_ACC_MAX_SHARDS */
  int32_t reserved2;
  float* stage_out;     /* optional, 3x3 layers on the bf16x3 kernels with a non-PLAIN source of (a0.C + a1.C) % 8 == 0 channels at sizes that are
                           not multiples of 16 (the channel-rich 56 / 28 / 14-pixel levels; the aligned ones have fused kernels): the staging
                           derives the virtual input of every pixel anyway (BatchNorm + LeakyReLU + Dropout [+ max-pool | + bilinear upsample and
                           concat] in a forward conv; dZ = k1*g + k2*z + k3 in a dgrad) -- the workgroups of output-channel slice 0 also store
                           it -- as the (hi | lo) bf16 pair the staging has in registers anyway (conv_bf16_kernel.h store_piece):
                           bf16 [N][H][W][(a0.C + a1.C) / 8][hi 8 | lo 8], the same 4 bytes per element an fp32 tensor takes.  The layer's weight
                           gradient reads them as HPFG_ACT_SPLIT16 sources (its input from the forward conv, its dZ from the dgrad): no
                           BatchNorm / LeakyReLU / Dropout chain, no fp32 -> bf16 split, in any of its (input-channel slice x output-channel
                           slice) workgroups -- its loader is a copy (autograd's saved tensors of nn.Conv2d, model/unet.py:18,22).
                           With an HPFG_ACT_UPBWD source (1x1 dgrad): fp32 [N][H][W][a0.C], the gathered gradient itself -- the dZ of the 1x1
                           conv's weight gradient -- stored by the workgroups of output-channel slice 0 */
  float* side_sums;     /* optional, HPFG_ACT_UPBWD source only: [hpfg_conv_stat_rows()][a0.C] per-workgroup channel sums of the gathered
                           gradient (slice 0), the rows of the 1x1 conv's bias gradient (summed in a fixed order by hpfg_slab_reduce_multi) */
} HpfgConvArgs;
#define HPFG_ACC_MAX_SHARDS 8
#define HPFG_ACC_WORDS(C, shards) ((shards) * 2 * 2 * (C))   /* long long words of one layer accumulator */

typedef struct HpfgWgradArgs {
  HpfgAct a0, a1;       /* conv input (as in forward) */
  HpfgAct g;            /* dZ: gradient w.r.t. the conv's raw output (DZ or PLAIN) */
  float* slab;          /* workspace [S][taps][CinPad][CoutPad] */
  float* dw_oihw;       /* result, PyTorch layout [Cout][Cin][k][k] (written, not accumulated) */
  int32_t Cin, CinPad, Cout, CoutPad;
  int32_t N, H, W, taps;
  int32_t S;            /* number of pixel splits (slabs), from hpfg_wgrad_splits() */
  int32_t math;         /* HPFG_MATH_F32 or HPFG_MATH_BF16X3 (3x3 only; 1x1 always runs the exact fp32 kernel) */
  int32_t defer_reduce; /* 1: leave the slabs in `slab`; the caller sums them later with hpfg_slab_reduce_multi (one launch per backward) */
} HpfgWgradArgs;

typedef struct HpfgFusedBwdArgs {   /* hpfg_fused_bwd: both gradients of a thin 3x3 layer from ONE staging of dZ (see below) */
  HpfgConvArgs d;       /* the dgrad side exactly as hpfg_conv_fwd takes it: a0 = dZ (DZ or PLAIN), wpk = wpk16_dgrad, out / out2 = dX,
                           Cout / CoutPad = the layer's Cin / CinPad, bwd_stats / bwd_of / stat_partials; math = HPFG_MATH_BF16X3 */
  HpfgAct xa0, xa1;     /* the layer's input as in forward (BNACT | BNACT_POOL | [BNACT | UP2X]) */
  float* slab;          /* [hpfg_fused_bwd_grid()][9][CinPad][CoutPad]: per-workgroup weight-gradient sums (hpfg_slab_reduce_multi) */
  int32_t Cin, CinPad, Cout, CoutPad;   /* of the layer: dW is [Cout][Cin][3][3] */
} HpfgFusedBwdArgs;

typedef struct HpfgSlabDesc {   /* one layer of hpfg_slab_reduce_multi: dw_oihw[co][ci][tap] = sum_s slab[s][tap][ci][co] */
  const float* slab;
  float* dw_oihw;
  int32_t S, taps, Cin, CinPad, Cout, CoutPad;
} HpfgSlabDesc;

typedef struct HpfgPackDesc {   /* one conv layer for hpfg_pack_weights (device array of these) */
  const float* w_oihw;  /* [Cout][Cin][k][k] (nn.Conv2d.weight, unet.py:18,22,50,99) */
  const float* b;       /* [Cout] */
  float* wpk_fwd;       /* [taps][CinPad/16][CoutPad/16][64][4]: B fragments of mfma_f32_16x16x4f32, k = input channel; may be NULL */
  float* wpk_dgrad;     /* [taps][CoutPad/16][CinPad/16][64][4]: transposed + tap-flipped weights for dgrad; may be NULL */
  float* bias_pad;      /* [CoutPad] */
  void* wpk16_fwd;      /* bf16x3 B fragments [k-step][CoutPad/16][hi|lo][64][8] (hpfg_wpk16_elems() bf16 values), or NULL */
  void* wpk16_dgrad;    /* same for dgrad (k = output channel, taps flipped), or NULL */
  int32_t Cout, Cin, CoutPad, CinPad, taps;
  int32_t kc;           /* K packing of the bf16 buffers: 32 = one tap x 32 channels, 16 = two taps x 16 channels (3x3 on 16x16 tiles);
                           use hpfg_conv_kc(H, W, taps) */
} HpfgPackDesc;

int hpfg_version(void);
const char* hpfg_last_error(void);
/* Kernel-form switches for tests and A/B tools (the launch paths read no environment variables): returns the previous value (>= 0) or < 0.
 * HPFG_OPT_CONV_THIN: 0 = the thin forward layers on the chunked conv kernel instead of the whole-tile kernel;
 * HPFG_OPT_FIRST_MFMA: 0 = the 1- / 3-channel first layer as a VALU loop instead of the MFMA form (bit-identical outputs);
 * HPFG_OPT_FIRST_WGRAD: 0 = the first layer's weight gradient on the tile kernel of hpfg_fused_bwd instead of the streaming kernel;
 * HPFG_OPT_NARROW_DEEP: the workgroup count up to which a 3x3 launch on 4 x 16-pixel tiles takes 32-wide instead of 64-wide output-channel slices
 *   (default 128: launches that would leave half of the CUs idle; 0 = never). */
enum { HPFG_OPT_CONV_THIN = 0, HPFG_OPT_FIRST_MFMA = 1, HPFG_OPT_FIRST_WGRAD = 2, HPFG_OPT_NARROW_DEEP = 3, HPFG_OPT_COUNT = 4 };
int hpfg_set_option(int which, int value);

/* ---- forward ------------------------------------------------------------------------------------------- */
/* nn.Conv2d(Cin<=4 -> 16, k3, p1) on the network input + BN partial sums (encoder.in_conv, unet.py:18-19,72). */
int hpfg_conv3x3_first_fwd(const HpfgAct* x, const float* w_oihw, const float* bias, float* out, float* stat_partials,
                           int N, int H, int W, int Cin, int Cout, void* stream);
int hpfg_conv_first_rows(int N, int H, int W);            /* rows of stat_partials hpfg_conv3x3_first_fwd writes (persistent grid) */
/* the same layer with its BatchNorm sums added to a layer accumulator (HpfgConvArgs.stat_acc / stat_shards) as well as / instead of rows of partial sums */
int hpfg_conv3x3_first_fwd_acc(const HpfgAct* x, const float* w_oihw, const float* bias, float* out, float* stat_partials, long long* stat_acc,
                               int stat_shards, int N, int H, int W, int Cin, int Cout, void* stream);
/* nn.Conv2d k3/k1 (+ fused producer BN/LeakyReLU/Dropout/MaxPool/Upsample/cat on load) + BN partial sums.
 * Also serves dgrad: a0 = dZ (mode DZ/PLAIN), wpk = wpk_dgrad. */
int hpfg_conv_fwd(const HpfgConvArgs* args, void* stream);
int hpfg_conv_stat_blocks(int N, int H, int W);          /* rows written by the fp32 kernels / upper bound for workspace sizing */
int hpfg_conv_stat_rows(const HpfgConvArgs* args);         /* rows hpfg_conv_fwd(args) writes (depends on args->math) */
/* BatchNorm2d train-mode statistics -> table rows mean/rstd/scale/shift, running stats (momentum .1, unbiased var)
 * (unet.py:19,23).  Either partials (float [nblk][2][C]) or pre-reduced sums (double [2][C], e.g. after an all-reduce). */
int hpfg_bn_fwd_finalize(const float* partials, int nblk, const double* sums, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, float* bn, int C, void* stream);
/* The same for EVERY BatchNorm layer of a forward pass in one launch, from the layers' sum accumulators (HpfgConvArgs.stat_acc): the forward
 * consumers derive scale / shift themselves (HpfgAct.bn_acc), so this call follows the LAST conv of the forward and writes the table rows
 * the backward kernels read, plus running_mean / running_var (momentum, unbiased variance).  Device array of descriptors + host copy. */
typedef struct HpfgBnAccDesc {
  const long long* acc;     /* [HPFG_ACC_WORDS(C, shards)]: read, then ZEROED for the next forward (every consumer has run by then) */
  const float* gamma;
  const float* beta;
  float* running_mean;      /* both or neither */
  float* running_var;
  float* bn;                /* table [HPFG_BN_ROWS][C]: rows mean, rstd, scale, shift are written */
  int32_t C;
  float count;              /* N * H * W */
  int32_t shards;
  int32_t reserved;
} HpfgBnAccDesc;
int hpfg_bn_acc_finalize(const HpfgBnAccDesc* table_dev, const HpfgBnAccDesc* table_host, int nlayers, float momentum, float eps, void* stream);
/* Backward counterpart.  With HpfgAct.bn_acc on a DZ source every dZ consumer (dgrad, fused dgrad + wgrad, wgrad) derives k1, k2, k3 from the
 * layer's BACKWARD accumulator -- sum(g), sum(g * xhat), added there by the dgrad epilogue that completed the layer's gradient
 * (HpfgConvArgs.bwd_stats + stat_acc) or by hpfg_bn_bwd_reduce[_pool]_acc -- so no hpfg_bn_bwd_finalize launch sits on the chain of backward.
 * This call follows the last dZ consumer of the listed layers: dgamma / dbeta (native_batch_norm_backward), the k1 .. k3 table rows, and the
 * accumulators are zeroed for the next pass. */
typedef struct HpfgBnAccBwdDesc {
  long long* acc;           /* [HPFG_ACC_WORDS(C, shards)]: read, then zeroed */
  const float* gamma;
  float* bn;                /* table: rows mean, rstd are read, k1, k2, k3 written */
  float* dgamma;            /* or NULL */
  float* dbeta;             /* or NULL */
  int32_t C;
  float count;
  int32_t shards;
  int32_t reserved;
} HpfgBnAccBwdDesc;
int hpfg_bn_acc_bwd_finalize(const HpfgBnAccBwdDesc* table_dev, const HpfgBnAccBwdDesc* table_host, int nlayers, void* stream);
int hpfg_bn_bwd_reduce_acc(const HpfgAct* g, int N, int H, int W, long long* acc, int shards, void* stream);
int hpfg_bn_bwd_reduce_pool_acc(const HpfgAct* g, const float* dP, int dp_pstride, int N, int Hp, int Wp, long long* acc, int shards, void* stream);
int hpfg_reduce_partials(const float* partials, int nblk, int C, double* sums, void* stream);
/* eval-mode BatchNorm (model.eval(), val.py:268-287): table rows from the running statistics */
int hpfg_bn_eval_table(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* bn, int C, void* stream);
int hpfg_conv_kc(int H, int W, int taps);
long hpfg_wpk16_elems(int Kchannels, int NchannelsPad, int taps, int kc);   /* bf16 elements of one wpk16 buffer */
int hpfg_pack_weights(const HpfgPackDesc* table_dev, const HpfgPackDesc* table_host, int nlayers, void* stream);
/* The same launch also advances two per-forward device counters (nn.BatchNorm2d's num_batches_tracked += 1 of a train-mode forward,
 * model/unet.py:14-27 via torch; and the engine's dropout seed word, so that a replayed hipGraph draws fresh nn.Dropout masks):
 * counters[0..n_counters) += 1 (int64), *seed_word = (*seed_word + seed_add) & 0x7fffffff.  Either may be absent (0 / NULL).
 * zero_words[0..n_zero) = 0: the BatchNorm sum accumulators (HpfgConvArgs.stat_acc) of the pass this launch precedes. */
int hpfg_pack_weights_bump(const HpfgPackDesc* table_dev, const HpfgPackDesc* table_host, int nlayers, long long* counters, int n_counters,
                           int32_t* seed_word, int seed_add, long long* zero_words, long n_zero, void* stream);
/* evaluate a virtual activation into memory (tests, projection-neck inputs): out [N,H,W,a0.C+a1.C] */
int hpfg_act_materialize(const HpfgAct* a0, const HpfgAct* a1, int N, int H, int W, float* out, void* stream);
/* the dropout keep-mask the loaders use, as bytes [n_elems] (tests feed it to the oracle) */
int hpfg_dropout_mask(uint8_t* out, long n_elems, float p, uint32_t seed, const uint32_t* seed_dev, void* stream);

/* ---- backward ------------------------------------------------------------------------------------------ */
/* per-channel sum(g), sum(g*xhat) with g = dA*dropmask*lrelu'(y): partials [hpfg_bn_bwd_blocks][2][C] */
int hpfg_bn_bwd_reduce(const HpfgAct* g /* mode DZ; k rows unused */, int N, int H, int W, float* partials, void* stream);
int hpfg_bn_bwd_blocks(int N, int H, int W, int C);
/* finalize: dgamma, dbeta (BatchNorm backward) and table rows k1,k2,k3 so that dz = k1*g + k2*z + k3 */
/* the same for a layer whose output also went through MaxPool2d(2) (model/unet.py:37): first dA += maxpool_backward(dP) in place
 * (arg-max recomputed from the raw output), then the sums -- replaces hpfg_pool_scatter_add + hpfg_bn_bwd_reduce in one pass */
int hpfg_bn_bwd_reduce_pool(const HpfgAct* g /* DZ source at the un-pooled size */, const float* dP, int dp_pstride, int N, int Hp, int Wp,
                            float* partials /* [hpfg_bn_bwd_pool_blocks()][2][C] */, void* stream);
int hpfg_bn_bwd_pool_blocks(int N, int Hp, int Wp, int C);
/* param_grad_scale multiplies dgamma / dbeta only: with all-reduced `sums` (data parallel, R ranks) every rank holds the GLOBAL
 * sums, so it writes global/R and the SUM all-reduce of the gradient buffer restores the global value; 1 otherwise */
int hpfg_bn_bwd_finalize(const float* partials, int nblk, const double* sums, double count, const float* gamma,
                         float* bn, float* dgamma, float* dbeta, int C, float param_grad_scale, void* stream);
int hpfg_wgrad(const HpfgWgradArgs* args, void* stream);
/* Fused backward of a thin 3x3 layer (bf16x3; H, W multiples of 16; CinPad <= 64, CoutPad <= 32): dX (as hpfg_conv_fwd with a dZ source,
 * including the bwd_stats epilogue and out2) AND the weight-gradient slabs (as hpfg_wgrad with defer_reduce) from one read of (dA, z) and
 * one read of the layer input -- replaces reference loss.backward()'s separate conv-transpose and weight-gradient kernels of one Conv2d
 * (model/unet.py:18,22).  hpfg_fused_bwd_grid: workgroups of the launch = slabs = rows of d.stat_partials; 0 = shape not instantiated
 * (the caller then uses hpfg_wgrad + hpfg_conv_fwd). */
int hpfg_fused_bwd(const HpfgFusedBwdArgs* args, void* stream);
int hpfg_fused_bwd_grid(const HpfgFusedBwdArgs* args);
int hpfg_slab_reduce_multi(const HpfgSlabDesc* table_dev, const HpfgSlabDesc* table_host, int nlayers, void* stream);
int hpfg_wgrad_splits(int N, int H, int W, int CinPad, int CoutPad, int taps);
long hpfg_wgrad_slab_floats(int N, int H, int W, int CinPad, int CoutPad, int taps);
/* first-layer wgrad: dW[co][ci][tap] for Cin<=4 from the strided input */
/* bias gradient of a conv without BN: db[c] = sum over pixels of g[p][c] (conv1x1 / out_conv) */
int hpfg_channel_sum(const float* g, int pstride, long npix, int C, float* out, float* scratch, void* stream);
/* first stage only: partials [hpfg_channel_sum_blocks(npix, C)][C]; the caller sums the rows later, e.g. as one more "layer"
 * {slab = partials, S = blocks, taps = 1, Cin = CinPad = 1, Cout = CoutPad = C, dw_oihw = db} of hpfg_slab_reduce_multi */
int hpfg_channel_sum_partials(const float* g, int pstride, long npix, int C, float* partials, void* stream);
int hpfg_channel_sum_blocks(long npix, int C);
/* MaxPool2d(2) backward: dA[argmax position] += dP, recomputing the arg-max from the producer's raw output */
int hpfg_pool_scatter_add(const HpfgAct* src /* BNACT view of the pooled tensor's producer */, const float* dP, int dp_pstride,
                          float* dA, int da_pstride, int N, int Hp, int Wp, void* stream);
/* bilinear x2 (align_corners) backward: dU[low res] = sum of taps of dUp (dUp has dup_pstride floats per pixel) */
int hpfg_upsample2x_bwd(const float* dUp, int dup_pstride, float* dU, int N, int Hl, int Wl, int C, void* stream);
/* same, and csum_partials[hpfg_upsample2x_bwd_blocks()][C] receives per-workgroup channel sums of dU (the bias gradient of the
 * 1x1 conv in front of the upsample, model/unet.py:50; rows are summed by hpfg_slab_reduce_multi like hpfg_channel_sum_partials') */
int hpfg_upsample2x_bwd_sums(const float* dUp, int dup_pstride, float* dU, int N, int Hl, int Wl, int C, float* csum_partials, void* stream);
int hpfg_upsample2x_bwd_blocks(int N, int Hl, int Wl, int C);

/* ---- data parallel: peer mailbox exchange (csrc/peer.h) ------------------------------------------------------------------------
 * The reference is single-device (main.py:44); under data parallel in the global-batch mode every BatchNorm layer's sums and the loss sums
 * must be added over the ranks between two kernels of the step.  Instead of ~90 latency-bound collectives per step, the kernel that
 * produces a sum exchanges it itself: it stores the value into every peer's mailbox (IPC-mapped fine-grained memory: xGMI between the
 * GPUs of a node), then reads the peers' values from its own mailbox and adds them in rank order.  No host code between the kernels,
 * so the step stays capturable into a hipGraph. */
#define HPFG_PEER_MAX_RANKS 8
#define HPFG_PEER_HANDLE_BYTES 64
#define HPFG_PEER_MAX_SPINS (1L << 22)      /* bounded poll (a few seconds): on expiry *err = 1 and the kernel carries on */
typedef struct HpfgPeerX {
  void* mbox[HPFG_PEER_MAX_RANKS];   /* mbox[r] = rank r's mailbox as mapped in this process (mbox[rank] = its own); unused entries NULL */
  const int32_t* epoch;              /* device word: the use count of `slot` (consecutive integers >= 1, the same on every rank) */
  int32_t* err;                      /* device word set to 1 when a poll expires (or NULL) */
  int32_t world, rank;               /* world <= 1: no exchange (the struct may be all zero) */
  int32_t slot, cap;                 /* slot index; cap = payload values per rank and parity in a slot */
  int64_t slot_bytes;                /* hpfg_peer_slot_bytes(world, cap) */
} HpfgPeerX;
long hpfg_peer_slot_bytes(int world, int cap);
int hpfg_peer_alloc(size_t bytes, void** ptr);                       /* zero-filled fine-grained device memory */
int hpfg_peer_free(void* ptr);
int hpfg_peer_handle(void* ptr, unsigned char* handle64);            /* HPFG_PEER_HANDLE_BYTES to send to the peers */
int hpfg_peer_open(const unsigned char* handle64, void** ptr);       /* map a peer's mailbox */
int hpfg_peer_close(void* ptr);
int hpfg_word_add(int32_t* word, int v, void* stream);               /* *word = (*word + v) & 0x7fffffff: the epoch bump, stream-ordered */
/* hpfg_bn_fwd_finalize / hpfg_bn_bwd_finalize / hpfg_seg_loss_partials with the cross-rank SUM of the per-channel (per-term) sums done by the
 * kernel itself (px->world > 1); `count` / the loss counts are the GLOBAL ones, param_grad_scale = 1 / world as with all-reduced sums */
int hpfg_bn_fwd_finalize_x(const float* partials, int nblk, const HpfgPeerX* px, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* bn, int C, void* stream);
int hpfg_bn_bwd_finalize_x(const float* partials, int nblk, const HpfgPeerX* px, double count, const float* gamma, float* bn, float* dgamma,
                           float* dbeta, int C, float param_grad_scale, void* stream);

/* Gradient all-reduce over the same kind of IPC-mapped memory: what DistributedDataParallel's bucketed NCCL all-reduce would be if the reference
 * ran on more than one card (main.py:44: it does not).  xGMI is point to point, so the exchange is not a ring: every rank owns one slice of the
 * buffer; (1) push: each rank stores its copy of slice p into rank p's window (all links busy at once), (2) reduce: the owner adds the
 * contributions in rank order and stores the reduced slice into every rank's window, (3) gather: each rank copies the reduced buffer back.
 * Flags carry the epoch (consecutive use counts, bumped by the call itself); polls are bounded like the mailbox polls.  Three launches on `stream`, no host code: the whole training step stays ONE hipGraph under data parallel.  Every rank receives bit-identical
 * sums (each slice is reduced once).  Window layout: 256 bytes of flags, world x slice floats of inbox, world x slice floats of result. */
typedef struct HpfgPeerBuf {
  void* win[HPFG_PEER_MAX_RANKS];    /* win[r] = rank r's window as mapped in this process */
  int32_t* epoch;                    /* device word counting the calls (the same sequence on every rank) */
  int32_t* err;                      /* device word set to 1 when a poll expires (or NULL) */
  int32_t world, rank;
  int64_t slice;                     /* floats per rank slice of THIS call: hpfg_peer_buf_slice(world, n) */
  int64_t n;                         /* floats of the buffer */
  int64_t stride;                    /* floats between the inbox regions of a window = hpfg_peer_buf_slice(world, CAPACITY the windows were sized
                                        for); the result region begins at world * stride.  The layout must not depend on the call's n: a rank
                                        that has finished call k pushes call k+1 while a peer may still copy its call-k result out */
} HpfgPeerBuf;
long hpfg_peer_buf_slice(int world, long n);                 /* ceil(n / world) rounded up to a multiple of 4 */
long hpfg_peer_buf_bytes(int world, long n);                 /* window size for buffers of up to n floats */
int hpfg_peer_allreduce_f32(const HpfgPeerBuf* pb, float* buf, void* stream);      /* buf[0..n) = SUM over ranks, in place */

/* ---- losses (main.py:164-197, medloss.py:44-56, diceloss.py:155-191, Mean-Teacher :103-106) -------------- */
typedef struct HpfgLossArgs {
  const float* logits;      /* [N,H,W,C] student logits (NHWC) */
  const float* t_logits;    /* teacher logits for the MSE consistency term, or NULL */
  const uint8_t* labels0;   /* labels for images [0,n_lab) (255 = ignore), group 0 */
  const uint8_t* labels1;   /* (pseudo-)labels for images [n_lab,N), group 1, or NULL */
  const float* coef;        /* device [8]: ce0, dice0, ce1, dice1, mse_w, 0,0,0  (per-step weights live on the device) */
  float* partials;          /* workspace [hpfg_loss_blocks()][HPFG_LOSS_NSUM] */
  float* sums;              /* out [HPFG_LOSS_NSUM] reduced sums (all-reduce these for data parallel) */
  float* out;               /* out [8]: total, ce0, dice0, ce1, dice1, mse, 0, 0 */
  float* dlogits;           /* backward output [N,H,W,C] */
  int32_t N, n_lab, H, W, C;
  int32_t world;            /* data-parallel world size (MSE / CE counts are global after the all-reduce) */
  int32_t input_is_prob;    /* 1: `logits` already holds probabilities (DiceLoss(softmax=False), diceloss.py:178); CE/MSE weights must be 0 */
  int32_t teacher_is_prob;  /* 1: `t_logits` already holds probabilities (ICT's mixed teacher prediction, 2022_02...ICT...py:126-137) */
  int32_t t_unlab_only;     /* 1: `t_logits` covers images [n_lab,N) only; 0: it is indexed like `logits` (all N images) */
  int32_t reserved0;
  const float* cons_mask;   /* [N-n_lab][H][W] 0/1 weights of the consistency term, or NULL.  With a mask the term is
                               sum(mask * d^2) / (2*sum(mask) + 1e-16)  (UAMT, 2019_07...Uncertainty_Aware...py:160-164) instead of the mean */
} HpfgLossArgs;
#define HPFG_LOSS_NSUM 32
int hpfg_loss_blocks(int N, int H, int W);
int hpfg_seg_loss_partials(const HpfgLossArgs* a, void* stream);   /* softmax + CE/Dice/MSE partial sums */
int hpfg_seg_loss_partials_x(const HpfgLossArgs* a, const HpfgPeerX* px, void* stream);   /* + the cross-rank sum of `sums` (peer mailbox) */
int hpfg_seg_loss_finalize(const HpfgLossArgs* a, void* stream);   /* sums -> loss scalars (device) */
int hpfg_seg_loss_bwd(const HpfgLossArgs* a, const float* grad_scale_dev /* NULL = 1 */, void* stream);   /* dlogits */
/* pseudo-labels: argmax over classes of teacher logits, optionally CutMix-blended with labels (main.py:177-178) */
int hpfg_argmax_labels(const float* logits, int N, int H, int W, int C, const uint8_t* mix_labels, const float* mix_mask,
                       uint8_t* out, void* stream);
/* evaluation (val.py:376-387, medpy binary dc): counts[gt*C + pred] += 1 over n voxels (labels >= C are ignored); the caller
 * zeroes `counts` (C*C uint64) and derives per-class Dice = 2*n(A&B) / (n(A) + n(B)) from rows / columns */
int hpfg_confusion_counts(const uint8_t* pred, const uint8_t* gt, long n, int C, unsigned long long* counts, void* stream);
/* Training-time slice augmentation on the device (datasets/utils.py:73-117 RandomGenerator.__call__: random_rot_flip | random_rotate,
 * then scipy zoom(order=0) to the network size, image and mask alike).  The host draws the random parameters in the reference's
 * order and supplies, per sample, the source slice, the rot90/flip or rotation parameters (rotation matrix and offset exactly
 * as scipy.ndimage.rotate computes them) and scipy's 1-D zoom index tables; the kernel gathers image [B][1][H][W] and mask [B][H][W]. */
typedef struct HpfgAugSample {
  int64_t img_off, lab_off;   /* element offsets of the source slice in the pools */
  int32_t h, w;               /* source slice size */
  int32_t mode;               /* 0 none, 1 rot90 + flip, 2 rotate */
  int32_t k, axis;            /* mode 1: np.rot90(a, k) then np.flip(axis) */
  int32_t tab_off;            /* offset into tabs: H row indices then W column indices of the intermediate image */
  double m00, m01, m10, m11, off_y, off_x;   /* mode 2: input coordinate = M * output index + offset */
} HpfgAugSample;
int hpfg_augment_batch(const float* img_pool, const uint8_t* lab_pool, const HpfgAugSample* samples_dev, const int* tabs_dev, int B, int H, int W,
                       float* out_img, uint8_t* out_lab, void* stream);
/* CutMix box masks (utils/utils.py:115-173 BoxMaskGenerator.generate_params): rects int32 [n][n_boxes][y0,y1,x0,x1] (bounds already
 * normalised like Python slices), out float [n][1][H][W] = (invert ? 0 : 1) flipped once per covering box */
int hpfg_box_masks(const int* rects, int n, int n_boxes, int H, int W, int invert, float* out, void* stream);
/* per-sample linear mix out[s] = a[s]*(1-f[s]) + b[s]*f[s] (ICT input mix, 2022_02_ISBI_ICT-MedSeg_ACDC.py:111-117) */
int hpfg_mix_samples(const float* a, const float* b, const float* f, float* out, int n, long per_sample, void* stream);
/* softmax(t0)*(1-f[s]) + softmax(t1)*f[s] over NHWC logits [n,H,W,C] (ICT mixed teacher prediction, :126-129) */
int hpfg_softmax_mix(const float* t0, const float* t1, const float* f, float* out_prob, int n, int H, int W, int C, void* stream);
/* diagnostics (tools/stream_timeline.py): *slot = s_memrealtime (100 MHz) when `stream` reaches this point; capturable */
int hpfg_timestamp(unsigned long long* slot, void* stream);
/* UAMT teacher-input noise (2019_07_MICCAI_Uncertainty_Aware_ACDC.py:130,142): out[i] = x[i % n_src] + clamp(noise[i]*scale, lo, hi) */
int hpfg_noise_add(const float* x, const float* noise, float* out, long n_src, long n_out, float scale, float lo, float hi, void* stream);
/* UAMT uncertainty mask (:147-151,162-163): T stochastic teacher predictions (NHWC logits) of the same S images, prediction
 * g = t*S + s stored in block g / per_block; mask[s,h,w] = (-sum_c pm*log(pm + 1e-6) < *threshold_dev), pm = mean_t softmax */
typedef struct HpfgPredBlocks {
  const float* p[8];
  int32_t n_blocks, per_block;
} HpfgPredBlocks;
int hpfg_uncertainty_mask(const HpfgPredBlocks* pb, int T, int S, int H, int W, int C, const float* threshold_dev, float* mask /* [S,H,W] */,
                          float* uncertainty /* [S,H,W] or NULL */, void* stream);
/* CutMix image blend x1*(1-M)+xu*M (main.py:149) */
int hpfg_cutmix_blend(const float* a, const float* b, const float* mask, float* out, long n, void* stream);

/* ---- token-layout ops of the SegFormer branch (model/segformer.py).  [B,N,C] tokens == NHWC pixels. --- */
/* nn.LayerNorm(C) (eps 1e-5) over the last dimension (segformer.py:107,174,192,195,232-244); mean / rstd [rows] are kept for backward */
int hpfg_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows, int C, void* stream);
int hpfg_ln_bwd(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta /* == dgamma + C */,
                float* partials /* [hpfg_ln_bwd_blocks(rows)][2][C] */, long rows, int C, void* stream);
int hpfg_ln_bwd_blocks(long rows);
/* Attention.forward :122-126 without its Linear layers: out[b,i,h,:] = softmax_j(scale * q[b,i,h,:].k[b,j,h,:]) v[b,j,h,:];
 * q [B,N,heads,32], kv [B,M,2,heads,32] (the kv Linear's output as reshaped at :120), M <= 64 keys, head dim 32 */
int hpfg_attn_fwd(const float* q, const float* kv, float* out, int B, int N, int M, int heads, float scale, void* stream);
/* backward: dq, and the matrices P, dS [B,heads,N,M] from which dV = P^T dO and dK = scale * dS^T Q follow (GEMMs) */
int hpfg_attn_bwd(const float* q, const float* kv, const float* dout, float* dq, float* P, float* dS, int B, int N, int M, int heads, float scale,
                  void* stream);
/* DWConv (3x3 depthwise, pad 1, :139-146) + F.gelu (:156) on [B,H,W,C]; w9 = the [C,1,3,3] weight transposed to [9][C] */
int hpfg_dwgelu_fwd(const float* x, const float* w9, const float* bias, float* y, int B, int H, int W, int C, void* stream);
int hpfg_dwgelu_bwd(const float* x, const float* w9, const float* bias, const float* dy, float* du /* scratch [B,H,W,C] */, float* dx, float* dw9,
                    float* dbias /* == dw9 + 9*C */, float* partials /* [hpfg_dwgelu_bwd_blocks()][10][C] */, int B, int H, int W, int C, void* stream);
int hpfg_dwgelu_bwd_blocks(int B, int H, int W);
/* F.interpolate(mode="bilinear", align_corners=False) of SegFormerHead.forward (:314,319) on NHWC [B,h,w,C] -> [B,H,W,C]; backward is a
 * gather over the outputs that tap a source pixel (no atomics), upsampling only */
int hpfg_resize_bilinear_fwd(const float* x, float* y, int B, int h, int w, int H, int W, int C, void* stream);
/* y = base + sum_k resize(xs[k]) (same bilinear map per source; base, y: [B,H,W,C]; xs[k]: [B,hs[k],ws[k],C]; nsrc <= 3; y may alias base):
 * the head's sum over stages of F.interpolate(...) (model/segformer.py:309-314 with linear_fuse applied per stage) in one pass */
int hpfg_resize_sum_fwd(const float* base, const float* const* xs, const int* hs, const int* ws, int nsrc, float* y, int B, int H, int W, int C,
                        void* stream);
int hpfg_resize_bilinear_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, int C, void* stream);
/* ConvModule's BatchNorm2d (train) + ReLU and the head's Dropout2d (segformer.py:288-296,307,318) over tokens [R,C]:
 * column sums (sum x, sum x^2) -> the caller forms mean / rstd (and the running statistics) -> apply; mask [R/rows_per_image][C] of
 * 0/1 keeps (or NULL), scaled by inv_keep.  Backward: sums [2][C] = (sum g, sum g*xhat) = (dbeta, dgamma), then dx. */
/* residual branch with stochastic depth (Block.forward :197-198, DropPath :23-30): out[b] = x[b] + y[b] * scale[b] (scale NULL = 1); the
 * branch gradient is dout[b] * scale[b] */
int hpfg_residual_scale(const float* x, const float* y, const float* scale, float* out, int B, long per_sample, void* stream);
int hpfg_scale_rows(const float* d, const float* scale, float* out, int B, long per_sample, void* stream);
/* im2col / col2im of PatchEmbed.proj (segformer.py:172: kernel k, stride s, padding k/2) on NHWC x [B,H,W,C]:
 * cols [B, Ho*Wo, k*k*C] with the patch ordered (u, v, c); col2im is the gather-form transpose (no atomics) */
int hpfg_im2col_nhwc(const float* x, float* cols, int B, int H, int W, int C, int k, int s, void* stream);
int hpfg_col2im_nhwc(const float* dcols, float* dx, int B, int H, int W, int C, int k, int s, void* stream);
/* weight gradient of nn.Linear over many tokens, dW[N][K] = dY[R][N]^T X[R][K] with R >> N, K (the q / kv / proj / fc1 / fc2 layers of the
 * high-resolution stages): row-split partial products + fixed-order reduction */
int hpfg_linear_wgrad(const float* dy, const float* x, float* dw, float* partials /* [hpfg_linear_wgrad_splits()][N][K] */, long R, int N, int K,
                      void* stream);
int hpfg_linear_wgrad_splits(long R, int N, int K);
int hpfg_tok_col_stats(const float* x, long R, int C, float* partials /* [hpfg_tok_stat_blocks(R)][2][C] */, float* sums /* [2][C] */, void* stream);
int hpfg_tok_stat_blocks(long R);
int hpfg_bnrelu_apply(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* mask, float inv_keep,
                      long rows_per_image, float* y, long R, int C, void* stream);
int hpfg_bnrelu_bwd(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* mask,
                    float inv_keep, long rows_per_image, float* dx, float* partials, float* sums, long R, int C, void* stream);

/* ---- parameter updates --------------------------------------------------------------------------------- */
/* torch.optim.SGD(momentum, weight_decay) over a flat parameter buffer (utils/__init__.py:15-16); lr read from device */
int hpfg_sgd_step(float* p, const float* g, float* mom, long n, const float* lr_dev, float momentum, float weight_decay,
                  float grad_scale, void* stream);
/* torch.optim.AdamW (utils/__init__.py:17-19: lr, weight_decay; betas (0.9, 0.999), eps 1e-8) over flat buffers; lr and the step count
 * (float, number of steps already taken; advanced by the call) are read from device memory, so the update replays inside a hipGraph */
int hpfg_adamw_step(float* p, const float* g, float* m, float* v, long n, const float* lr_dev, float* step_dev, float beta1, float beta2, float eps,
                    float weight_decay, float grad_scale, void* stream);
/* EMA teacher: t = alpha*t + (1-alpha)*s over flat buffers (utils/utils.py:82-86); alpha read from device */
int hpfg_ema_update(float* t, const float* s, long n, const float* alpha_dev, void* stream);

/* Both in one pass over the student's flat buffers -- the last two statements of every Mean-Teacher-family iteration
 * (2017_03_NIPS_Mean-Teacher_ACDC.py:108-113: optimizer.step(); update_ema_variables(...)): t[i] = alpha*t[i] + (1-alpha)*p_new[i] for i < n_ema
 * (n_ema = n, or the leading backbone slice of main.py:68-76).  Bit-identical to hpfg_sgd_step followed by hpfg_ema_update. */
int hpfg_sgd_ema_step(float* p, const float* g, float* mom, long n, const float* lr_dev, float momentum, float weight_decay, float grad_scale,
                      float* t, long n_ema, const float* alpha_dev, void* stream);

/* ---- attention core on the matrix cores (SegFormer branch; reference model/segformer.py:92-128 after the q / kv projections) -------------- */
/* out = softmax(scale * q k^T) v per (image, head): q [B,N,heads,32], kv [B,M,2,heads,32] (the kv Linear's output), M <= 64 keys, head dim 32;
 * split-bf16 MFMA products, fp32 softmax.  Backward: dq like q, dkv like kv; scratch B * heads * hpfg_attn_mfma_blocks(N) * 2*64*32 floats
 * (per-workgroup dK / dV partials, summed in a fixed order). */
int hpfg_attn_mfma_fwd(const float* q, const float* kv, float* out, int B, int N, int M, int heads, float scale, void* stream);
int hpfg_attn_mfma_bwd(const float* q, const float* kv, const float* dout, float* dq, float* dkv, float* scratch, int B, int N, int M, int heads,
                       float scale, void* stream);
int hpfg_attn_mfma_blocks(int N);

/* ---- projection necks + Dense_Loss (UNet_Plus; reference model/unet.py:120-152, utils/loss/dense_loss.py:17-40) ------------------------ */
/* C[m,n] = act(sum_k A(m,k) B(k,n) + bias[n]) in exact fp32 on the matrix cores; A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn], C row-major
 * (ldc); relu: max(.,0) in the epilogue; accumulate: C += (before the activation).  Replaces nn.Linear / 1x1 nn.Conv2d forward and both backward
 * products of the necks (unet.py:125-138) and torch.mm of dense_loss.py:24.  Deterministic: no split-K, no atomics. */
int hpfg_gemm_f32(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N, int K,
                  const float* bias, int relu, int accumulate, void* stream);
/* the same with the contraction split over workgroups when the product has few output tiles and a long K (fixed-order sum of the partials);
 * scratch: hpfg_gemm_f32_splits(M, N, K) * M * N floats */
int hpfg_gemm_f32_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N, int K,
                         const float* bias, int relu, int accumulate, float* scratch, void* stream);
int hpfg_gemm_f32_splits(int M, int N, int K);
/* the same contract in split-bf16 arithmetic (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate; 128 x 128 tiles): every
 * nn.Linear / kernel==stride conv / 1x1 conv of the SegFormer branch (reference model/segformer.py:92-177, 298-320), forward and input
 * gradient, and the weight gradient of layers with few tokens.  Operands need a unit stride along one index and 4-element alignment:
 * hpfg_gemm_bf16x3_ok() tells; callers fall back to hpfg_gemm_f32 otherwise. */
int hpfg_gemm_bf16x3(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc, int M, int N, int K,
                     const float* bias, int relu, int accumulate, void* stream);
/* The same product with the contraction split over workgroups where the output has few 128 x 128 tiles and K is long (hpfg_gemm_bf16x3_splits > 1:
 * the 1568-token layers, K up to 2048); scratch: splits * M * N floats; partial products are added in split order (deterministic) */
int hpfg_gemm_bf16x3_splits(int M, int N, int K);
int hpfg_gemm_bf16x3_splitk(const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* Cm, long ldc, int M, int N, int K,
                            const float* bias, int relu, int accumulate, float* scratch, void* stream);
int hpfg_gemm_bf16x3_ok(const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N, int K);
/* weight gradient dW[N][K] = dY^T X over R tokens (dY [R][N], X [R][K] contiguous) in split-bf16 arithmetic: operands staged as they lie in
 * memory, fragments through the transposing LDS read, rows split over workgroups, partials summed in a fixed order.
 * with_db: the bias gradient db[N] = column sums of dY comes out of the same pass, written right behind dW (dw_db: N*K + N floats).
 * partials: hpfg_gemm_tn_splits(R, N, K) * (N*K + N) floats of scratch.  (autograd's mm / sum for the parameters of nn.Linear, segformer.py) */
int hpfg_gemm_tn_bf16x3(const float* dy, const float* x, float* dw_db, float* partials, long R, int N, int K, int with_db, void* stream);
int hpfg_gemm_tn_splits(long R, int N, int K);
int hpfg_col_sum(const float* x, long R, int M, long ldx, float* out /* [M] = sum over rows */, void* stream);   /* bias gradients */
int hpfg_col_sum2(const float* x, long R, int M, long ldx, float* out, float* scratch /* hpfg_col_sum_splits(R) * M floats */, void* stream);
int hpfg_col_sum_splits(long R);
int hpfg_relu_bwd(float* dy, const float* y, long n, void* stream);                                         /* dy *= (y > 0): nn.ReLU backward */
/* nn.AdaptiveAvgPool2d((1,1)) and ((S,S)) of an NHWC tensor in one launch (unet.py:141-142,146): gap [N,C], pool [N,S*S,C] */
int hpfg_neck_pool_fwd(const float* x, int pstride, int N, int H, int W, int C, int S, float* gap, float* pool, void* stream);
int hpfg_neck_pool_bwd(const float* dgap /* or NULL */, const float* dpool /* or NULL */, int N, int H, int W, int C, int S,
                       float* dx /* NHWC contiguous, overwritten */, void* stream);
/* F.normalize(x, dim=1) of x viewed as [G,D,S] with element (g,d,s) at g*D*S + d*sd + s*ss, (sd,ss) = (S,1) or (1,D) (dense_loss.py:18-19) */
int hpfg_l2norm_fwd(const float* x, int G, int D, int S, int sd, int ss, float* u, float* norms /* [G*S] */, void* stream);
int hpfg_l2norm_bwd(const float* du, const float* u, const float* norms, int G, int D, int S, int sd, int ss,
                    const float* scale /* device scalar multiplied into dx, or NULL */, float* dx, void* stream);
/* NT-Xent of dense_loss.py:24-36 from the Gram matrix [2n,2n] of the normalised rows [student ; teacher]: loss[0] = mean_i -log(pos_i / denom_i);
 * Q (or NULL) [n,2n] = dL/dGram + its transpose for the student rows, so that dL/d(student row i) = sum_j Q[i,j] * row_j */
int hpfg_ntxent_rows(const float* gram, int n, float temperature, float* loss, float* Q, void* stream);

#ifdef __cplusplus
}
#endif
#endif
