/*
 * asv_amd.h - C ABI of libasv_amd.so, the MI355X (gfx950) embedding-extraction and scoring
 * engine that sits behind asv-subtools' TopVirtualNnet.extract_embedding() boundary.
 *
 * Plain C: pointers, sizes and ints only; no torch / HIP types in any signature (a HIP
 * stream is passed as `void*`, NULL = the default stream).  Every function returns 0 on
 * success or a negative errno-style code; asv_last_error() returns the thread-local text.
 * Nothing throws across this boundary.  One asv_net_t belongs to one device; the caller
 * owns every buffer it passes in.  Pointers documented "device" must be device-accessible
 * on the current HIP device; "host" pointers are read during the call only.
 *
 * Reference interfaces replaced (paths relative to /root/reference/pytorch):
 *   asv_net_*  builder  <- the nn.Module graph a model blueprint builds in `init()`
 *                          (model/xvector.py:18-40, model/ecapa_tdnn_xvector.py:201-357),
 *                          one call per layer object instead of one torch module.
 *   asv_net_extract     <- `for_extract_embedding` + `<Model>.extract_embedding`
 *                          (libs/nnet/framework.py:12-55, model/xvector.py:77-98), batched over
 *                          utterances instead of batch=1 (pipeline/onestep/extract_embeddings.py:73-83).
 *   asv_tdnn_forward    <- TdnnAffine.forward + _BaseActivationBatchNorm.forward
 *                          (libs/nnet/components.py:107-149, 418-431) on one layer.
 *   asv_stats_pool_forward <- StatisticsPooling.forward (libs/nnet/pooling.py:32-69).
 *   asv_cosine_* / asv_length_norm / asv_plda_* / asv_eer: see the scoring section below.
 *
 * Data layout: activations are "frames x channels" row-major (the Kaldi matrix layout,
 * libs/support/kaldi_io.py:466-496), utterances packed back to back; `offsets[n_utts+1]`
 * gives each utterance's first row.
 */
#ifndef ASV_AMD_H
#define ASV_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASV_AMD_VERSION 100            /* 0.1.0 */

/* ---- error codes ------------------------------------------------------------------ */
#define ASV_OK        0
#define ASV_EINVAL  (-22)              /* bad argument / malformed graph               */
#define ASV_ENOMEM  (-12)              /* host or device allocation failed             */
#define ASV_EHIP     (-5)              /* a HIP runtime call failed (see last_error)   */
#define ASV_ENOTSUP (-95)              /* valid request this build does not implement  */
#define ASV_ESTATE  (-1)               /* call order violated (e.g. extract before finalize) */

int         asv_version(void);
const char *asv_last_error(void);      /* thread-local, never NULL                      */
int         asv_device_count(int *count);

/* ---- enums ------------------------------------------------------------------------ */
/* arithmetic of the frame layers */
#define ASV_PREC_F32   0   /* f32 activations/weights, v_mfma_f32_32x32x2_f32 (exact f32 fma chain) */
#define ASV_PREC_BF16  1   /* bf16 activations/weights, v_mfma_f32_32x32x16_bf16, f32 accumulate   */
#define ASV_PREC_F32X  2   /* f32 activations; the wide frame layers and the pooled layers run on the bf16 matrix cores with both
                            * operands split into bf16 hi + lo halves (hi*hi + hi*lo + lo*hi, f32 accumulate): f32-grade results
                            * (<= 1e-4 of the reference) at ~1/3 of the bf16 rate instead of the 1/16 of the f32-input MFMA */
#define ASV_PREC_F16   3   /* IEEE-half activations/weights, v_mfma_f32_32x32x16_f16, f32 accumulate: the rate and footprint of
                            * ASV_PREC_BF16 with an 11-bit significand (8x less operand rounding).  Range +-65504: activations
                            * behind eval BatchNorm and mean-normalised features are O(1..100); a value beyond it becomes
                            * inf and the embedding NaN - loud, never silently saturated                                   */
/* flags for asv_net_create */
#define ASV_FLAG_REF_KERNELS 1u  /* run the plain-VALU self-check kernels instead of MFMA ones */
#define ASV_FLAG_NO_FUSE     2u  /* disable epilogue fusions (stats pooling into the producer)  */
#define ASV_FLAG_SMALL_TILES 4u  /* never pick the 256x256 kernels (A/B testing)                    */
#define ASV_FLAG_BIG_V2      8u  /* reserved, ignored: the 256x256 both-operands-through-LDS kernel it selected was removed in round 2 */
#define ASV_FLAG_NO_CHAIN   16u  /* one launch per layer: do not run tdnn -> 1-tap tdnn -> ... -> pooling chains in one kernel */
/* ASV_PREC_F32X only: the 16-bit type both operands are split into (neither bit: the library default, half) ... */
#define ASV_FLAG_X3_SPLIT_BF16 32u /* bf16 hi + bf16 lo: 16 significant bits per operand, the full f32 exponent range              */
#define ASV_FLAG_X3_SPLIT_F16  64u /* half hi + half lo: 22 significant bits per operand (weights pre-scaled by a power of two per
                                    * layer so that both halves are normal numbers), operands within +-65504                      */
/* ... and measurement variants that drop one of the three products hi*hi + w_hi*x_lo + w_lo*x_hi (NOT f32-grade: they exist so
 * that "two matrix instructions per product are not enough" is a measured statement, LABLOG.md "Precision modes") */
#define ASV_FLAG_X3_NO_XLO   128u  /* activations rounded to one 16-bit value (no w_hi*x_lo product) */
#define ASV_FLAG_X3_NO_WLO   256u  /* weights rounded to one 16-bit value (no w_lo*x_hi product)     */
#define ASV_FLAG_X3_MX8     1024u  /* ASV_PREC_F32X with half halves ("f32m"): the two correction products w_hi*x_lo + w_lo*x_hi run as ONE block-scaled
                                    * 8-bit matrix instruction per 32 channels (e4m3 weights, e5m2 activations, power-of-two block scales) next to
                                    * the exact half product w_hi*x_hi: ~1e-5 of the f32 forward instead of ~1e-7, 4 instead of 6 matrix-pipe time
                                    * units per product.  Kernels without this form run the three half products as before. */
#define ASV_FLAG_X3_TILE128  512u  /* ASV_PREC_F32X: the split kernel's 128-row tiles whatever the batch size (by default batches too small to
                                    * fill the chip with them take its 64-row tiles); same results to the last bit of the f32 sums' order */

#define ASV_ACT_NONE    0
#define ASV_ACT_RELU    1
#define ASV_ACT_TANH    2
#define ASV_ACT_SIGMOID 3

#define ASV_DOMAIN_FRAMES 0  /* one row per frame, ragged over utterance segments */
#define ASV_DOMAIN_UTTS   1  /* one row per utterance segment (after pooling); always f32 */
/* ids >= 2: (time, frequency) grids of the 2-D ResNet trunk, created by asv_net_define_grid() */

#define ASV_MAX_TAPS 9

/* ---- net builder ------------------------------------------------------------------ */
typedef struct asv_net asv_net_t;

/* Creates an empty layer program on HIP device `device`.  Buffer 0 is the input feature
 * matrix (frames domain, `feat_dim` channels). */
int  asv_net_create(asv_net_t **net, int device, int precision, unsigned flags, int feat_dim);
void asv_net_destroy(asv_net_t *net);

/* Declares a 2-D row domain for libs/nnet/resnet.py-style trunks: a segment of T frames owns
 * ceil(T / 2^time_shift) frames x `pitch` rows ((time, frequency) positions, frequency fastest),
 * of which the first `width` rows of every frame are real and the rest zero padding.  A 3x3
 * convolution (resnet.py:12-15) over such a buffer is a 9-tap layer with row offsets dt*pitch + df.
 * 1 <= width < pitch <= 83; or width = pitch = 1: a SEQUENCE domain - one row per frame of the subsampled time axis, two
 * zero rows between utterances - on which frame-level layers (|tap| <= 2) and every pooling run as on the frames domain
 * (asv_net_add_grid_flatten writes the 2-D trunk's [B, C*F', T'] reshape there).  Returns the domain id (>= 2). */
int  asv_net_define_grid(asv_net_t *net, int time_shift, int width, int pitch);

/* Declares an activation buffer; returns its id (>= 1) or a negative error. */
int  asv_net_new_buffer(asv_net_t *net, int domain, int channels);

/* One TdnnAffine [+ activation + eval-BatchNorm] layer (components.py:107-149,418-431):
 *   z   = sum_taps W[:, :, tap] . (x[t+tap] (+ x2[t+tap]))  + bias (+ seg_bias[segment])
 *   z   = affine_first ? act1(z*scale+shift) : act1(z)*scale+shift
 *   y   = act2(z) (* seg_scale[segment]) (+ residual)
 * Frames outside an utterance segment read as zero at every layer (components.py:116-117).
 * All host arrays are copied during the call. */
typedef struct asv_tdnn_desc {
  uint32_t struct_size;          /* = sizeof(asv_tdnn_desc_t) */
  int32_t in_buf, in_ch_off;     /* input view: channels [in_ch_off, in_ch_off+in_ch)        */
  int32_t in2_buf, in2_ch_off;   /* optional second input added to the first, or in2_buf=-1  */
  int32_t out_buf, out_ch_off;
  int32_t in_ch, out_ch;
  int32_t n_taps;
  int32_t taps[ASV_MAX_TAPS];    /* frame offsets, ascending (the layer's `context`)         */
  const float *weight;           /* host, dense [out_ch][in_ch][w_tot_context] as in the
                                    checkpoint; tap offset o lives at index o - w_left_context */
  int32_t w_tot_context, w_left_context;
  const float *bias;             /* host [out_ch] or NULL                                     */
  int32_t seg_bias_buf;          /* utts-domain buffer [segments][out_ch] added to z, or -1  */
  int32_t act1;
  const float *scale, *shift;    /* host [out_ch] folded eval-BN (both or neither)            */
  int32_t affine_first;          /* 1 = the "bn-relu" order (components.py:365-386)           */
  int32_t act2;
  int32_t seg_scale_buf;         /* utts-domain buffer [segments][out_ch] multiplier, or -1  */
  int32_t res_buf, res_ch_off;   /* residual view added last, or res_buf=-1                   */
  float alg_fraction;            /* profiling only: share of the layer's taps x in_ch weight blocks that are not structurally zero
                                    (a stride-2 3x3 convolution lowered as 2x2 over four gathered phases: 9/16); the algorithmic
                                    FLOPs reported for the layer are this x 2 in_ch out_ch n_taps per row.  0 = 1              */
} asv_tdnn_desc_t;
int asv_net_add_tdnn(asv_net_t *net, const asv_tdnn_desc_t *d);

/* Pooling over the frames of each segment -> one utts-domain row (pooling.py:58-67,
 * ecapa_tdnn_xvector.py:176-178).  out[0:C]=mean, out[C:2C]=std when stddev. */
#define ASV_POOL_VAR_CLAMP 0     /* std = sqrt(max(var, eps))   (StatisticsPooling)          */
#define ASV_POOL_VAR_ADD   1     /* std = sqrt(var + eps)       (ECAPA global context)       */
typedef struct asv_pool_desc {
  uint32_t struct_size;
  int32_t in_buf, in_ch_off, channels;
  int32_t out_buf, out_ch_off;   /* utts domain; needs channels*(1+stddev) columns            */
  int32_t stddev, unbiased, var_mode;
  float   eps;
  int32_t per_bin;               /* grid inputs: 1 = pool every frequency bin separately over time; the
                                    output row holds [bin][mean(C) | std(C)] (ResNetXvector reshape +
                                    StatisticsPooling, resnet_xvector.py:193-194, in bin-major order) */
} asv_pool_desc_t;
int asv_net_add_stats_pool(asv_net_t *net, const asv_pool_desc_t *d);

/* ECAPA attentive statistics (ecapa_tdnn_xvector.py:182-188): alpha = softmax over the
 * segment's frames of `logits`; mean = sum alpha*x; std = sqrt(max(sum alpha*x^2 - mean^2, eps)). */
typedef struct asv_attpool_desc {
  uint32_t struct_size;
  int32_t x_buf, x_ch_off, logit_buf, logit_ch_off, channels;
  int32_t out_buf, out_ch_off;
  float   eps;
  int32_t shared_logits;         /* != 0: ONE logit per frame (column logit_ch_off of logit_buf) weights every channel -
                                    AttentiveStatisticsPooling with its shared single head, libs/nnet/pooling.py:322-370 */
  int32_t logit_group;           /* > 1: every logit_group consecutive channels share logit column (channel / logit_group) -
                                    MultiHeadAttentionPooling with shared weights, pooling.py:371-438 (channels / num_head);
                                    0: per shared_logits */
  /* xi-vector posterior pooling (pooling.py:165-218, per-channel logits only): */
  int32_t logit_softplus2;       /* != 0: the stored logits are raw precisions z; the weight logit is 2 log(softplus(z)) */
  const float *prior_logit;      /* host [channels] or NULL: one more "frame" per utterance with these logits ...        */
  const float *prior_value;      /* ... and these values (prior_logprec / prior_mean), not transformed                  */
} asv_attpool_desc_t;
int asv_net_add_attentive_pool(asv_net_t *net, const asv_attpool_desc_t *d);

/* Learnable dictionary encoding pooling (libs/nnet/pooling.py:130-162, LDEPooling): with r = x_t - mu_k,
 * w[t][k] = softmax over the centres k of -beta_k |r|^2, out[c * n_centres + k] = mean over the frames of w[t][k] r[c].
 * mu: host [channels][n_centres]; beta: host [n_centres] (= s^2 + eps); n_centres <= 64; the output view takes
 * channels * n_centres columns of an utterance-domain buffer. */
typedef struct asv_lde_desc {
  uint32_t struct_size;
  int32_t x_buf, x_ch_off, channels, n_centres;
  int32_t out_buf, out_ch_off;
  const float *mu, *beta;
} asv_lde_desc_t;
int asv_net_add_lde_pool(asv_net_t *net, const asv_lde_desc_t *d);

/* Res2NetBlock of ECAPA-TDNN (reference model/ecapa_tdnn_xvector.py:42-75) as one op: the input view holds `branches + 1`
 * groups of 128 channels x_0 .. x_n; y_0 = x_0, y_i = BN(ReLU(TDNN_{[-d,0,d]}(y_{i-1} + x_i))) (i = 1: TDNN(x_1)); the output view
 * receives cat(y_0 .. y_n).  bf16 precision mode only (the other modes keep one TDNN layer per branch).
 * weight: host f32 [branches][128][128][2 d + 1] - the checkpoint's dense kernels, one after the other (masked positions are
 * ignored, as for asv_tdnn_desc_t); bias / scale / shift: host f32 [branches][128] (scale / shift = the folded eval BN). */
typedef struct asv_res2_desc {
  uint32_t struct_size;
  int32_t in_buf, in_ch_off;
  int32_t out_buf, out_ch_off;
  int32_t branches, dilation;
  const float *weight, *bias, *scale, *shift;
} asv_res2_desc_t;
int asv_net_add_res2(asv_net_t *net, const asv_res2_desc_t *d);

/* Elementwise: out = a (* seg_scale[segment]) (+ b) (+ c); any domain; views as above. */
typedef struct asv_eltwise_desc {
  uint32_t struct_size;
  int32_t channels;
  int32_t a_buf, a_ch_off, b_buf, b_ch_off, c_buf, c_ch_off;   /* b/c: -1 = absent */
  int32_t seg_scale_buf;                                        /* -1 = absent       */
  int32_t out_buf, out_ch_off;
  const float *scale, *shift;    /* optional host per-channel affine applied to `a` first    */
  int32_t act;                   /* activation applied to the sum (BasicBlock's final ReLU)  */
  int32_t seg_norm_buf;          /* utts-domain buffer [segments][mean(C) | std(C)] or -1: `a` becomes
                                    (a - mean) / std per segment first - InputSequenceNormalization,
                                    components.py:780-842; seg_norm_mode bit 0 = subtract mean, bit 1 = divide by std */
  int32_t seg_norm_mode;
  int32_t d_buf, d_ch_off;       /* optional SECOND output: out2 = out (as stored, i.e. rounded to the buffer's element  */
  int32_t out2_buf, out2_ch_off; /* type) + d - a following plain addition folded into this pass (ECAPA's running sum of
                                    block outputs, ecapa_tdnn_xvector.py:262-268); -1 / -1 = absent                        */
} asv_eltwise_desc_t;
int asv_net_add_eltwise(asv_net_t *net, const asv_eltwise_desc_t *d);

/* ResNetXvector's `x.unsqueeze(1)` (resnet_xvector.py:191): the input features [T][F] become a
 * one-channel grid buffer (domain: time_shift 0, width = feat_dim). */
typedef struct asv_grid_input_desc {
  uint32_t struct_size;
  int32_t out_buf;
  int32_t in_buf;                /* frames-domain buffer with feat_dim channels (0 = the raw input features) */
} asv_grid_input_desc_t;
int asv_net_add_grid_input(asv_net_t *net, const asv_grid_input_desc_t *d);

/* im2col gather between grids for strided convolutions (resnet.py stride-2 conv3x3 / conv1x1
 * downsample): out[(t', f')][k*C + c] = v[(stride*t' + dt[k], stride*f' + df[k])][c], zero outside, where v = in - or, with the
 * optional elementwise prologue (round 4), v = act(in * seg_scale[segment] + b) rounded to the buffers' element type: the block
 * output `relu(y * s + identity)` (resnet.py:70-85) of the block in front of a stride-2 stage is then never written in its own layout
 * (nothing but the gather reads it): one pass instead of two.  Same arithmetic, same rounding as asv_net_add_eltwise + a plain gather. */
typedef struct asv_im2col_desc {
  uint32_t struct_size;
  int32_t in_buf, out_buf, channels, n_taps, stride;
  int32_t dt[ASV_MAX_TAPS], df[ASV_MAX_TAPS];
  int32_t b_buf;                 /* optional addend, a whole buffer of in_buf's grid and width; -1 (or 0) = none */
  int32_t seg_scale_buf;         /* optional per-(segment, channel) scale: utts-domain buffer; -1 (or 0: a zero-initialised descriptor) = none */
  int32_t act;                   /* ASV_ACT_NONE | ASV_ACT_RELU, applied last                              */
} asv_im2col_desc_t;
int asv_net_add_im2col(asv_net_t *net, const asv_im2col_desc_t *d);

/* ResNetXvector's `x.reshape(B, C*F', T')` (resnet_xvector.py:193) in front of a pooling that weights FRAMES (attentive /
 * multi-head / multi-resolution / LDE, resnet_xvector.py:104-111; plain statistics pooling needs no copy - asv_pool_desc_t.per_bin):
 * out[t'][c*F' + f] = in[(t', f)][c], the reference's channel order.  `out_buf` lives on a SEQUENCE domain: a grid of width 1 and
 * pitch 1 at the input grid's time shift (asv_net_define_grid(net, shift, 1, 1)), on which frame-level layers (context within +-2
 * frames) and every pooling run as they do on the frames domain. */
typedef struct asv_grid_flatten_desc {
  uint32_t struct_size;
  int32_t in_buf, out_buf;       /* whole buffers: C channels on a (shift, F', pitch) grid -> C*F' channels on the (shift, 1, 1) grid */
} asv_grid_flatten_desc_t;
int asv_net_add_grid_flatten(asv_net_t *net, const asv_grid_flatten_desc_t *d);

/* Freezes the program; `out_buf` must be an utts-domain buffer: its first `embed_dim`
 * channels are the embedding. */
int asv_net_finalize(asv_net_t *net, int out_buf, int embed_dim);
int asv_net_embed_dim(const asv_net_t *net);
/* Human-readable dump of the program (for logs/tests); returns bytes written. */
int asv_net_describe(const asv_net_t *net, char *buf, size_t cap);

/* Embeds a packed batch:  feats  device  float32 [offsets[n_utts]][feat_dim] row-major,
 *                         offsets host   int32   [n_utts+1], offsets[0]=0, non-decreasing,
 *                         out    device  float32 [n_utts][embed_dim].
 * Utterances longer than max_chunk frames are split and averaged exactly as
 * framework.py:34-47 does (max_chunk <= 0: 10000).  Asynchronous on `stream`; `offsets`
 * is consumed before return.  Zero-length utterances are rejected (the reference asserts,
 * components.py:119). */
int asv_net_extract(asv_net_t *net, const float *feats, const int32_t *offsets, int n_utts,
                    float *out, int max_chunk, void *stream);

/* Range status of the extractions since the last call (ASV_PREC_F32X with IEEE-half operand halves, the default split): bit
 * ASV_STATUS_HALF_RANGE is set when an activation beyond +-65504 (or a NaN) met the operand split - the affected embeddings are then
 * wrong (the f32 reference has no such limit) and the batch should be re-extracted on a net created with ASV_FLAG_X3_SPLIT_BF16
 * (bf16 halves: the whole f32 exponent range, ~6e-6 relative).  Waits for `stream`, returns the bits in *status and clears them. */
#define ASV_STATUS_HALF_RANGE 1u
int asv_net_status(asv_net_t *net, unsigned *status, void *stream);

/* The same without the wait, for pipelined callers (the extraction scripts): enqueues on `stream` the copy of the status bits into
 * *host_status (the caller's word, page-locked for the copy to be asynchronous; valid once work enqueued behind it on `stream` is
 * known to have finished - e.g. the event the caller records behind its own result copy) and the clearing of the device word.
 * One status word per net: keep one batch per net in flight between two such calls. */
int asv_net_status_async(asv_net_t *net, unsigned *host_status, void *stream);

/* Diagnostic: launches of a kernel family by this process since the library was loaded (tests assert with it that the kernel they
 * mean to exercise is the one that ran - no silent fall-back onto another tile).  which: ASV_KERNEL_* below; unknown ids return 0. */
#define ASV_KERNEL_TDNN_P8 1     /* kernels_tdnn_p8.hip: 256 x 256 tiles, both operands through LDS-DMA */
#define ASV_KERNEL_TDNN_P8X 3    /* kernels_tdnn_p8x.hip: the same structure for the f32x mode (f32 rows split in registers) */
#define ASV_KERNEL_TDNN_BIG3 2   /* kernels_tdnn_v3.hip: 128 x 256 tiles, window through LDS, weight fragments from L2 */
#define ASV_KERNEL_TDNN_CHAINM 4 /* kernels_tdnn_chainm.hip: the f32x layer chain with its correction products on the scaled 8-bit instruction */
#define ASV_KERNEL_TDNN_X3M 5    /* kernels_tdnn_x3m.hip: the f32x wide-layer kernel in the same form */
#define ASV_KERNEL_TDNN_X3M_IMAGE 6 /* launches of that kernel that wrote their output rows as images for an f32m reader (counted in 5 as well) */
unsigned long long asv_kernel_launch_count(int which);

/* Bytes of device memory currently held by the net (weights + activation arena). */
size_t asv_net_device_bytes(const asv_net_t *net);

/* Name + average device time (ms, hipEvent on the extract stream) of the dominant kernel
 * class of the most recent profiled extract.  asv_net_set_profiling(1) makes extract record
 * events around every launch (costs a few us per launch); enable = 2 additionally keeps one
 * row per program op instead of one per kernel class; enable = 3 records the GEMM launches only
 * (the roofline's dominant kernel) to keep the instrumentation out of the other launch gaps. */
int asv_net_set_profiling(asv_net_t *net, int enable);
typedef struct asv_kernel_time {
  char     name[48];
  int32_t  op_index;             /* program op the row belongs to; -1 = all ops of this kernel class */
  int32_t  launches;
  float    total_ms;
  double   flops;                /* algorithmic 2*MAC of those launches (active taps only) */
} asv_kernel_time_t;
int asv_net_get_profile(asv_net_t *net, asv_kernel_time_t *rows, int cap, int *n_rows);

/* ---- single-layer entry points (kernel parity tests, other callers) ----------------- */
/* One TDNN layer on a packed ragged batch.  x device [rows][in_ch] f32, y device
 * [rows][out_ch] f32; computed in `precision`; desc buffers/views are ignored. */
int asv_tdnn_forward(const asv_tdnn_desc_t *d, int precision, unsigned flags,
                     const float *x, const int32_t *offsets, int n_utts, float *y, void *stream);
/* StatisticsPooling on a packed ragged batch: x device [rows][channels] f32 ->
 * y device [n_utts][channels*(1+stddev)] f32. */
int asv_stats_pool_forward(const float *x, int channels, const int32_t *offsets, int n_utts,
                           int stddev, int unbiased, int var_mode, float eps, float *y, void *stream);

/* ---- scoring back-end --------------------------------------------------------------- */
/* In-place x <- (x - mean) if mean != NULL, then x <- x / ||x||_2 if normalize
 * (ivector-subtract-global-mean + ivector-normalize-length --scaleup=false,
 *  score/process.sh:181-203).  x device [n][dim] f32, mean device [dim] f32. */
int asv_length_norm(float *x, int n, int dim, const float *mean, int normalize, void *stream);
/* Column means of x device [n][dim] -> mean device [dim] (ivector-mean, process.sh:165,177). */
int asv_mean_vec(const float *x, int n, int dim, float *mean, void *stream);
/* Full cosine/dot score matrix S[i][j] = <enroll_i, test_j>; device f32, row-major
 * (ivector-compute-dot-products over all pairs, score/score.sh:82-97). */
int asv_dot_score_matrix(const float *enroll, int n_enroll, const float *test, int n_test, int dim,
                         float *scores, void *stream);
/* Trial-list scoring: scores[t] = <enroll[ei[t]], test[ti[t]]>; index arrays device int32. */
int asv_dot_score_trials(const float *enroll, const float *test, int dim, const int32_t *ei,
                         const int32_t *ti, int n_trials, float *scores, void *stream);
/* Kaldi-style PLDA scoring (score/pyplda/plda_base.py:93-136, the Python restatement of
 * Kaldi ivector/plda.cc that score/score.sh:99-121 runs through ivector-plda-scoring).
 *   asv_plda_transform: y = T (x - mean); then, by length_norm,
 *       0: nothing;  1: y *= sqrt(dim)/||y||  (simple_length_norm, plda_base.py:99-100);
 *       2: y *= sqrt(dim / sum_i y_i^2/(psi_i + 1/n))   (get_normalization_factor, 165-172)
 *     num_examples: device int32 [n] or NULL (= 1 each).
 *   asv_plda_llr_trials: log-likelihood ratio of plda_base.py:109-136 per trial, evaluated
 *     in float64 like the reference; enroll_n: device int32 [n_enroll rows] utterances behind
 *     each enrolment vector, or NULL (= 1). */
#define ASV_PLDA_NORM_NONE   0
#define ASV_PLDA_NORM_SIMPLE 1
#define ASV_PLDA_NORM_PSI    2
int asv_plda_transform(const float *x, int n, int dim, const float *mean, const float *transform,
                       const float *psi, const int32_t *num_examples, int length_norm, float *y,
                       void *stream);
int asv_plda_llr_trials(const float *enroll, const float *test, int dim, const float *psi,
                        const int32_t *enroll_n, const int32_t *ei, const int32_t *ti, int n_trials,
                        float *scores, void *stream);
/* Speaker-level enrolment (score/process.sh:156-167: `ivector-mean ark:spk2utt <vectors> ark:<spk means> ark,t:<num_utts>`):
 * means[g] = (sum of x[order[k]], offsets[g] <= k < offsets[g+1], added in that order in f32) * (float)(1 / n_g) - Kaldi's
 * AddVec / Scale sequence - and counts[g] = n_g, the num_utts that asv_plda_transform / asv_plda_llr_trials take
 * (score/score.sh:99-121 --num-utts).  order: device int32 [offsets[n_groups]], offsets: device int32 [n_groups + 1]. */
int asv_group_mean(const float *x, int n, int dim, const int32_t *order, const int32_t *offsets, int n_groups,
                   float *means, int32_t *counts, void *stream);
/* Two-covariance PLDA scorer (score/pyplda/gaussian-plda-scoring.py:23-50) per trial, in float64 like the reference:
 *   s = e^T L t + t^T L e + e^T G e + t^T G t + (e + t)^T c          (k = 0 as in the reference)
 * gamma (G) / lambda (L): HOST float64 [dim][dim], c: HOST float64 [dim] (CalculateVar's outputs, D x D algebra done once per
 * model on the host); enroll / test: device f32 vectors; ei / ti: device int32 trial indices; scores: device float64 [n_trials].
 * The quadratic and bilinear forms are four f64 GEMMs (vectors x G, vectors x L) + one wave per trial. */
int asv_two_cov_trials(const float *enroll, int n_enroll, const float *test, int n_test, int dim, const double *gamma,
                       const double *lambda, const double *c, const int32_t *ei, const int32_t *ti, int n_trials,
                       double *scores, void *stream);
/* Equal error rate (score/computeEER-like-Bosaris.py:50-91 semantics) of device scores with
 * device int32 labels (1 target / 0 non-target).  eer_percent / threshold are host outputs;
 * the call synchronises `stream`. */
int asv_eer(const float *scores, const int32_t *labels, int n, float *eer_percent, float *threshold,
            void *stream);
/* Score normalisation (score/ScoreNormalization.py:70-179; protocol recipe/voxcelebSRC/gather_results_from_epochs.sh:103-183).
 *   enroll_cohort device [n_enroll][n_cohort], test_cohort device [n_test][n_cohort]: scores of every enrolment / test
 *   vector against every cohort vector (e.g. from asv_dot_score_matrix); ei / ti / scores: the trials (device).
 *   top_n <= 0 or >= n_cohort: S-norm (all cohort scores, lines 70-109); otherwise AS-norm over the top_n largest
 *   cohort scores of each vector (111-179); cross_select != 0: statistics per trial over the other side's top
 *   cohort (139-148).  Mean and sample standard deviation (ddof = 1) in float64 like pandas;
 *   normed[t] = 0.5 * ((s - mu_e) / sd_e + (s - mu_t) / sd_t).  Asynchronous on `stream`. */
int asv_score_norm(const float *enroll_cohort, int n_enroll, const float *test_cohort, int n_test, int n_cohort,
                   const int32_t *ei, const int32_t *ti, const float *scores, int n_trials, int top_n,
                   int cross_select, float *normed, void *stream);

/* ---- PLDA training --------------------------------------------------
 * Statistics + EM of score/pyplda/plda_base.py:37-81, 227-300 in float64 on the device.  x: device f32 [n_rows][ldx]
 * embeddings; order (host int32 [n_rows]): row indices grouped by class; class_offsets (host int64 [n_classes+1]): class k
 * owns order[class_offsets[k] .. class_offsets[k+1]), classes in ascending size like the reference's sorted stats.
 * Outputs (host float64): mean [dim] (mean of the class means), within_var and between_var [dim][dim] after num_iters EM
 * iterations - what PldaEstimation.plda_write stores and Plda.from_covariances diagonalises. */
int asv_plda_train(const float *x, int ldx, int n_rows, int dim, const int *order, const long long *class_offsets, int n_classes,
                   int num_iters, double *mean_out, double *within_out, double *between_out, void *stream);

/* Sum [dim] and second moment X^T X [dim][dim] of device f32 vectors, accumulated in float64 (host outputs): the statistics of
 * PldaUnsupervisedAdaptor.add_stats (plda_base.py:360-367) and of ZCA whitening (score/whiten/train_ZCA_Whitening.py:46-47). */
int asv_scatter_f64(const float *x, int ldx, int n_rows, int dim, double *sum_out, double *xtx_out, void *stream);

/* The same plus sum_k n_k mu_k mu_k^T over the class means (order / class_offsets as in asv_plda_train, any class order): the
 * CovarianceStats of Kaldi's ivector-compute-lda, the LDA stage of the scoring chain (score/process.sh:218-229). */
int asv_class_scatter_f64(const float *x, int ldx, int n_rows, int dim, const int *order, const long long *class_offsets, int n_classes,
                          double *sum_out, double *xtx_out, double *class_scatter_out, void *stream);

/* ---- acoustic front-end -------------------------------------------
 * Kaldi-compatible log-mel filterbank features of packed waveforms on the device: what
 * torchaudio.compliance.kaldi.fbank computes in pytorch/libs/egs/kaldi_features.py:72-137 and kaldifeat::Fbank in
 * runtime/kaldifeat/csrc/feature-fbank.cc (the field names and defaults below are FbankOptions / FrameExtractionOptions /
 * MelBanksOptions of that code).  Samples are floats in the int16 value range (Kaldi WaveData); dither is not offered
 * (it is the one random step; extraction configs set it to 0). */
#define ASV_WINDOW_POVEY       0
#define ASV_WINDOW_HAMMING     1
#define ASV_WINDOW_HANNING     2
#define ASV_WINDOW_RECTANGULAR 3
#define ASV_WINDOW_SINE        4
#define ASV_WINDOW_BLACKMAN    5
typedef struct asv_fbank_opts {
  uint32_t struct_size;
  float   sample_rate;          /* 16000 */
  float   frame_length_ms;      /* 25    */
  float   frame_shift_ms;       /* 10    */
  float   preemph;              /* 0.97  */
  int32_t remove_dc_offset;     /* 1     */
  int32_t window_type;          /* ASV_WINDOW_POVEY */
  int32_t round_to_power_of_two;/* 1     */
  int32_t snip_edges;           /* 1     */
  int32_t num_bins;             /* 23    */
  float   low_freq;             /* 20    */
  float   high_freq;            /* 0: Nyquist; negative: offset from Nyquist */
  int32_t use_energy;           /* 0     */
  float   energy_floor;         /* 0: none */
  int32_t raw_energy;           /* 1     */
  int32_t htk_compat;           /* 0: energy first */
  int32_t use_log_fbank;        /* 1     */
  int32_t use_power;            /* 1     */
  /* MFCC (feature-mfcc.cc:78-150): num_ceps > 0 turns the output into num_ceps cepstra = DCT-II of the log mel energies
   * (always log of power; use_log_fbank / use_power are ignored), liftered; with use_energy C0 is replaced by the log
   * energy; htk_compat moves C0 / the energy last (C0 scaled by sqrt 2 when it is not the energy). */
  int32_t num_ceps;             /* 0: filterbank output */
  float   cepstral_lifter;      /* 22    */
  float   blackman_coeff;       /* 0.42 (ASV_WINDOW_BLACKMAN, feature-window.cc:47-49)                                   */
  float   vtln_warp;            /* 1: none.  VTLN warping of the mel bin edges (mel-computations.cc:20-89, 129-161)      */
  float   vtln_low, vtln_high;  /* 100 / -500 (negative: offset from Nyquist): the cut-offs of the piecewise-linear warp */
} asv_fbank_opts_t;
/* Frames an utterance of num_samples yields (feature-window.cc:71-114); -1 on bad options. */
long long asv_fbank_num_frames(const asv_fbank_opts_t *opts, long long num_samples);
/* wave: device floats, utterance u = samples [sample_offsets[u], sample_offsets[u+1]) (host int64 [n_utts+1]);
 * feats: device [sum of frames][dim] f32, utterances back to back in order; dim = num_bins + use_energy, or num_ceps. */
int asv_fbank(const asv_fbank_opts_t *opts, const float *wave, const long long *sample_offsets, int n_utts,
              float *feats, void *stream);
/* The same with 16-bit PCM samples in HBM (half the ingest; SURVEY.md 8(f) rank 2 "ship int16 PCM"). */
int asv_fbank_pcm16(const asv_fbank_opts_t *opts, const short *wave, const long long *sample_offsets, int n_utts,
                    float *feats, void *stream);
/* Kaldi apply-cmvn-sliding (pipeline/extract_xvectors_for_pytorch.sh:105-118 runs it with --cmn-window=300 --center=true
 * in front of the extractor): every frame minus the mean of its window, optionally divided by the window's standard
 * deviation; out must not alias feats. */
int asv_cmvn_sliding(const float *feats, float *out, const long long *frame_offsets, int n_utts, int dim, int cmn_window,
                     int min_window, int center, int norm_vars, void *stream);
/* Energy VAD of runtime/extractor/torch_asv_extractor.cc:14-62 (Kaldi compute-vad-decision) on column 0 of the
 * features: voiced[frame] = 0/1 (device bytes), voiced_counts[u] (HOST int64 [n_utts], filled before returning). */
int asv_vad_energy(const float *feats, const long long *frame_offsets, int n_utts, int dim, float energy_threshold,
                   float energy_mean_scale, int frames_context, float proportion_threshold, unsigned char *voiced,
                   long long *voiced_counts, void *stream);
/* Keeps the voiced rows, in order (select-voiced-frames; torch_asv_extractor.cc:104-108): out_offsets = host int64
 * [n_utts+1] running sum of voiced_counts; out: device [out_offsets[n_utts]][dim]. */
int asv_select_frames(const float *feats, const unsigned char *voiced, const long long *frame_offsets,
                      const long long *out_offsets, int n_utts, int dim, float *out, void *stream);
/* Per-utterance mean / variance normalisation of every column, in place (kaldi_features.py:11-66
 * InputSequenceNormalization: mean over the frames, unbiased std floored at eps).  frame_offsets: host int64 [n_utts+1]. */
int asv_cmvn(float *feats, const long long *frame_offsets, int n_utts, int dim, int mean_norm, int std_norm, float eps,
             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ASV_AMD_H */
