// Internal declarations shared by the HIP translation units of libasv_amd.so.
// gfx950 (MI355X / CDNA4) only: wave64, MFMA, 160 KiB LDS.  No other target is supported.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/asv_amd.h"

namespace asv {

// ---------------------------------------------------------------------------------------
// error plumbing: nothing throws across the C ABI
void set_error(const char *fmt, ...);

#define ASV_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::asv::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ASV_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

#define ASV_REQUIRE(cond, ...)                                                                \
  do {                                                                                        \
    if (!(cond)) {                                                                            \
      ::asv::set_error(__VA_ARGS__);                                                          \
      return ASV_EINVAL;                                                                      \
    }                                                                                         \
  } while (0)

// ---------------------------------------------------------------------------------------
// Every C-ABI entry point runs on the device that owns its handle / its first device pointer and leaves the calling
// thread's current HIP device as it found it, on every return path (a multi-GPU torch process calls these with tensors of
// cuda:1 while cuda:0 is current: a stray hipSetDevice would silently move its later allocations and streams).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  int enter(int device) {
    ASV_HIP_CHECK(hipGetDevice(&prev));
    if (prev != device) {
      ASV_HIP_CHECK(hipSetDevice(device));
      switched = true;
    }
    return ASV_OK;
  }
  // the device that owns `dev_ptr` (device or managed memory only)
  int enter_owner(const void *dev_ptr, const char *what) {
    hipPointerAttribute_t attr;
    ASV_HIP_CHECK(hipPointerGetAttributes(&attr, dev_ptr));
    ASV_REQUIRE(attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged, "%s: pointer %p is not device memory", what, dev_ptr);
    return enter(attr.device);
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard() = default;
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define ASV_ON_DEVICE(dev) ::asv::DeviceGuard _asv_dev_guard; do { int _rc = _asv_dev_guard.enter(dev); if (_rc) return _rc; } while (0)
#define ASV_ON_OWNER(ptr, what) ::asv::DeviceGuard _asv_dev_guard; do { int _rc = _asv_dev_guard.enter_owner(ptr, what); if (_rc) return _rc; } while (0)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int64_t round_up64(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------
// Row layout of the frames domain (see DESIGN.md "Data layout in HBM").
//
// A batch of S utterance segments is laid out as
//     [HALO zero rows][seg 0][HALO zero rows][seg 1] ... [seg S-1][HALO zero rows][zero fill to R_pad]
// so a tap that reaches up to HALO frames across a segment edge reads zeros: exactly the
// per-layer zero padding of TdnnAffine (components.py:116-117), with no masks in the GEMM
// main loop.  Every producer writes zeros into gap rows (row_valid bit clear).
constexpr int kHalo = 4;           // max |tap offset| supported (ECAPA dilation 4)
constexpr int kRowTile = 256;      // rows are padded to a multiple of the largest GEMM M tile
constexpr int kChanAlign = 16;     // channel pitch granularity (elements)

// ---------------------------------------------------------------------------------------
// GEMM kernel parameters (one TDNN layer), passed by value.
struct TdnnKernelParams {
  const void *x;        // input rows, element type = activation type; already offset to in_ch_off
  const void *x2;       // optional second input (added), or nullptr
  const void *w;        // packed weights [cout_pad][n_taps][cin_pad], element type = activation type
  const float *bias;    // [cout_pad] (zeros when absent)
  const float *scale;   // [cout_pad] or nullptr
  const float *shift;   // [cout_pad] or nullptr
  const float *seg_bias;   // [segments][ld_seg] or nullptr
  const float *seg_scale;  // [segments][ld_segscale] or nullptr
  const void *res;      // residual rows (activation type) or nullptr
  void *y;              // output rows
  const int32_t *row_seg;     // [rows] segment id or -1 (gap)
  const uint32_t *row_valid;  // [rows/32] bit r%32 set <=> row r holds a real frame
  // fused statistics pooling (optional): per (row tile, channel) partial sums of the valid rows,
  // segmented by utterance; see kernels_tdnn.hip
  float *pool_partial;  // [rows/128 half-tiles][pool_slots][3 (sum (u-pv), sum (u-pv)^2, pivot pv)][ld_partial]
  int pool_slots;
  const void *zero16;   // >= 16 bytes of device zeros: source of masked direct-to-LDS loads
  const void *wfrag;    // bf16 weights in MFMA-fragment order [n_frag32][tap][chunk64][k_group][lane][8] or nullptr;
                        // pooled-domain layers (kernels_utts.hip): the bf16 'hi' halves of the f32 weights, [cout_pad][cin_pad]
  const void *wconv;    // 3x3 grid convolutions (kernels_conv2d.hip): bf16 weights in THAT kernel family's fragment order [tap][k-group][n-frag][lane][8]
                        // - its own pointer, so that neither family can ever be handed the other's layout - or nullptr
  const void *wlo;      // pooled-domain layers and the f32x frame kernel: the bf16 'lo' halves (w - hi), same layout as wfrag, or nullptr
  const void *wx3p;     // f32x 8-phase kernel (kernels_tdnn_p8x.hip): [cout_pad][tap][chunk32][hi 32 | lo 32] 16-bit halves of w * 2^s, or nullptr
  int x_image, y_image; // f32m form: the input rows are / the output rows become IMAGES - per (row, 32-channel group) the 128 bytes [hi halves | x_lo8 | x_hi8]
                        // the readers' window conversion would make of the f32 values (kernels_tdnn_x3m.hip's epilogue; run_ops decides per buffer and run)
  const void *w8;       // f32m form (kernels_tdnn_x3m.hip): 8-bit fragments of w * 2^s, planes [w_lo8][w_hi8] (pack_tdnn_weight_mx8), or nullptr
  // split-K (small-M layers: the pooled domain): blockIdx.y walks `ksplit` slices of the channel
  // chunks, raw f32 accumulators go to partial[slice][rows][ld_partial]; a second kernel sums
  // the slices in order and applies the epilogue (deterministic, no atomics)
  float *partial;
  int ksplit, ld_partial;
  int ldx, ldx2, ldy, ldres, ld_segbias, ld_segscale;   // pitches in elements
  int rows;             // padded row count (multiple of kRowTile)
  int cin_pad;          // multiple of kChanAlign
  int cout_store;       // columns [0, cout_store) of y are written (zeros beyond out_ch)
  int n_taps;
  int taps[ASV_MAX_TAPS];
  int act1, act2, affine_first;
  // pooled-domain kernel, last layer of a program whose utterances are all one chunk: also write the caller's result
  // [utterance][final_ld] with the frame-weighted-mean arithmetic of combine_kernel ((len * e) / len), saving that launch
  float *final_out; int final_ld; const int32_t *final_len;
  int row_begin, row_count;   // variant-3 kernel: the launch covers rows [row_begin, row_begin + row_count) (0, 0 = all rows)
  int tune;             // experiment knobs of the variant-3 kernel (tools/gemm_ablate): priorities / start stagger
  int big_one_per_cu;   // variant-3 kernel: 256x256 tiles, one workgroup per CU (default: 128x256, two per CU)
  int halo;             // max |tap offset| of this layer (selects the window size of the 128x128 kernel)
  int x3_et, x3_terms;  // f32x kernel (kernels_tdnn_x3.hip): 16-bit type of the operand halves (ET_BF16 / ET_F16) and which products run -
                        // bit 0: w_hi x_hi, bit 1: w_hi x_lo, bit 2: w_lo x_hi (7 = the f32-grade mode; the others are the measured
                        // "why not two matrix instructions" variants, LABLOG.md "Precision modes")
  int x3_tile;          // f32x kernel: 0 = pick the tile rows from the batch size, 128 = ASV_FLAG_X3_TILE128
  float w_unscale;      // f32x kernel: the accumulators are multiplied by this (1 / the power of two the host scaled the weights by)
  int et;               // ET_*: element type of x / x2 / res / y rows and of the packed weights (the launchers without an `et` argument read it)
  uint32_t *status;     // f32x kernels: device word, bit ASV_STATUS_HALF_RANGE is OR-ed in when an operand's half split overflowed (or nullptr)
};

struct PoolKernelParams {
  const void *x;  int ldx;  int channels;
  const int32_t *seg_row0;   // [segments] first row of the segment
  const int32_t *seg_len;    // [segments] rows of the segment
  float *out; int ld_out;    // utts-domain row pitch
  int stddev, unbiased, var_mode; float eps;
  // grid (2-D) inputs: `groups` frequency bins are pooled separately over time; bin g reads rows
  // row0 + g + k*row_stride (k < len/row_stride) and writes columns [g*C*(1+stddev), ...) of the row
  int row_stride, groups;
  // mean-only pooling of long segments (the SE squeeze over a 2-D map) in two steps: every kPoolChunkRows rows of a segment
  // one workgroup, sums to chunk_partial[(segment * chunks + chunk)][ld_chunk], then a finish kernel.  chunks = 0: one pass.
  float *chunk_partial; int chunks, ld_chunk;
};
constexpr int kPoolChunkRows = 2048;     // a property of the kernel, never of the batch: an utterance's mean does not depend on its neighbours

// Element type of frames-domain storage and of the matrix operands: the launchers' `et` argument (a former `bool bf16`
// converts to the first two values).
enum : int { ET_F32 = 0, ET_BF16 = 1, ET_F16 = 2 };

// launchers (kernels_*.hip).  et: element type of the activations (ET_*).
int launch_tdnn_mfma(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s);
int launch_tdnn_ref(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s);
int launch_utts_gemm(const TdnnKernelParams &p, int rows_valid, bool split, hipStream_t s);
// kernels_conv2d.hip: 3x3 grid convolutions with 32 / 64 channels (weights in p.wfrag, [tap][k-group][n-frag][lane][8])
bool grid_conv_narrow_supported(const TdnnKernelParams &p, int et);
size_t grid_conv_frag_elems(int cin_pad, int cout_pad32, int n_taps = 9);
bool grid_conv_s2d_supported(const TdnnKernelParams &p, int et);      // 128 (4 phases x 32) -> 64 channels, 4 backward taps
int launch_grid_conv_s2d(const TdnnKernelParams &p, hipStream_t s);
int launch_grid_conv_narrow(const TdnnKernelParams &p, hipStream_t s);
bool grid_conv_wide_supported(const TdnnKernelParams &p, int et);      // the C = 128 / 256 stages (same fragment order)
int launch_grid_conv_wide(const TdnnKernelParams &p, hipStream_t s);
bool grid_conv_c1_supported(const TdnnKernelParams &p, int et, int in_ch);
int launch_grid_conv_c1(const TdnnKernelParams &p, hipStream_t s);
// kernels_conv2d_x3.hip: the grid-domain layers of the f32x precision mode (f32 rows, three 16-bit matrix instructions per product);
// weights in p.wconv as [chunk32][tap][k-group][n-fragment][hi | lo][lane][8], scaled by 1 / p.w_unscale
bool grid_conv_x3_shape_ok(int cin_pad, int cout_store);
size_t grid_conv_x3_frag_elems(int cin_pad, int cout_store, int n_taps);
void pack_grid_conv_x3_frags(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps, int cin_pad, int cout_store,
                             int et, float scale, uint16_t *dst);
bool grid_conv_x3_supported(const TdnnKernelParams &p);
int launch_grid_conv_x3(const TdnnKernelParams &p, hipStream_t s);
int launch_splitk_epilogue(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s);
// 256-channel tiles of the bf16 frame-layer kernel (kernels_tdnn_v3.hip): weights are padded to kBigTileN output channels
constexpr int kBigTileN = 256;
// variant 3: feature window via LDS-DMA ring, weight fragments straight from L2 (kernels_tdnn_v3.hip)
bool tdnn_big3_supported(const TdnnKernelParams &p, int et, bool out_f32);
int launch_tdnn_big3(const TdnnKernelParams &p, hipStream_t s);
int launch_tdnn_big3_variant(const TdnnKernelParams &p, int variant, hipStream_t s);
// 256 x 256 tiles, both operands through LDS-DMA, four phases per K-tile with counted waits (kernels_tdnn_p8.hip); weights in p.w
bool tdnn_p8_supported(const TdnnKernelParams &p, int et, bool out_f32);
int launch_tdnn_p8(const TdnnKernelParams &p, hipStream_t s);
int launch_tdnn_p8_variant(const TdnnKernelParams &p, int variant, hipStream_t s);
// the same structure for the f32x mode: f32 rows split in registers, three matrix instructions per product (kernels_tdnn_p8x.hip); weights in p.wx3p
bool tdnn_p8x_supported(const TdnnKernelParams &p);
int launch_tdnn_p8x(const TdnnKernelParams &p, hipStream_t s);
void pack_tdnn_weight_x3p(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps, int cout_pad, int cin_pad,
                          int et, float scale, uint16_t *dst);
// tdnn -> [1-tap 512 -> 512]* -> 1-tap + fused statistics pooling in one kernel, the 128 x 512 intermediate tiles resident in
// LDS (kernels_tdnn_chain.hip).  Weight fragments as for the variant-3 kernel.
constexpr int kChainWidth = 512;
struct TdnnChainLayer {
  const void *wfrag; const float *bias, *scale, *shift;   // scale / shift may be nullptr (1, 0)
  int relu, cout_pad;
  const void *wlo; float w_scale;                          // f32x chain (kernels_tdnn_chainx.hip): lo halves, and the power of two both halves carry
  const void *w8;                                          // f32m chain (kernels_tdnn_chainm.hip): 8-bit fragments, planes [w_lo8][w_hi8] (pack_tdnn_weight_mx8)
};
struct TdnnChainParams {
  const void *x; int ldx, rows, cin_pad, n_taps; int taps[ASV_MAX_TAPS];     // input of the first layer (bf16 rows)
  TdnnChainLayer first; int n_mid; TdnnChainLayer mid[2]; TdnnChainLayer last;
  float *pool_partial; int pool_slots, ld_partial; const int32_t *row_seg;    // as in TdnnKernelParams
  unsigned long long *dbg;      // developer aid (ASV_AMD_CHAIN_DBG=1): [workgroup][wave][32] s_memtime stamps at the phase boundaries, or nullptr
  int dbg_fine;                 // ASV_AMD_CHAIN_DBG >= 3: also stamps inside the first pooling epilogue of every wave
  int et;                       // ET_BF16 / ET_F16: element type of the rows and of every layer's weight fragments
  int min_seg_len;              // shortest utterance of the batch in frames (the 4-wave kernel needs >= 32: at most one seam per 32-frame fragment)
  uint32_t *status;             // f32x chain: as TdnnKernelParams::status
  int n128, n_tail, tail_rows;  // 16-bit chain: the launch's tile plan (chain_tile_plan): n128 tiles of 128 frames, then n_tail of tail_rows (96 | 64)
  int row_base, tile_base;      // set by the launcher per kernel launch: first row / first partial-moment block of that launch
  int x_image;                  // f32m chain: the first layer's input rows are images (TdnnKernelParams::x_image): no window conversion
  int abl;                      // f32m chain, developer aid with ASV_AMD_CHAIN_DBG (ASV_AMD_CHAINM_ABL; results are garbage): bit 0 no in-loop conversion, 1 no in-loop window DMA, 2 no chunk barrier (these three: garbage results), 3 no alternating issue priority (results unchanged)
};
// How the 16-bit chain kernel cuts `rows` (a multiple of 128) into tiles.  One workgroup per CU (160 KiB of LDS), so:
//   * a batch that does not fill ONE round of the chip's CUs in 128-frame tiles runs in the smallest tile - 64 or 96 frames -
//     that still fits one round: the same rows on more CUs, each for 1/2 or 3/4 of the time (a single utterance: 2 tiles -> 4);
//   * larger batches run in 128-frame tiles throughout.  Cutting only their last, partly filled round into 96-frame tiles (a second
//     launch) was measured at configs[1]'s 256 utterances - 408 tiles on 256 CUs -: +2 % on one stream, -2 % with two engines on two
//     streams (the other stream's workgroups fill the idle CUs anyway, the extra launch is pure cost): opt-in, ASV_AMD_CHAIN_TAIL=2
//     (profiles/r4f_chain_tail_ab.txt).
// Partial-moment block t covers rows [row0(t), row0(t) + rows_of(t)); rows beyond `rows` (the last tile may overhang) count as gaps.
struct ChainTilePlan {
  int n128 = 0, n_tail = 0, tail_rows = 0;
  int rows128() const { return n128 * 128; }
  int tiles() const { return n128 + n_tail; }
  int tile_of(int row) const { return row < rows128() ? row >> 7 : n128 + (row - rows128()) / tail_rows; }
  int row0(int t) const { return t < n128 ? t * 128 : rows128() + (t - n128) * tail_rows; }
  int rows_of(int t) const { return t < n128 ? 128 : tail_rows; }
};
ChainTilePlan chain_tile_plan(int rows, bool allow_tail);
int launch_tdnn_chain(const TdnnChainParams &p, hipStream_t s);
#ifdef ASV_WITH_ABLATION
// the same chain as four waves of 512 registers, pooling arithmetic inside the next unit's K loop (tools/kernels_tdnn_chain4.hip:
// measured 0 - 1 % slower; part of the developer build libasv_amd_dev.so only, `make dev`)
bool tdnn_chain4_supported(const TdnnChainParams &p);
int launch_tdnn_chain4(const TdnnChainParams &p, hipStream_t s);
#endif
// the same chain with f32-grade split products and the tiles resident as hi / lo half images; 64-row tiles (kernels_tdnn_chainx.hip)
int launch_tdnn_chainx(const TdnnChainParams &p, hipStream_t s);
// the same with the two correction products on the block-scaled 8-bit matrix instruction ("f32m", ASV_FLAG_X3_MX8; kernels_tdnn_chainm.hip)
int launch_tdnn_chainm(const TdnnChainParams &p, hipStream_t s);
// ... in 96-frame tiles (kernels_tdnn_chainm96.hip): 1.5 x the matrix work per byte of the weight stream that bounds the 64-frame kernel
int chainm96_tiles(int rows);
int launch_tdnn_chainm96(const TdnnChainParams &p, hipStream_t s);
// ECAPA Res2NetBlock as one kernel (kernels_res2.hip)
constexpr int kRes2Width = 128;
struct Res2KernelParams {
  const void *x; void *y; int ldx, ldy, rows;       // bf16 rows; x / y already offset to their channel views
  const void *wfrag;                                 // [branches][4 n-frag][3 taps][2 chunks][4 k-groups][lane][8] bf16
  const float *bias, *scale, *shift;                 // [branches][128]
  const uint32_t *row_valid;
  int branches, dilation;
  unsigned long long *dbg;      // developer aid (ASV_AMD_RES2_DBG=1): [workgroup][wave][16] s_memtime stamps, or nullptr
  int et;                       // ET_BF16 / ET_F16
};
int launch_res2_chain(const Res2KernelParams &p, hipStream_t s);
void pack_tdnn_weight_frags(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps,
                            int cout_pad, int cin_pad, uint16_t *dst, uint16_t *dst_lo = nullptr, int et = ET_BF16, float scale = 1.0f);
// f32-grade split-bf16 kernel of the f32x precision mode (kernels_tdnn_x3.hip): f32 activations, hi / lo weight fragments
bool tdnn_x3_supported(const TdnnKernelParams &p);
bool tdnn_x3_pool_supported(const TdnnKernelParams &p);      // fused statistics pooling (128-row tiles, plain epilogue)
int launch_tdnn_x3(const TdnnKernelParams &p, hipStream_t s);
// the same layers with the correction products on the block-scaled 8-bit matrix instruction ("f32m"; kernels_tdnn_x3m.hip)
bool tdnn_x3m_supported(const TdnnKernelParams &p);
int launch_tdnn_x3m(const TdnnKernelParams &p, hipStream_t s);
bool tdnn_x3m_image_in_supported(const TdnnKernelParams &p);
bool tdnn_x3m_image_out_supported(const TdnnKernelParams &p);
size_t tdnn_weight_frag_elems(int cout_pad, int cin_pad, int n_taps);
int launch_stats_pool(const PoolKernelParams &p, int segments, int et, hipStream_t s);
// second half of the fused pooling: adds each segment's half-tile partials in row order, adds the BN shift
// back to the mean and writes mean || std like stats_pool_kernel
struct PoolFinishParams {
  const float *partial; int ld_partial, pool_slots;
  int lh_split;                  // partials of the chain kernel: [tile][slot][lh][3][ld], lh = bit 2 of the row index of the frames summed
  int tile_shift;                // log2 of the rows one partial covers: 7 (128-row tiles; 0 means 7) or 6 (the f32x chain's 64-row tiles)
  int rows_shift, tail_rows, n_shift;   // tail_rows > 0: behind row `rows_shift` the blocks n_shift + k cover `tail_rows` rows each (ChainTilePlan: 96)
  const int32_t *row_seg; int rows;
  const int32_t *seg_row0, *seg_len;
  const float *shift;            // per-channel BN shift that the producer left out (or nullptr)
  float *out; int ld_out, channels;
  int stddev, unbiased, var_mode; float eps;
};
int launch_pool_finish(const PoolFinishParams &p, int segments, hipStream_t s);
int launch_attentive_pool(const void *x, int ldx, const void *logits, int ldl, int channels,
                          const int32_t *seg_row0, const int32_t *seg_len, int segments, float eps,
                          float *out, int ld_out, int et, int group, int softplus2, const float *prior_logit, const float *prior_value,
                          hipStream_t s);
// row r of a segment is valid iff (r - row0) % pitch < width (frames domain: pitch = width = 1)
int launch_rowmap(const int32_t *seg_row0, const int32_t *seg_len, int segments, int rows, int pitch, int width,
                  int32_t *row_seg, uint32_t *row_valid, hipStream_t s);
// frames-domain feature rows [t][f] -> grid rows (t*pitch + f) with one channel (pitch of `out` = ldo)
int launch_grid_from_frames(const void *x, int ldx, int feat_dim, const int32_t *fr_row0, const int32_t *g_row0, const int32_t *g_row_seg,
                            const uint32_t *g_row_valid, int g_rows, int pitch, void *out, int ldo, int et, hipStream_t s);
// im2col gather between grid domains: out[(t',f')][k*C + c] = in[(stride*t'+dt_k, stride*f'+df_k)][c] or 0
struct Im2colParams {
  const void *in; void *out; int ldi, ldo, channels, n_taps, stride;
  int dt[ASV_MAX_TAPS], df[ASV_MAX_TAPS];
  const int32_t *in_row0, *in_len, *out_row0, *out_row_seg; const uint32_t *out_row_valid;
  int in_pitch, in_width, out_pitch, out_rows;
  // optional elementwise prologue: value = act(in * seg_scale[segment] + b), rounded to the element type (the arithmetic of eltwise_kernel)
  const void *b; int ldb; const float *seg_scale; int ld_segscale; int act;
};
int launch_im2col(const Im2colParams &p, int et, hipStream_t s);
// [rows (t', f)][C] on a grid -> [rows t'][c*F + f] on the sequence domain of the same time shift
int launch_grid_flatten(const void *in, int ldi, int channels, int width, int in_pitch, const int32_t *in_row0, const int32_t *out_row0,
                        const int32_t *out_row_seg, const uint32_t *out_row_valid, int out_rows, void *out, int ldo, int et, hipStream_t s);
int launch_pack_input(const float *feats, int feat_dim, const int32_t *seg_src0, const int32_t *seg_row0,
                      const int32_t *row_seg, int rows, void *x, int ldx, int et, hipStream_t s);
int launch_unpack_rows(const void *y, int ldy, int channels, const int32_t *seg_src0, const int32_t *seg_row0,
                       const int32_t *row_seg, int rows, float *out, int et, hipStream_t s);
int launch_combine(const float *seg_emb, int ld_seg, const int32_t *utt_seg0, const int32_t *utt_nseg,
                   const int32_t *seg_len, int n_utts, int embed_dim, float *out, hipStream_t s);
struct EltwiseKernelParams {
  const void *a, *b, *c; void *out;
  int lda, ldb, ldc, ldo, channels, rows;
  const float *scale, *shift;          // per-channel affine on a (or nullptr)
  const float *seg_scale; int ld_segscale;
  const int32_t *row_seg; const uint32_t *row_valid;   // nullptr in the utts domain
  int act;                             // activation applied to the final sum
  const float *seg_norm; int ld_segnorm, seg_norm_mode;   // [segments][mean(C) | std(C)]: a <- (a - mean) / std first
  const void *d; void *out2; int ldd, ldo2;              // second output: out2 = (out as stored) + d, or nullptr
};
int launch_lde_pool(const void *x, int ldx, int channels, int rows, const float *mu, const float *beta, int n_centres, float *weights,
                    const int32_t *seg_row0, const int32_t *seg_len, int segments, float *out, int ld_out, int et, hipStream_t s);
int launch_eltwise(const EltwiseKernelParams &p, int et, hipStream_t s);

// weight packing (host): dense checkpoint kernel -> [cout_pad][n_taps][cin_pad] in element type
void pack_tdnn_weight(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps,
                      int n_taps, int cout_pad, int cin_pad, int et, void *dst);
// f32 <-> bf16 / IEEE half on the host: host_convert.h

}  // namespace asv
