// Host runtime of libasv_amd.so: the layer-program builder (asv_net_*), weight packing,
// the per-batch segment/row planner and the launch sequencer behind asv_net_extract().
//
// Replaces, for the batched device path, what `for_extract_embedding` +
// `<Model>.extract_embedding` do one utterance at a time in the reference
// (libs/nnet/framework.py:12-55, model/xvector.py:77-98, model/ecapa_tdnn_xvector.py:403-426).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <vector>

#include <atomic>
#include "asv_internal.h"
#include "host_convert.h"

namespace asv {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

static uint16_t f32_to_h16_host(float f, int et) { return et == ET_F16 ? f32_to_f16_host(f) : f32_to_bf16_host(f); }
static float h16_to_f32_host(uint16_t h, int et) { return et == ET_F16 ? f16_to_f32_host(h) : bf16_to_f32_host(h); }

void pack_tdnn_weight(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps,
                      int cout_pad, int cin_pad, int et, void *dst) {
  const size_t n = (size_t)cout_pad * n_taps * cin_pad;
  if (et != ET_F32) memset(dst, 0, n * 2); else memset(dst, 0, n * 4);
  for (int co = 0; co < out_ch; ++co)
    for (int t = 0; t < n_taps; ++t) {
      const int k = taps[t] - left_ctx;
      const size_t base = ((size_t)co * n_taps + t) * cin_pad;
      for (int ci = 0; ci < in_ch; ++ci) {
        const float v = w[((size_t)co * in_ch + ci) * tot_ctx + k];
        if (et != ET_F32) reinterpret_cast<uint16_t *>(dst)[base + ci] = f32_to_h16_host(v, et);
        else reinterpret_cast<float *>(dst)[base + ci] = v;
      }
    }
}

size_t tdnn_weight_frag_elems(int cout_pad, int cin_pad, int n_taps) {
  return (size_t)cout_pad * n_taps * round_up(cin_pad, 64);
}

// MFMA-fragment order for kernels_tdnn_v3.hip: [n_frag32][tap][chunk64][k_group16][lane][8];
// lane = (k half lh, channel lr): channel n_frag*32 + lr, k = chunk*64 + k_group*16 + lh*8 + e.
// et: ET_BF16 / ET_F16 = the 16-bit type of the fragments.  dst_lo (the f32x mode): the second halves w * scale - hi;
// `scale` (a power of two, exact) multiplies every weight first - the half-precision split uses it to lift the weights of a
// layer into the upper part of the half range, where hi AND lo are normal numbers (22 significant bits together); the
// kernel's epilogue multiplies the accumulator by 1 / scale (exact).
void pack_tdnn_weight_frags(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps,
                            int cout_pad, int cin_pad, uint16_t *dst, uint16_t *dst_lo, int et, float scale) {
  const int nchunks = round_up(cin_pad, 64) / 64;
  memset(dst, 0, tdnn_weight_frag_elems(cout_pad, cin_pad, n_taps) * 2);
  if (dst_lo) memset(dst_lo, 0, tdnn_weight_frag_elems(cout_pad, cin_pad, n_taps) * 2);
  for (int co = 0; co < out_ch; ++co) {
    const int nf = co / 32, lr = co % 32;
    for (int t = 0; t < n_taps; ++t) {
      const int k = taps[t] - left_ctx;
      for (int ci = 0; ci < in_ch; ++ci) {
        const int c = ci / 64, kg = (ci % 64) / 16, lh = (ci % 16) / 8, e = ci % 8;
        const size_t idx = ((((size_t)nf * n_taps + t) * nchunks + c) * 4 + kg) * 512 + (size_t)(lh * 32 + lr) * 8 + e;
        const float v = w[((size_t)co * in_ch + ci) * tot_ctx + k] * scale;
        dst[idx] = f32_to_h16_host(v, et);
        if (dst_lo) dst_lo[idx] = f32_to_h16_host(v - h16_to_f32_host(dst[idx], et), et);      // f32x mode: w = hi + lo
      }
    }
  }
}

// Plain order for kernels_tdnn_p8x.hip (the weight half-tiles arrive by LDS-DMA, 128-byte rows): [cout_pad][tap][chunk32][hi 32 | lo 32],
// the same halves of w * scale as pack_tdnn_weight_frags writes (the two kernels give the same bits)
void pack_tdnn_weight_x3p(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps, int cout_pad, int cin_pad,
                          int et, float scale, uint16_t *dst) {
  const int nchunks = cin_pad / 32;
  memset(dst, 0, (size_t)cout_pad * n_taps * nchunks * 64 * 2);
  for (int co = 0; co < out_ch; ++co)
    for (int t = 0; t < n_taps; ++t) {
      const int k = taps[t] - left_ctx;
      for (int ci = 0; ci < in_ch; ++ci) {
        const size_t idx = (((size_t)co * n_taps + t) * nchunks + ci / 32) * 64 + ci % 32;
        const float v = w[((size_t)co * in_ch + ci) * tot_ctx + k] * scale;
        dst[idx] = f32_to_h16_host(v, et);
        dst[idx + 32] = f32_to_h16_host(v - h16_to_f32_host(dst[idx], et), et);
      }
    }
}

// 8-bit fragments of the f32m form (kernels_tdnn_chainm.hip / kernels_tdnn_x3m.hip, the scaled 8-bit matrix instruction): for the halves
// hi = half(w * scale), lo = w * scale - hi that pack_tdnn_weight_frags splits into, two planes of
// [32-channel output fragment][tap][32-channel input group][lane = (lh, output channel lr)][16]: plane 0 = e4m3(lo 2^6) (|lo| <= 2^-11 |hi|,
// hi < 2^14 under x3_weight_scale), plane 1 = e4m3(hi 2^-6).  Byte q of lane half lh = input channel 32 group + (q < 8 ? 8 lh + q :
// 16 + 8 lh + q - 8) - the channels the lane's two half fragments of the group hold, in their order, so that a kernel may also MAKE plane 1
// from those half fragments in registers (v_cvt_scalef32_pk_fp8_f16: kernels_tdnn_x3m.hip and the 96-frame chain do; the 64-frame chain
// fetches it - 24 conversions per K step cost its waves more issue time than two more 16-byte loads, profiles/r6s_*).  The kernels' block
// scales undo the 2^6 / 2^-6.
size_t tdnn_weight_mx8_plane_bytes(int cout_pad, int cin_pad, int n_taps) { return (size_t)(cout_pad / 32) * n_taps * ((cin_pad + 31) / 32) * 1024; }
size_t tdnn_weight_mx8_bytes(int cout_pad, int cin_pad, int n_taps) { return 2 * tdnn_weight_mx8_plane_bytes(cout_pad, cin_pad, n_taps); }
void pack_tdnn_weight_mx8(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps, int cout_pad, int cin_pad,
                          float scale, uint8_t *dst) {
  const int ngroups = (cin_pad + 31) / 32;                    // (a last, partial group is zero-padded: e4m3(0) = 0)
  memset(dst, 0, tdnn_weight_mx8_bytes(cout_pad, cin_pad, n_taps));
  const size_t plane = tdnn_weight_mx8_plane_bytes(cout_pad, cin_pad, n_taps);
  for (int co = 0; co < out_ch; ++co) {
    const int nf = co / 32, lr = co % 32;
    for (int t = 0; t < n_taps; ++t) {
      const int k = taps[t] - left_ctx;
      for (int ci = 0; ci < in_ch; ++ci) {
        const int g = ci / 32, r = ci % 32, kg = r / 16, lh = (r % 16) / 8, q = kg * 8 + r % 8;
        const float v = w[((size_t)co * in_ch + ci) * tot_ctx + k] * scale;
        const float hi = f16_to_f32_host(f32_to_f16_host(v));
        const size_t at = (((size_t)nf * n_taps + t) * ngroups + g) * 1024 + (size_t)(lh * 32 + lr) * 16 + q;
        dst[at] = f32_to_e4m3_host((v - hi) * 64.0f);
        dst[plane + at] = f32_to_e4m3_host(hi * (1.0f / 64.0f));
      }
    }
  }
}

// Power of two that lifts the largest weight of a layer to [2^13, 2^14) (the half-precision split of the f32x mode): every
// weight down to 2^-15 of the largest keeps a normal lo half; the products grow by the same factor, far inside f32.
static float x3_weight_scale(const float *w, size_t n) {
  float mx = 0.0f;
  for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
  if (!(mx > 0.0f) || !std::isfinite(mx)) return 1.0f;
  int e = 0;
  (void)std::frexp(mx, &e);                       // mx = m * 2^e, 0.5 <= m < 1
  return std::ldexp(1.0f, 14 - e);
}

namespace {

struct Buffer {
  int domain, channels, ld;
};

// Row domains.  0 = frames (1 row per frame, HALO-row gaps), 1 = utts (1 row per segment),
// >= 2 = grids for the 2-D ResNet trunk: a segment of T frames owns ceil(T / 2^shift) x pitch rows
// ((time, frequency), frequency fastest, `width` valid rows per frame) with pitch+2 gap rows around it.
struct Domain {
  int kind;                // ASV_DOMAIN_FRAMES / ASV_DOMAIN_UTTS / 2 (grid)
  int shift = 0, width = 1, pitch = 1;
  int halo() const { return kind == ASV_DOMAIN_FRAMES ? kHalo : (kind == ASV_DOMAIN_UTTS ? 0 : pitch + 1); }
  int gap() const { return kind == ASV_DOMAIN_FRAMES ? kHalo : pitch + 2; }
};

enum OpKind { OP_TDNN = 0, OP_POOL = 1, OP_ATTPOOL = 2, OP_ELTWISE = 3, OP_GRID_INPUT = 4, OP_IM2COL = 5, OP_LDE = 6, OP_RES2 = 7, OP_FLATTEN = 8 };

struct Op {
  OpKind kind;
  // common views
  asv_tdnn_desc_t tdnn;          // host pointers nulled after packing
  asv_pool_desc_t pool;
  asv_attpool_desc_t att;
  asv_eltwise_desc_t elt;
  asv_grid_input_desc_t gin;
  asv_im2col_desc_t i2c;
  asv_grid_flatten_desc_t flat;
  asv_lde_desc_t lde;            // mu / beta live in `scale` / `shift` on the device
  asv_res2_desc_t res2;          // fragments in `wfrag`, per-branch constants in bias / scale / shift
  // device parameters
  void *w = nullptr;
  void *wfrag = nullptr;         // fragment-ordered copy for the variant-3 kernel (bf16 frame layers); pooled layers: bf16 hi halves
  void *wconv = nullptr;         // 3x3 trunk convolutions with 32 / 64 / 128 / 256 channels: fragment order of kernels_conv2d.hip
  void *wlo = nullptr;           // pooled layers in bf16 precision mode: bf16 lo halves (kernels_utts.hip)
  void *wx3p = nullptr;          // f32x mode, wide frame layers with the plain epilogue: [hi | lo] rows for kernels_tdnn_p8x.hip
  void *w8 = nullptr, *w8_fold = nullptr;      // f32m form (ASV_FLAG_X3_MX8): 8-bit fragments of the layer / of its folded weights (pack_tdnn_weight_mx8)
  float *bias = nullptr, *scale = nullptr, *shift = nullptr;
  int cin_pad = 0, cout_pad = 0, cout_store = 0;
  // chain candidates (16-bit modes, 1-tap layers reading a 512-channel buffer): host copies kept until asv_net_finalize, which
  // folds the eval BatchNorm of the PREVIOUS chain layer into this layer's weights and bias (see the chain pass there)
  std::vector<float> host_w, host_bias, host_scale, host_shift;
  void *wfrag_fold = nullptr;    // fragment-ordered W diag(s_prev) ... (f32x: its hi halves, wlo_fold the lo halves, w_scale_fold their power of two)
  void *wlo_fold = nullptr;
  float w_scale_fold = 1.0f;
  float *bias_fold = nullptr;    // ... and b + W t_prev
  float w_scale = 1.0f;          // f32x mode, half-precision split: the power of two the fragment weights were multiplied by
  bool utts = false;             // op runs in the utts domain (always f32)
  bool has_affine = false;
  int fused_pool = -1;           // TDNN op: index of the statistics-pooling op folded into its epilogue
  int chain_last = -1;           // TDNN op heading a chain (kernels_tdnn_chain.hip): index of the chain's last layer
  int image_reader = -1;         // f32m: the ONE op that reads this TDNN op's whole output buffer, as its input rows - the buffer may hold images
                                 // (TdnnKernelParams::y_image) in a run where both ops go to image-capable kernels (run_ops)
  bool skipped = false;          // pool op executed by its producer
};

struct DevMem {
  void *ptr = nullptr;
  size_t cap = 0;
};

const char *kKernelNames[] = {"tdnn_gemm", "stats_pool", "attentive_pool", "eltwise", "rowmap", "pack_input", "combine", "grid_gather", "utts_gemm"};
enum { K_TDNN = 0, K_POOL, K_ATT, K_ELT, K_ROWMAP, K_PACK, K_COMBINE, K_GATHER, K_UTTS, K_COUNT };   // K_UTTS: affine layers of the pooled domain

}  // namespace
}  // namespace asv

using namespace asv;

struct asv_net {
  int device = 0, precision = 0, feat_dim = 0;
  unsigned flags = 0;
  bool finalized = false;
  int out_buf = -1, embed_dim = 0;
  std::vector<Buffer> bufs;
  std::vector<Domain> domains;
  std::vector<Op> ops;
  std::vector<void *> weight_allocs;
  size_t weight_bytes = 0;
  // per-call state
  std::vector<DevMem> arena;             // one region per buffer
  DevMem meta_dev;                       // int32 metadata (segments etc.)
  DevMem rowmeta_dev;                    // row_seg / row_valid for both domains
  DevMem splitk_dev;                     // split-K partial accumulators
  DevMem poolpart_dev;                   // fused-pooling partial moments
  DevMem lde_dev;                        // LDE pooling: per-row centre weights [rows][64]
  void *zero_page = nullptr;             // 256 zero bytes (masked direct-to-LDS loads); bytes 128..131: the status word (asv_net_status)
  void *meta_host = nullptr;             // pinned staging
  size_t meta_host_cap = 0;
  hipEvent_t meta_copied = nullptr;      // H2D of meta_host finished
  bool meta_inflight = false;
  // plan cache: a batch with the same utterance lengths as the previous one reuses the uploaded segment
  // tables and row maps (steady-state extraction loops over equally shaped batches)
  std::vector<int32_t> cached_offsets;
  int cached_max_chunk = -1;
  bool cache_valid = false;
  void *plan_cache = nullptr;                 // PlanCache (defined with the launch sequence)
  void (*plan_cache_free)(void *) = nullptr;
  // profiling
  int profiling = 0;               // 0 off, 1 per kernel class, 2 per op, 3 frame-level GEMM launches only, 4 = 3 with ONE event
                                   // pair around every run of consecutive frame-level GEMM launches
  struct Stamp { int kclass; int op; double flops; hipEvent_t a, b; int launches = 1; bool open = false; };
  std::vector<Stamp> stamps;
  std::vector<hipEvent_t> event_pool;

  bool frames_h16() const { return precision == ASV_PREC_BF16 || precision == ASV_PREC_F16; }     // 16-bit frames-domain storage
  int frames_et() const { return precision == ASV_PREC_BF16 ? ET_BF16 : (precision == ASV_PREC_F16 ? ET_F16 : ET_F32); }
  int x3_et() const { return (flags & ASV_FLAG_X3_SPLIT_BF16) ? ET_BF16 : ET_F16; }                  // f32x: type of the operand halves
  int x3_terms() const { return 1 | ((flags & ASV_FLAG_X3_NO_XLO) ? 0 : 2) | ((flags & ASV_FLAG_X3_NO_WLO) ? 0 : 4); }
  bool x3() const { return precision == ASV_PREC_F32X; }        // f32 storage, split-bf16 matrix products
  bool x3_mx() const { return x3() && (flags & ASV_FLAG_X3_MX8) != 0 && x3_et() == ET_F16 && x3_terms() == 7; }      // "f32m": corrections on the scaled 8-bit instruction
  bool is_utts(int domain) const { return domains[domain].kind == ASV_DOMAIN_UTTS; }
  // one row per frame: the frames domain, or a width-1 / pitch-1 grid (the 2-D trunk's output flattened, asv_net_add_grid_flatten)
  bool is_sequence(int domain) const { return domains[domain].kind == ASV_DOMAIN_FRAMES || (domains[domain].kind == 2 && domains[domain].width == 1 && domains[domain].pitch == 1); }
  int dom_et(int domain) const { return is_utts(domain) ? ET_F32 : frames_et(); }                    // element type of a domain's rows
  size_t elem_size(int domain) const { return dom_et(domain) != ET_F32 ? 2 : 4; }
};

namespace {

std::atomic<unsigned long long> g_kernel_launches[7];       // asv_kernel_launch_count

// the range-status word of the f32x kernels lives behind the zero page's zeros (own 64-byte line; kernels only ever OR into it)
uint32_t *status_word(asv_net *net) { return reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(net->zero_page) + 128); }

int dev_upload(asv_net *net, const void *host, size_t bytes, void **out) {
  void *d = nullptr;
  ASV_HIP_CHECK(hipMalloc(&d, bytes));
  hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
    return ASV_EHIP;
  }
  net->weight_allocs.push_back(d);
  net->weight_bytes += bytes;
  *out = d;
  return ASV_OK;
}

int upload_padded(asv_net *net, const float *src, int n, int n_pad, float fill, float **out) {
  std::vector<float> tmp((size_t)n_pad, fill);
  if (src) memcpy(tmp.data(), src, (size_t)n * sizeof(float));
  void *d = nullptr;
  int rc = dev_upload(net, tmp.data(), tmp.size() * sizeof(float), &d);
  *out = reinterpret_cast<float *>(d);
  return rc;
}

int check_view(const asv_net *net, int buf, int ch_off, int ch, const char *what) {
  ASV_REQUIRE(buf >= 0 && buf < (int)net->bufs.size(), "%s: buffer id %d out of range", what, buf);
  ASV_REQUIRE(ch_off >= 0 && ch_off % kChanAlign == 0, "%s: channel offset %d must be a non-negative multiple of %d", what, ch_off, kChanAlign);
  ASV_REQUIRE(ch >= 1 && ch_off + ch <= net->bufs[buf].channels, "%s: view [%d,%d) exceeds buffer %d (%d channels)", what, ch_off, ch_off + ch, buf, net->bufs[buf].channels);
  return ASV_OK;
}

int ensure(DevMem &m, size_t bytes, hipStream_t s, bool zero) {
  if (bytes <= m.cap) return ASV_OK;
  if (m.ptr) {
    ASV_HIP_CHECK(hipStreamSynchronize(s));
    ASV_HIP_CHECK(hipFree(m.ptr));
    m.ptr = nullptr; m.cap = 0;
  }
  const size_t cap = bytes + bytes / 8 + 4096;
  ASV_HIP_CHECK(hipMalloc(&m.ptr, cap));
  m.cap = cap;
  if (zero) ASV_HIP_CHECK(hipMemsetAsync(m.ptr, 0, cap, s));
  return ASV_OK;
}

struct Prof {
  asv_net *net; hipStream_t s; bool active = false;
  // mode 4: a run of back-to-back frame-level GEMM launches shares one event pair (every recorded event is a barrier
  // packet between two kernels; ten of them per step slowed the sampled steps by up to 2x)
  int close_span() {
    if (net->profiling == 4 && !net->stamps.empty() && net->stamps.back().open) {
      net->stamps.back().open = false;
      ASV_HIP_CHECK(hipEventRecord(net->stamps.back().b, s));
    }
    return ASV_OK;
  }
  int begin(int kclass, double flops, int op = -1) {
    if (net->profiling == 4) {
      active = false;
      if (kclass != K_TDNN) return close_span();
      if (!net->stamps.empty() && net->stamps.back().open) {
        net->stamps.back().flops += flops;
        net->stamps.back().launches += 1;
        return ASV_OK;
      }
      asv_net::Stamp st; st.kclass = kclass; st.flops = flops; st.op = -1; st.open = true;
      for (hipEvent_t *e : {&st.a, &st.b}) {
        if (!net->event_pool.empty()) { *e = net->event_pool.back(); net->event_pool.pop_back(); }
        else ASV_HIP_CHECK(hipEventCreate(e));
      }
      ASV_HIP_CHECK(hipEventRecord(st.a, s));
      net->stamps.push_back(st);
      return ASV_OK;
    }
    active = net->profiling != 0 && (net->profiling != 3 || kclass == K_TDNN);
    if (!active) return ASV_OK;
    asv_net::Stamp st; st.kclass = kclass; st.flops = flops; st.op = op;
    for (hipEvent_t *e : {&st.a, &st.b}) {
      if (!net->event_pool.empty()) { *e = net->event_pool.back(); net->event_pool.pop_back(); }
      else ASV_HIP_CHECK(hipEventCreate(e));
    }
    ASV_HIP_CHECK(hipEventRecord(st.a, s));
    net->stamps.push_back(st);
    return ASV_OK;
  }
  int end() {
    if (!active) return ASV_OK;
    ASV_HIP_CHECK(hipEventRecord(net->stamps.back().b, s));
    return ASV_OK;
  }
};

// Per-call plan of segments and rows.
struct DomainPlan {
  int rows = 0, rows_pad = 0;
  std::vector<int32_t> seg_row0, seg_len;      // first row / number of rows of each segment in this domain
};
struct BatchPlan {
  int n_utts = 0, segments = 0, seg_pad = 0;
  long long frames = 0;
  std::vector<int32_t> seg_src0, seg_frames, utt_seg0, utt_nseg;
  std::vector<DomainPlan> dom;                 // indexed like asv_net::domains
};

int make_plan(const asv_net *net, const int32_t *offsets, int n_utts, int max_chunk, BatchPlan &bp) {
  ASV_REQUIRE(offsets != nullptr && n_utts >= 1, "extract: need at least one utterance");
  ASV_REQUIRE(offsets[0] == 0, "extract: offsets[0] must be 0");
  if (max_chunk <= 0) max_chunk = 10000;
  bp.n_utts = n_utts;
  bp.utt_seg0.resize(n_utts); bp.utt_nseg.resize(n_utts);
  for (int u = 0; u < n_utts; ++u) {
    const long long T = (long long)offsets[u + 1] - offsets[u];
    ASV_REQUIRE(T >= 1, "extract: utterance %d has %lld frames (the reference asserts T >= tot_context, components.py:119)", u, T);
    // framework.py:34-47
    const int num_split = (int)((T + max_chunk - 1) / max_chunk);
    const int split = (int)(T / num_split);
    bp.utt_seg0[u] = (int32_t)bp.seg_frames.size();
    bp.utt_nseg[u] = num_split;
    for (int i = 0; i < num_split; ++i) {
      const int off = i * split;
      const int len = (i == num_split - 1) ? (int)(T - off) : split;
      bp.seg_src0.push_back(offsets[u] + off);
      bp.seg_frames.push_back(len);
    }
    bp.frames += T;
  }
  bp.segments = (int)bp.seg_frames.size();
  bp.seg_pad = round_up(bp.segments, kRowTile);
  bp.dom.resize(net->domains.size());
  for (size_t di = 0; di < net->domains.size(); ++di) {
    const Domain &dm = net->domains[di];
    DomainPlan &dp = bp.dom[di];
    if (dm.kind == ASV_DOMAIN_UTTS) {
      dp.rows = bp.segments; dp.rows_pad = bp.seg_pad;
      dp.seg_row0 = {0}; dp.seg_len = {bp.segments};          // one pseudo segment covering the valid rows
      continue;
    }
    long long row = dm.gap();
    dp.seg_row0.reserve(bp.segments); dp.seg_len.reserve(bp.segments);
    for (int sidx = 0; sidx < bp.segments; ++sidx) {
      long long fr = bp.seg_frames[sidx];
      for (int k = 0; k < dm.shift; ++k) fr = (fr + 1) / 2;       // stride-2 stages: ceil(T/2) each
      const long long len = fr * dm.pitch;
      dp.seg_row0.push_back((int32_t)row);
      dp.seg_len.push_back((int32_t)len);
      row += len + dm.gap();
      ASV_REQUIRE(row < (1ll << 30), "extract: batch too large (%lld rows in domain %zu)", row, di);
    }
    dp.rows = (int)row;
    dp.rows_pad = round_up(dp.rows, kRowTile);
  }
  return ASV_OK;
}

}  // namespace

extern "C" {

int asv_version(void) { return ASV_AMD_VERSION; }
const char *asv_last_error(void) { return g_last_error.c_str(); }

int asv_device_count(int *count) {
  ASV_REQUIRE(count != nullptr, "asv_device_count: null argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
  *count = n;
  return ASV_OK;
}

int asv_net_create(asv_net_t **out, int device, int precision, unsigned flags, int feat_dim) {
  ASV_REQUIRE(out != nullptr, "asv_net_create: null out pointer");
  ASV_REQUIRE(precision == ASV_PREC_F32 || precision == ASV_PREC_BF16 || precision == ASV_PREC_F32X || precision == ASV_PREC_F16, "asv_net_create: unknown precision %d", precision);
  ASV_REQUIRE((flags & ASV_FLAG_X3_SPLIT_BF16) == 0 || (flags & ASV_FLAG_X3_SPLIT_F16) == 0, "asv_net_create: both split types requested");
  ASV_REQUIRE(feat_dim >= 1, "asv_net_create: feat_dim %d", feat_dim);
  int ndev = 0;
  ASV_HIP_CHECK(hipGetDeviceCount(&ndev));
  ASV_REQUIRE(device >= 0 && device < ndev, "asv_net_create: device %d of %d", device, ndev);
  ASV_ON_DEVICE(device);
  asv_net *net = new (std::nothrow) asv_net();
  if (!net) { set_error("out of host memory"); return ASV_ENOMEM; }
  net->device = device; net->precision = precision; net->flags = flags; net->feat_dim = feat_dim;
  net->domains.push_back(Domain{ASV_DOMAIN_FRAMES});
  net->domains.push_back(Domain{ASV_DOMAIN_UTTS});
  net->bufs.push_back({ASV_DOMAIN_FRAMES, feat_dim, round_up(feat_dim, kChanAlign)});
  hipError_t ze = hipMalloc(&net->zero_page, 256);
  if (ze == hipSuccess) ze = hipMemset(net->zero_page, 0, 256);
  if (ze != hipSuccess) { set_error("zero page allocation failed: %s", hipGetErrorString(ze)); delete net; return ASV_EHIP; }
  *out = net;
  return ASV_OK;
}

void asv_net_destroy(asv_net_t *net) {
  if (!net) return;
  DeviceGuard guard;
  (void)guard.enter(net->device);
  (void)hipDeviceSynchronize();
  for (void *p : net->weight_allocs) (void)hipFree(p);
  for (auto &m : net->arena) if (m.ptr) (void)hipFree(m.ptr);
  if (net->meta_dev.ptr) (void)hipFree(net->meta_dev.ptr);
  if (net->rowmeta_dev.ptr) (void)hipFree(net->rowmeta_dev.ptr);
  if (net->splitk_dev.ptr) (void)hipFree(net->splitk_dev.ptr);
  if (net->poolpart_dev.ptr) (void)hipFree(net->poolpart_dev.ptr);
  if (net->lde_dev.ptr) (void)hipFree(net->lde_dev.ptr);
  if (net->zero_page) (void)hipFree(net->zero_page);
  if (net->plan_cache && net->plan_cache_free) net->plan_cache_free(net->plan_cache);
  if (net->meta_host) (void)hipHostFree(net->meta_host);
  if (net->meta_copied) (void)hipEventDestroy(net->meta_copied);
  for (auto &st : net->stamps) { (void)hipEventDestroy(st.a); (void)hipEventDestroy(st.b); }
  for (auto e : net->event_pool) (void)hipEventDestroy(e);
  delete net;
}

int asv_net_define_grid(asv_net_t *net, int time_shift, int width, int pitch) {
  ASV_REQUIRE(net && !net->finalized, "asv_net_define_grid: net is null or finalized");
  ASV_REQUIRE(time_shift >= 0 && time_shift <= 8, "asv_net_define_grid: time_shift %d", time_shift);
  // (width 1, pitch 1: a sequence at the grid's frame rate - no frequency neighbours, so no zero column between frames; asv_net_add_grid_flatten)
  ASV_REQUIRE(width >= 1 && (pitch >= width + 1 || (width == 1 && pitch == 1)) && pitch + 1 <= 84, "asv_net_define_grid: need 1 <= width < pitch <= 83, or width = pitch = 1 (got %d, %d)", width, pitch);
  Domain d{2};
  d.shift = time_shift; d.width = width; d.pitch = pitch;
  net->domains.push_back(d);
  return (int)net->domains.size() - 1;
}

int asv_net_new_buffer(asv_net_t *net, int domain, int channels) {
  ASV_REQUIRE(net && !net->finalized, "asv_net_new_buffer: net is null or finalized");
  ASV_REQUIRE(domain >= 0 && domain < (int)net->domains.size(), "asv_net_new_buffer: domain %d", domain);
  ASV_REQUIRE(channels >= 1 && channels <= (1 << 20), "asv_net_new_buffer: channels %d", channels);
  net->bufs.push_back({domain, channels, round_up(channels, kChanAlign)});
  return (int)net->bufs.size() - 1;
}

int asv_net_add_tdnn(asv_net_t *net, const asv_tdnn_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_tdnn: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_tdnn_desc_t), "asv_net_add_tdnn: struct_size %u != %zu (ABI mismatch)", d->struct_size, sizeof(asv_tdnn_desc_t));
  int rc;
  if ((rc = check_view(net, d->in_buf, d->in_ch_off, d->in_ch, "tdnn input"))) return rc;
  if ((rc = check_view(net, d->out_buf, d->out_ch_off, d->out_ch, "tdnn output"))) return rc;
  const int dom = net->bufs[d->in_buf].domain;
  ASV_REQUIRE(net->bufs[d->out_buf].domain == dom, "tdnn: input and output domains differ");
  // a layer may read and write the same buffer only through disjoint channel slices (tiles of
  // other workgroups read the halo rows of the input while this one stores its output rows)
  auto disjoint = [&](int buf, int off, int ch) {
    return buf != d->out_buf || off + round_up(ch, kChanAlign) <= d->out_ch_off || d->out_ch_off + round_up(d->out_ch, kChanAlign) <= off;
  };
  ASV_REQUIRE(d->out_buf != 0, "tdnn: the input feature buffer cannot be an output");
  ASV_REQUIRE(disjoint(d->in_buf, d->in_ch_off, d->in_ch), "tdnn: output slice overlaps the input slice of the same buffer");
  if (d->in2_buf >= 0) {
    if ((rc = check_view(net, d->in2_buf, d->in2_ch_off, d->in_ch, "tdnn second input"))) return rc;
    ASV_REQUIRE(net->bufs[d->in2_buf].domain == dom, "tdnn: second input domain differs");
    ASV_REQUIRE(disjoint(d->in2_buf, d->in2_ch_off, d->in_ch), "tdnn: output slice overlaps the second input slice");
  }
  if (d->res_buf >= 0) {
    if ((rc = check_view(net, d->res_buf, d->res_ch_off, d->out_ch, "tdnn residual"))) return rc;
    ASV_REQUIRE(net->bufs[d->res_buf].domain == dom, "tdnn: residual domain differs");
    ASV_REQUIRE(disjoint(d->res_buf, d->res_ch_off, d->out_ch), "tdnn: output slice overlaps the residual slice");
  }
  for (int sb : {d->seg_bias_buf, d->seg_scale_buf})
    if (sb >= 0) {
      ASV_REQUIRE(!net->is_utts(dom), "tdnn: per-segment bias/scale only applies to frame-level layers");
      ASV_REQUIRE(sb < (int)net->bufs.size() && net->is_utts(net->bufs[sb].domain) && net->bufs[sb].channels >= d->out_ch,
                  "tdnn: per-segment buffer %d must be an utts-domain buffer with >= %d channels", sb, d->out_ch);
    }
  ASV_REQUIRE(d->n_taps >= 1 && d->n_taps <= ASV_MAX_TAPS, "tdnn: n_taps %d", d->n_taps);
  for (int t = 0; t < d->n_taps; ++t) {
    ASV_REQUIRE(t == 0 || d->taps[t] > d->taps[t - 1], "tdnn: context must be strictly ascending (components.py:33-35)");
    ASV_REQUIRE(std::abs(d->taps[t]) <= net->domains[dom].halo() || (net->is_utts(dom) && d->taps[t] == 0),
                "tdnn: tap offset %d beyond the +-%d rows this domain keeps zero-padded", d->taps[t], net->domains[dom].halo());
    const int k = d->taps[t] - d->w_left_context;
    ASV_REQUIRE(k >= 0 && k < d->w_tot_context, "tdnn: tap %d outside the dense kernel [%d, %d)", d->taps[t], d->w_left_context, d->w_left_context + d->w_tot_context);
  }
  ASV_REQUIRE(!net->is_utts(dom) || (d->n_taps == 1 && d->taps[0] == 0), "tdnn: utts-domain layers have context [0] only");
  ASV_REQUIRE(d->weight != nullptr, "tdnn: null weight");
  ASV_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "tdnn: scale and shift come together");
  for (int a : {d->act1, d->act2}) ASV_REQUIRE(a >= ASV_ACT_NONE && a <= ASV_ACT_SIGMOID, "tdnn: unknown activation %d", a);

  Op op;
  op.kind = OP_TDNN;
  op.tdnn = *d;
  op.utts = net->is_utts(dom);
  const int et = net->dom_et(dom);
  const bool bf16 = et != ET_F32;                  // 16-bit rows and weights (bf16 or half)
  op.cin_pad = round_up(d->in_ch, kChanAlign);
  op.cout_pad = round_up(d->out_ch, kBigTileN);
  op.cout_store = round_up(d->out_ch, kChanAlign);
  ASV_REQUIRE(d->out_ch_off + op.cout_store <= net->bufs[d->out_buf].ld, "tdnn: padded output view exceeds the buffer pitch");
  ASV_ON_DEVICE(net->device);
  {
    const size_t n = (size_t)op.cout_pad * d->n_taps * op.cin_pad;
    std::vector<unsigned char> packed(n * (bf16 ? 2 : 4));
    pack_tdnn_weight(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cout_pad,
                     op.cin_pad, et, packed.data());
    if ((rc = dev_upload(net, packed.data(), packed.size(), &op.w))) return rc;
    // the two fragment orders are mutually exclusive per layer: a 3x3 trunk convolution the conv2d kernels take never needs
    // the v3 order (grid_conv_* precede big3 in the dispatch and accept every such layer), and each family has its own pointer
    // (+ the 32 -> 64 stride-2 convolution in space-to-depth form: 4 x 32 input channels, 4 taps, 64 output channels)
    const bool s2d_pack = bf16 && net->domains[dom].kind == 2 && d->n_taps == 4 && op.cin_pad == 128 && d->in_ch == 128 && d->out_ch == 64;
    const bool conv2d_pack = s2d_pack || (bf16 && net->domains[dom].kind == 2 && d->n_taps == 9 && (op.cin_pad == 32 || op.cin_pad == 64 || op.cin_pad == 128 || op.cin_pad == 256) &&
                             d->in_ch == op.cin_pad && d->out_ch == d->in_ch);
    if (bf16 && !conv2d_pack && op.cout_store > 96 && op.cin_pad >= 64) {         // candidates of the 256- / 128-channel tiles (kernels_tdnn_v3.hip)
      std::vector<uint16_t> frags(tdnn_weight_frag_elems(op.cout_pad, op.cin_pad, d->n_taps));
      pack_tdnn_weight_frags(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cout_pad, op.cin_pad, frags.data(), nullptr, et);
      if ((rc = dev_upload(net, frags.data(), frags.size() * 2, &op.wfrag))) return rc;
    }
    if (net->x3() && !op.utts && net->domains[dom].kind == ASV_DOMAIN_FRAMES && op.cout_store >= 192 && op.cin_pad >= 32) {
      // f32x mode: hi / lo halves of the weights in the same fragment order (kernels_tdnn_x3.hip); the half-precision split
      // scales the layer's weights by a power of two first (see x3_weight_scale)
      std::vector<uint16_t> hi(tdnn_weight_frag_elems(op.cout_pad, op.cin_pad, d->n_taps)), lo(hi.size());
      op.w_scale = net->x3_et() == ET_F16 ? x3_weight_scale(d->weight, (size_t)d->out_ch * d->in_ch * d->w_tot_context) : 1.0f;
      pack_tdnn_weight_frags(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cout_pad, op.cin_pad, hi.data(), lo.data(),
                             net->x3_et(), op.w_scale);
      if ((rc = dev_upload(net, hi.data(), hi.size() * 2, &op.wfrag))) return rc;
      if ((rc = dev_upload(net, lo.data(), lo.size() * 2, &op.wlo))) return rc;
      if (net->x3_mx() && op.cout_pad % 32 == 0) {
        std::vector<uint8_t> w8(tdnn_weight_mx8_bytes(op.cout_pad, op.cin_pad, d->n_taps));
        pack_tdnn_weight_mx8(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cout_pad, op.cin_pad, op.w_scale, w8.data());
        if ((rc = dev_upload(net, w8.data(), w8.size(), &op.w8))) return rc;
      }
      // the candidates of the 8-phase form (256 x 256 tiles; the dispatch decides per batch): the plain epilogue, whole 32-channel chunks
      const bool plain = (d->act1 == ASV_ACT_NONE || d->act1 == ASV_ACT_RELU) && d->act2 == ASV_ACT_NONE && !d->affine_first;
      if (plain && op.cin_pad % 32 == 0 && d->in2_buf < 0 && d->seg_bias_buf < 0 && d->seg_scale_buf < 0 && d->res_buf < 0 && (long long)op.cin_pad * d->n_taps >= 512) {
        std::vector<uint16_t> rows((size_t)op.cout_pad * d->n_taps * (op.cin_pad / 32) * 64);
        pack_tdnn_weight_x3p(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cout_pad, op.cin_pad, net->x3_et(), op.w_scale,
                             rows.data());
        if ((rc = dev_upload(net, rows.data(), rows.size() * 2, &op.wx3p))) return rc;
      }
    }
    const bool x3_wide_frames = net->x3() && !op.utts && net->domains[dom].kind == ASV_DOMAIN_FRAMES && op.cout_store >= 192 && op.cin_pad >= 32 && d->in2_buf < 0;
    if (net->x3() && !op.utts && (net->domains[dom].kind == 2 || !x3_wide_frames) && grid_conv_x3_shape_ok(op.cin_pad, op.cout_store) &&
        (net->flags & ASV_FLAG_SMALL_TILES) == 0) {
      // f32x mode, grid domain (the 2-D ResNet trunk) and the frames-domain layers the wide split kernel does not take (fewer than
      // 192 output channels, or a second input): hi / lo halves in the fragment order of kernels_conv2d_x3.hip, scaled by the
      // layer's power of two for the half-precision split
      std::vector<uint16_t> frags(grid_conv_x3_frag_elems(op.cin_pad, op.cout_store, d->n_taps));
      op.w_scale = net->x3_et() == ET_F16 ? x3_weight_scale(d->weight, (size_t)d->out_ch * d->in_ch * d->w_tot_context) : 1.0f;
      pack_grid_conv_x3_frags(d->weight, d->out_ch, d->in_ch, d->w_tot_context, d->w_left_context, d->taps, d->n_taps, op.cin_pad, op.cout_store, net->x3_et(),
                              op.w_scale, frags.data());
      if ((rc = dev_upload(net, frags.data(), frags.size() * 2, &op.wconv))) return rc;
    }
    if (conv2d_pack) {
      // 3x3 trunk convolutions with 32 / 64 channels: fragment order of kernels_conv2d.hip,
      // [tap][k-group][n-fragment][lane = (k half lh, channel lr)][8], k = kg * 16 + lh * 8 + e
      const int kgs = op.cin_pad / 16, nfs = (d->out_ch + 31) / 32;
      std::vector<uint16_t> frags(grid_conv_frag_elems(op.cin_pad, nfs * 32, d->n_taps), 0);
      for (int t = 0; t < d->n_taps; ++t) {
        const int k = d->taps[t] - d->w_left_context;
        for (int co = 0; co < d->out_ch; ++co)
          for (int ci = 0; ci < d->in_ch; ++ci) {
            const int kg = ci / 16, lh = (ci % 16) / 8, e = ci % 8, nf = co / 32, lr = co % 32;
            frags[((size_t)(t * kgs + kg) * nfs + nf) * 512 + (size_t)(lh * 32 + lr) * 8 + e] =
                f32_to_h16_host(d->weight[((size_t)co * d->in_ch + ci) * d->w_tot_context + k], et);
          }
      }
      if ((rc = dev_upload(net, frags.data(), frags.size() * 2, &op.wconv))) return rc;
    }
    if (op.utts && net->frames_h16()) {
      // pooled-domain layers keep f32 activations; their GEMM runs on the bf16 matrix cores with every
      // operand split into two bf16 halves, the weight halves in the fragment order kernels_utts.hip walks:
      // [32-channel fragment][32-k step][j][lane = (k half lh, channel lr)][8], k = 32 * step + 16 * lh + 8 * j + e
      const float *wf = reinterpret_cast<const float *>(packed.data());
      const int ksteps = (op.cin_pad + 31) / 32;
      const size_t nfrag = (size_t)(op.cout_pad / 32) * ksteps * 1024;
      std::vector<uint16_t> hi(nfrag, 0), lo(nfrag, 0);
      for (int co = 0; co < d->out_ch; ++co)
        for (int ci = 0; ci < d->in_ch; ++ci) {
          const float v = wf[(size_t)co * op.cin_pad + ci];
          const int r = ci % 32;
          const size_t idx = (((size_t)(co / 32) * ksteps + ci / 32) * 2 + (r % 16) / 8) * 512 + (size_t)((r / 16) * 32 + co % 32) * 8 + r % 8;
          hi[idx] = f32_to_bf16_host(v);
          lo[idx] = f32_to_bf16_host(v - bf16_to_f32_host(hi[idx]));
        }
      if ((rc = dev_upload(net, hi.data(), nfrag * 2, &op.wfrag))) return rc;
      if ((rc = dev_upload(net, lo.data(), nfrag * 2, &op.wlo))) return rc;
    }
  }
  if ((rc = upload_padded(net, d->bias, d->out_ch, op.cout_pad, 0.0f, &op.bias))) return rc;
  op.has_affine = d->scale != nullptr;
  if (op.has_affine) {
    if ((rc = upload_padded(net, d->scale, d->out_ch, op.cout_pad, 0.0f, &op.scale))) return rc;
    if ((rc = upload_padded(net, d->shift, d->out_ch, op.cout_pad, 0.0f, &op.shift))) return rc;
  }
  if ((bf16 || net->x3()) && !op.utts && net->domains[dom].kind == ASV_DOMAIN_FRAMES && op.wfrag != nullptr && (net->flags & (ASV_FLAG_NO_FUSE | ASV_FLAG_NO_CHAIN)) == 0) {
    // possible member of a layer chain: its BatchNorm may be folded into the next member, or the previous member's into it
    if (d->scale) { op.host_scale.assign(d->scale, d->scale + d->out_ch); op.host_shift.assign(d->shift, d->shift + d->out_ch); }
    if (d->n_taps == 1 && d->taps[0] == 0 && d->in_ch == kChainWidth) {
      const size_t tot = (size_t)d->w_tot_context, k = (size_t)(0 - d->w_left_context);
      op.host_w.resize((size_t)d->out_ch * d->in_ch);
      for (size_t i = 0; i < op.host_w.size(); ++i) op.host_w[i] = d->weight[i * tot + k];           // the one active tap, [out][in]
      op.host_bias.assign((size_t)d->out_ch, 0.0f);
      if (d->bias) op.host_bias.assign(d->bias, d->bias + d->out_ch);
    }
  }
  op.tdnn.weight = nullptr; op.tdnn.bias = nullptr; op.tdnn.scale = nullptr; op.tdnn.shift = nullptr;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_stats_pool(asv_net_t *net, const asv_pool_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_stats_pool: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_pool_desc_t), "asv_net_add_stats_pool: struct_size mismatch");
  int rc;
  if ((rc = check_view(net, d->in_buf, d->in_ch_off, d->channels, "pool input"))) return rc;
  int out_ch = d->channels * (d->stddev ? 2 : 1);
  ASV_REQUIRE(d->in_buf >= 0 && d->in_buf < (int)net->bufs.size(), "pool: input buffer id %d", d->in_buf);
  if (d->per_bin) out_ch *= net->domains[net->bufs[d->in_buf].domain].width;
  ASV_REQUIRE(d->out_buf > 0 && d->out_buf < (int)net->bufs.size(), "pool: output buffer id %d", d->out_buf);
  ASV_REQUIRE(d->out_ch_off >= 0 && d->out_ch_off + out_ch <= net->bufs[d->out_buf].channels, "pool: output view exceeds buffer");
  ASV_REQUIRE(!net->is_utts(net->bufs[d->in_buf].domain) && net->is_utts(net->bufs[d->out_buf].domain),
              "pool: goes from a frame-level domain to the utts domain");
  ASV_REQUIRE(d->per_bin == 0 || net->domains[net->bufs[d->in_buf].domain].kind == 2, "pool: per_bin needs a grid-domain input");
  ASV_REQUIRE(d->unbiased >= 0 && d->unbiased <= 2 && (d->var_mode == ASV_POOL_VAR_CLAMP || d->var_mode == ASV_POOL_VAR_ADD), "pool: bad mode");
  Op op; op.kind = OP_POOL; op.pool = *d;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_attentive_pool(asv_net_t *net, const asv_attpool_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_attentive_pool: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_attpool_desc_t), "asv_net_add_attentive_pool: struct_size mismatch");
  int rc;
  if ((rc = check_view(net, d->x_buf, d->x_ch_off, d->channels, "attentive pool x"))) return rc;
  ASV_REQUIRE(d->logit_group >= 0 && d->logit_group <= d->channels, "attentive pool: logit_group %d", d->logit_group);
  const int group = d->logit_group > 1 ? d->logit_group : (d->shared_logits ? d->channels : 1);
  if (group > 1) {                 // head logits are read one element at a time: any column offset (head h of the global poolings)
    ASV_REQUIRE(d->logit_buf >= 0 && d->logit_buf < (int)net->bufs.size() && d->logit_ch_off >= 0 &&
                d->logit_ch_off + (d->channels + group - 1) / group <= net->bufs[d->logit_buf].channels,
                "attentive pool logits: columns [%d,%d) exceed buffer %d", d->logit_ch_off, d->logit_ch_off + (d->channels + group - 1) / group, d->logit_buf);
  } else if ((rc = check_view(net, d->logit_buf, d->logit_ch_off, d->channels, "attentive pool logits"))) return rc;
  ASV_REQUIRE(d->out_buf > 0 && d->out_buf < (int)net->bufs.size(), "attentive pool: output buffer id %d", d->out_buf);
  ASV_REQUIRE(d->out_ch_off >= 0 && d->out_ch_off + 2 * d->channels <= net->bufs[d->out_buf].channels, "attentive pool: output view exceeds buffer");
  ASV_REQUIRE(net->is_sequence(net->bufs[d->x_buf].domain) && net->bufs[d->logit_buf].domain == net->bufs[d->x_buf].domain &&
              net->is_utts(net->bufs[d->out_buf].domain), "attentive pool: frames (or a sequence domain) -> utts");
  ASV_REQUIRE((d->prior_logit == nullptr) == (d->prior_value == nullptr), "attentive pool: prior logits and values come together");
  ASV_REQUIRE(!(d->logit_softplus2 || d->prior_logit) || group == 1, "attentive pool: the xi-vector options need per-channel logits");
  Op op; op.kind = OP_ATTPOOL; op.att = *d;
  if (d->prior_logit) {                                         // parked in the scale / shift slots of the op
    ASV_ON_DEVICE(net->device);
    if ((rc = upload_padded(net, d->prior_logit, d->channels, round_up(d->channels, kChanAlign), 0.0f, &op.scale))) return rc;
    if ((rc = upload_padded(net, d->prior_value, d->channels, round_up(d->channels, kChanAlign), 0.0f, &op.shift))) return rc;
  }
  op.att.prior_logit = nullptr; op.att.prior_value = nullptr;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_lde_pool(asv_net_t *net, const asv_lde_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_lde_pool: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_lde_desc_t), "asv_net_add_lde_pool: struct_size mismatch");
  ASV_REQUIRE(d->mu && d->beta && d->n_centres >= 1 && d->n_centres <= 64, "lde pool: mu / beta are required and 1 <= n_centres <= 64 (got %d)", d->n_centres);
  int rc;
  if ((rc = check_view(net, d->x_buf, d->x_ch_off, d->channels, "lde pool x"))) return rc;
  ASV_REQUIRE(d->out_buf > 0 && d->out_buf < (int)net->bufs.size(), "lde pool: output buffer id %d", d->out_buf);
  ASV_REQUIRE(d->out_ch_off >= 0 && d->out_ch_off + d->channels * d->n_centres <= net->bufs[d->out_buf].channels, "lde pool: output view exceeds buffer");
  ASV_REQUIRE(net->is_sequence(net->bufs[d->x_buf].domain) && net->is_utts(net->bufs[d->out_buf].domain), "lde pool: frames (or a sequence domain) -> utts");
  Op op; op.kind = OP_LDE; op.lde = *d;
  ASV_ON_DEVICE(net->device);
  const int n = d->channels * d->n_centres;
  if ((rc = upload_padded(net, d->mu, n, round_up(n, kChanAlign), 0.0f, &op.scale))) return rc;
  if ((rc = upload_padded(net, d->beta, d->n_centres, 64, 0.0f, &op.shift))) return rc;
  op.lde.mu = nullptr; op.lde.beta = nullptr;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_res2(asv_net_t *net, const asv_res2_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_res2: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_res2_desc_t), "asv_net_add_res2: struct_size mismatch");
  ASV_REQUIRE(net->frames_h16(), "res2: the one-kernel Res2NetBlock exists for the 16-bit precision modes (bf16, f16) only");
  ASV_REQUIRE(d->branches >= 1 && d->branches <= 7 && d->dilation >= 1 && d->dilation <= kHalo, "res2: %d branches, dilation %d", d->branches, d->dilation);
  ASV_REQUIRE(d->weight && d->bias && d->scale && d->shift, "res2: weight / bias / scale / shift are required");
  const int ch = (d->branches + 1) * kRes2Width;
  int rc;
  if ((rc = check_view(net, d->in_buf, d->in_ch_off, ch, "res2 input"))) return rc;
  if ((rc = check_view(net, d->out_buf, d->out_ch_off, ch, "res2 output"))) return rc;
  ASV_REQUIRE(net->bufs[d->in_buf].domain == ASV_DOMAIN_FRAMES && net->bufs[d->out_buf].domain == ASV_DOMAIN_FRAMES && d->out_buf != d->in_buf && d->out_buf != 0,
              "res2: frames-domain input and a different frames-domain output buffer");
  ASV_REQUIRE(d->in_ch_off % 8 == 0 && d->out_ch_off % 8 == 0, "res2: channel offsets must be multiples of 8 (16-byte rows pieces)");
  Op op; op.kind = OP_RES2; op.res2 = *d;
  ASV_ON_DEVICE(net->device);
  const int W = kRes2Width, tot = 2 * d->dilation + 1;
  const int taps[3] = {-d->dilation, 0, d->dilation};
  const size_t per_branch = tdnn_weight_frag_elems(W, W, 3);
  std::vector<uint16_t> frags(per_branch * d->branches);
  for (int b = 0; b < d->branches; ++b)
    pack_tdnn_weight_frags(d->weight + (size_t)b * W * W * tot, W, W, tot, -d->dilation, taps, 3, W, W, frags.data() + per_branch * b, nullptr, net->frames_et());
  if ((rc = dev_upload(net, frags.data(), frags.size() * 2, &op.wfrag))) return rc;
  if ((rc = upload_padded(net, d->bias, d->branches * W, d->branches * W, 0.0f, &op.bias))) return rc;
  if ((rc = upload_padded(net, d->scale, d->branches * W, d->branches * W, 0.0f, &op.scale))) return rc;
  if ((rc = upload_padded(net, d->shift, d->branches * W, d->branches * W, 0.0f, &op.shift))) return rc;
  op.res2.weight = nullptr; op.res2.bias = nullptr; op.res2.scale = nullptr; op.res2.shift = nullptr;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_eltwise(asv_net_t *net, const asv_eltwise_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_eltwise: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_eltwise_desc_t), "asv_net_add_eltwise: struct_size mismatch");
  int rc;
  if ((rc = check_view(net, d->a_buf, d->a_ch_off, d->channels, "eltwise a"))) return rc;
  if ((rc = check_view(net, d->out_buf, d->out_ch_off, d->channels, "eltwise out"))) return rc;
  const int dom = net->bufs[d->a_buf].domain;
  ASV_REQUIRE(net->bufs[d->out_buf].domain == dom && d->out_buf != 0, "eltwise: bad output buffer");
  if (d->b_buf >= 0) { if ((rc = check_view(net, d->b_buf, d->b_ch_off, d->channels, "eltwise b"))) return rc; ASV_REQUIRE(net->bufs[d->b_buf].domain == dom, "eltwise: b domain"); }
  if (d->c_buf >= 0) { if ((rc = check_view(net, d->c_buf, d->c_ch_off, d->channels, "eltwise c"))) return rc; ASV_REQUIRE(net->bufs[d->c_buf].domain == dom, "eltwise: c domain"); }
  if (d->seg_scale_buf >= 0)
    ASV_REQUIRE(!net->is_utts(dom) && d->seg_scale_buf < (int)net->bufs.size() && net->is_utts(net->bufs[d->seg_scale_buf].domain) &&
                net->bufs[d->seg_scale_buf].channels >= d->channels, "eltwise: bad per-segment scale buffer");
  ASV_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "eltwise: scale and shift come together");
  if (d->seg_norm_buf >= 0)
    ASV_REQUIRE(!net->is_utts(dom) && d->seg_norm_buf < (int)net->bufs.size() && net->is_utts(net->bufs[d->seg_norm_buf].domain) &&
                net->bufs[d->seg_norm_buf].channels >= 2 * d->channels && d->seg_norm_mode >= 1 && d->seg_norm_mode <= 3,
                "eltwise: bad per-segment normalisation buffer / mode");
  ASV_REQUIRE(d->act >= ASV_ACT_NONE && d->act <= ASV_ACT_SIGMOID, "eltwise: unknown activation %d", d->act);
  const int vec = net->dom_et(dom) != ET_F32 ? 8 : 4;
  ASV_REQUIRE(d->out_ch_off + round_up(d->channels, vec) <= net->bufs[d->out_buf].ld, "eltwise: padded view exceeds pitch");
  ASV_REQUIRE((d->out2_buf >= 0) == (d->d_buf >= 0), "eltwise: the second output and its addend come together");
  if (d->out2_buf >= 0) {
    if ((rc = check_view(net, d->d_buf, d->d_ch_off, d->channels, "eltwise d"))) return rc;
    if ((rc = check_view(net, d->out2_buf, d->out2_ch_off, d->channels, "eltwise out2"))) return rc;
    ASV_REQUIRE(net->bufs[d->d_buf].domain == dom && net->bufs[d->out2_buf].domain == dom && d->out2_buf != 0 && d->out2_buf != d->out_buf,
                "eltwise: bad second output / addend buffer");
    ASV_REQUIRE(d->out2_ch_off + round_up(d->channels, vec) <= net->bufs[d->out2_buf].ld, "eltwise: padded second view exceeds pitch");
  }
  Op op; op.kind = OP_ELTWISE; op.elt = *d; op.utts = net->is_utts(dom);
  ASV_ON_DEVICE(net->device);
  if (d->scale) {
    const int n_pad = round_up(d->channels, kChanAlign);
    if ((rc = upload_padded(net, d->scale, d->channels, n_pad, 0.0f, &op.scale))) return rc;
    if ((rc = upload_padded(net, d->shift, d->channels, n_pad, 0.0f, &op.shift))) return rc;
  }
  op.elt.scale = nullptr; op.elt.shift = nullptr;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_grid_input(asv_net_t *net, const asv_grid_input_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_grid_input: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_grid_input_desc_t), "asv_net_add_grid_input: struct_size mismatch");
  ASV_REQUIRE(d->out_buf > 0 && d->out_buf < (int)net->bufs.size(), "grid_input: output buffer id %d", d->out_buf);
  ASV_REQUIRE(d->in_buf >= 0 && d->in_buf < (int)net->bufs.size() && net->bufs[d->in_buf].domain == ASV_DOMAIN_FRAMES && net->bufs[d->in_buf].channels == net->feat_dim,
              "grid_input: input must be a frames-domain buffer with feat_dim channels");
  const Domain &dm = net->domains[net->bufs[d->out_buf].domain];
  ASV_REQUIRE(dm.kind == 2 && dm.shift == 0 && dm.width == net->feat_dim, "grid_input: output must be a full-resolution grid of width feat_dim (%d)", net->feat_dim);
  Op op; op.kind = OP_GRID_INPUT; op.gin = *d;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_im2col(asv_net_t *net, const asv_im2col_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_im2col: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_im2col_desc_t), "asv_net_add_im2col: struct_size mismatch");
  int rc;
  if ((rc = check_view(net, d->in_buf, 0, d->channels, "im2col input"))) return rc;
  ASV_REQUIRE(d->n_taps >= 1 && d->n_taps <= ASV_MAX_TAPS && (d->stride == 1 || d->stride == 2), "im2col: n_taps %d stride %d", d->n_taps, d->stride);
  ASV_REQUIRE(d->channels % kChanAlign == 0 && d->channels == net->bufs[d->in_buf].channels, "im2col: channels must be the whole input buffer and a multiple of %d", kChanAlign);
  if ((rc = check_view(net, d->out_buf, 0, d->channels * d->n_taps, "im2col output"))) return rc;
  const Domain &di = net->domains[net->bufs[d->in_buf].domain], &dq = net->domains[net->bufs[d->out_buf].domain];
  ASV_REQUIRE(di.kind == 2 && dq.kind == 2 && d->out_buf != d->in_buf && d->out_buf != 0, "im2col: grid -> grid");
  ASV_REQUIRE(dq.shift == di.shift + (d->stride == 2 ? 1 : 0) && dq.width == (di.width + d->stride - 1) / d->stride,
              "im2col: output grid (T/%d x %d) does not match stride %d over input grid (T/%d x %d)", 1 << dq.shift, dq.width, d->stride, 1 << di.shift, di.width);
  ASV_REQUIRE(d->act == ASV_ACT_NONE || d->act == ASV_ACT_RELU, "im2col: the elementwise prologue takes no activation or ReLU (got %d)", d->act);
  // optional buffers: -1 = none, and so is 0 - buffer 0 is the feature matrix (frames domain), which can be neither a grid addend nor a
  // per-segment scale, so a caller that zero-initialises the descriptor (memset + struct_size, the usual C pattern) gets what the
  // descriptor meant before these fields existed
  asv_im2col_desc_t norm = *d;
  if (norm.b_buf <= 0) norm.b_buf = -1;
  if (norm.seg_scale_buf <= 0) norm.seg_scale_buf = -1;
  d = &norm;
  if (d->b_buf >= 0) {
    if ((rc = check_view(net, d->b_buf, 0, d->channels, "im2col addend"))) return rc;
    ASV_REQUIRE(net->bufs[d->b_buf].domain == net->bufs[d->in_buf].domain && net->bufs[d->b_buf].channels == d->channels && d->b_buf != d->out_buf,
                "im2col: the addend must be a whole buffer of the input's grid and width");
  }
  if (d->seg_scale_buf >= 0)
    ASV_REQUIRE(d->seg_scale_buf < (int)net->bufs.size() && net->is_utts(net->bufs[d->seg_scale_buf].domain) && net->bufs[d->seg_scale_buf].channels >= d->channels,
                "im2col: bad per-segment scale buffer");
  Op op; op.kind = OP_IM2COL; op.i2c = *d;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_add_grid_flatten(asv_net_t *net, const asv_grid_flatten_desc_t *d) {
  ASV_REQUIRE(net && d && !net->finalized, "asv_net_add_grid_flatten: net is null or finalized");
  ASV_REQUIRE(d->struct_size == sizeof(asv_grid_flatten_desc_t), "asv_net_add_grid_flatten: struct_size mismatch");
  ASV_REQUIRE(d->in_buf > 0 && d->in_buf < (int)net->bufs.size() && d->out_buf > 0 && d->out_buf < (int)net->bufs.size() && d->in_buf != d->out_buf,
              "grid_flatten: buffer ids %d -> %d", d->in_buf, d->out_buf);
  const Domain &di = net->domains[net->bufs[d->in_buf].domain], &dq = net->domains[net->bufs[d->out_buf].domain];
  ASV_REQUIRE(di.kind == 2 && dq.kind == 2 && dq.width == 1 && dq.pitch == 1 && dq.shift == di.shift,
              "grid_flatten: the output must live on the sequence domain (width 1, pitch 1) of the input grid's time shift");
  ASV_REQUIRE(net->bufs[d->out_buf].channels == net->bufs[d->in_buf].channels * di.width, "grid_flatten: %d channels x %d bins do not give %d output channels",
              net->bufs[d->in_buf].channels, di.width, net->bufs[d->out_buf].channels);
  Op op; op.kind = OP_FLATTEN; op.flat = *d;
  net->ops.push_back(op);
  return ASV_OK;
}

int asv_net_finalize(asv_net_t *net, int out_buf, int embed_dim) {
  ASV_REQUIRE(net && !net->finalized, "asv_net_finalize: net is null or already finalized");
  ASV_REQUIRE(out_buf > 0 && out_buf < (int)net->bufs.size() && net->is_utts(net->bufs[out_buf].domain),
              "asv_net_finalize: output must be an utts-domain buffer");
  ASV_REQUIRE(embed_dim >= 1 && embed_dim <= net->bufs[out_buf].channels, "asv_net_finalize: embed_dim %d", embed_dim);
  ASV_REQUIRE(!net->ops.empty(), "asv_net_finalize: empty program");
  // fuse "layer -> StatisticsPooling" when the layer's output has no other reader: the 157 MB tensor
  // (C2: 52k frames x 1500 channels) is then never written to or re-read from HBM
  if ((net->flags & ASV_FLAG_NO_FUSE) == 0 && (net->frames_h16() || net->x3())) {
    for (size_t i = 0; i + 1 < net->ops.size(); ++i) {
      Op &a = net->ops[i], &b = net->ops[i + 1];
      if (a.kind != OP_TDNN || b.kind != OP_POOL || a.utts || a.wfrag == nullptr) continue;
      const auto &d = a.tdnn; const auto &q = b.pool;
      if (net->bufs[d.in_buf].domain != ASV_DOMAIN_FRAMES || q.in_buf != d.out_buf || q.in_ch_off != d.out_ch_off || q.channels != d.out_ch || q.per_bin) continue;
      if (d.in2_buf >= 0 || d.seg_bias_buf >= 0 || d.seg_scale_buf >= 0 || d.res_buf >= 0 || d.affine_first || d.act2 != ASV_ACT_NONE ||
          (d.act1 != ASV_ACT_NONE && d.act1 != ASV_ACT_RELU)) continue;
      bool other_reader = false;
      for (size_t k = 0; k < net->ops.size(); ++k) {
        if (k == i || k == i + 1) continue;
        const Op &o = net->ops[k];
        const int reads[] = {o.kind == OP_TDNN ? o.tdnn.in_buf : -1, o.kind == OP_TDNN ? o.tdnn.in2_buf : -1, o.kind == OP_TDNN ? o.tdnn.res_buf : -1,
                             o.kind == OP_POOL ? o.pool.in_buf : -1, o.kind == OP_ATTPOOL ? o.att.x_buf : -1, o.kind == OP_ATTPOOL ? o.att.logit_buf : -1,
                             o.kind == OP_ELTWISE ? o.elt.a_buf : -1, o.kind == OP_ELTWISE ? o.elt.b_buf : -1, o.kind == OP_ELTWISE ? o.elt.c_buf : -1,
                             o.kind == OP_ELTWISE ? o.elt.d_buf : -1,
                             o.kind == OP_IM2COL ? o.i2c.in_buf : -1, o.kind == OP_IM2COL ? o.i2c.b_buf : -1, o.kind == OP_LDE ? o.lde.x_buf : -1, o.kind == OP_RES2 ? o.res2.in_buf : -1, o.kind == OP_FLATTEN ? o.flat.in_buf : -1};
        for (int rbuf : reads) other_reader |= (rbuf == d.out_buf);
      }
      if (other_reader || d.out_buf == out_buf) continue;
      a.fused_pool = (int)(i + 1);
      b.skipped = true;
    }
  }
  auto sole_reader = [&](int buf, size_t reader) {
    if (buf == out_buf) return false;
    for (size_t k = 0; k < net->ops.size(); ++k) {
      if (k == reader) continue;
      const Op &o = net->ops[k];
      const int reads[] = {o.kind == OP_TDNN ? o.tdnn.in_buf : -1, o.kind == OP_TDNN ? o.tdnn.in2_buf : -1, o.kind == OP_TDNN ? o.tdnn.res_buf : -1,
                           o.kind == OP_TDNN ? o.tdnn.seg_bias_buf : -1, o.kind == OP_TDNN ? o.tdnn.seg_scale_buf : -1,
                           o.kind == OP_POOL ? o.pool.in_buf : -1, o.kind == OP_ATTPOOL ? o.att.x_buf : -1, o.kind == OP_ATTPOOL ? o.att.logit_buf : -1,
                           o.kind == OP_ELTWISE ? o.elt.a_buf : -1, o.kind == OP_ELTWISE ? o.elt.b_buf : -1, o.kind == OP_ELTWISE ? o.elt.c_buf : -1,
                           o.kind == OP_ELTWISE ? o.elt.seg_scale_buf : -1, o.kind == OP_ELTWISE ? o.elt.seg_norm_buf : -1, o.kind == OP_ELTWISE ? o.elt.d_buf : -1,
                           o.kind == OP_IM2COL ? o.i2c.in_buf : -1, o.kind == OP_IM2COL ? o.i2c.b_buf : -1, o.kind == OP_IM2COL ? o.i2c.seg_scale_buf : -1, o.kind == OP_LDE ? o.lde.x_buf : -1, o.kind == OP_GRID_INPUT ? o.gin.in_buf : -1,
                           o.kind == OP_RES2 ? o.res2.in_buf : -1, o.kind == OP_FLATTEN ? o.flat.in_buf : -1};
      for (int rbuf : reads) if (rbuf == buf) return false;
    }
    return true;
  };
  // chains "layer -> 512, [1-tap 512 -> 512]*, 1-tap + fused pooling" whose intermediate tensors nobody else reads run as
  // ONE kernel with the 128 x 512 tiles resident in LDS (x-vector: tdnn3 -> tdnn4 -> tdnn5 -> pooling)
  if ((net->flags & (ASV_FLAG_NO_FUSE | ASV_FLAG_NO_CHAIN)) == 0 && (net->frames_h16() || (net->x3() && net->x3_terms() == 7))) {
    auto plain = [&](const Op &o) {
      const auto &d = o.tdnn;
      return o.kind == OP_TDNN && !o.utts && o.wfrag != nullptr && (!net->x3() || o.wlo != nullptr) && net->bufs[d.in_buf].domain == ASV_DOMAIN_FRAMES && d.in2_buf < 0 && d.seg_bias_buf < 0 &&
             d.seg_scale_buf < 0 && d.res_buf < 0 && !d.affine_first && d.act2 == ASV_ACT_NONE && (d.act1 == ASV_ACT_NONE || d.act1 == ASV_ACT_RELU);
    };
    auto from_resident = [&](const Op &o) {            // 1-tap layer that consumes a whole 512-channel buffer
      const auto &d = o.tdnn;
      return d.n_taps == 1 && d.taps[0] == 0 && d.in_ch == kChainWidth && d.in_ch_off == 0 && net->bufs[d.in_buf].channels == kChainWidth;
    };
    for (size_t l = 1; l < net->ops.size(); ++l) {
      Op &last = net->ops[l];
      if (!plain(last) || last.fused_pool < 0 || !from_resident(last)) continue;
      size_t head = l;
      while (head > 0 && l - head < 3) {
        const Op &prev = net->ops[head - 1], &cur = net->ops[head];
        if (!plain(prev) || prev.fused_pool >= 0 || prev.tdnn.out_buf != cur.tdnn.in_buf || prev.tdnn.out_ch_off != 0 || prev.tdnn.out_ch != kChainWidth ||
            prev.cout_pad != kChainWidth || !sole_reader(prev.tdnn.out_buf, head)) break;
        --head;
        if (!from_resident(prev)) break;               // any-tap layer: it can only be the chain's first
      }
      const Op &first = net->ops[head];
      if (head == l || first.cin_pad % 64 != 0 || first.cin_pad < 64) continue;
      bool mids_ok = true;
      for (size_t k = head + 1; k < l; ++k) mids_ok &= from_resident(net->ops[k]);
      if (!mids_ok) continue;
      net->ops[head].chain_last = (int)l;
      // Fold the eval BatchNorm y = s u + t of every chain layer but the last into its (1-tap) consumer:
      //   W (s u + t) + b = (W diag s) u + (W t + b)          - exact in real arithmetic; the products W s are rounded to the
      // 16-bit element type once, here, instead of s u + t being rounded per value in the kernel.  The producing layer's
      // epilogue inside the chain kernel is then ReLU + conversion alone (kernels_tdnn_chain.hip store_Y).  The per-layer
      // path (ASV_FLAG_NO_CHAIN, batches the chain cannot take) keeps the unfolded weights: both are uploaded.
      static const bool no_fold = getenv("ASV_AMD_CHAIN_FOLD") != nullptr && atoi(getenv("ASV_AMD_CHAIN_FOLD")) == 0;
      for (size_t k = head + 1; k <= l && !no_fold; ++k) {
        Op &cur = net->ops[k];
        const Op &prev = net->ops[k - 1];
        if (!prev.has_affine || prev.host_scale.empty() || cur.host_w.empty()) continue;
        const int out_ch = cur.tdnn.out_ch, in_ch = cur.tdnn.in_ch;
        std::vector<float> wf((size_t)out_ch * in_ch), bf((size_t)cur.cout_pad, 0.0f);
        for (int co = 0; co < out_ch; ++co) {
          double acc = cur.host_bias[co];
          for (int ci = 0; ci < in_ch; ++ci) {
            const float w = cur.host_w[(size_t)co * in_ch + ci];
            wf[(size_t)co * in_ch + ci] = w * prev.host_scale[ci];
            acc += (double)w * (double)prev.host_shift[ci];
          }
          bf[co] = (float)acc;
        }
        const int tap0 = 0;
        std::vector<uint16_t> frags(tdnn_weight_frag_elems(cur.cout_pad, cur.cin_pad, 1)), frags_lo;
        ASV_ON_DEVICE(net->device);
        int rc;
        if (net->x3()) {
          frags_lo.resize(frags.size());
          cur.w_scale_fold = net->x3_et() == ET_F16 ? x3_weight_scale(wf.data(), wf.size()) : 1.0f;
          pack_tdnn_weight_frags(wf.data(), out_ch, in_ch, 1, 0, &tap0, 1, cur.cout_pad, cur.cin_pad, frags.data(), frags_lo.data(), net->x3_et(), cur.w_scale_fold);
          if ((rc = dev_upload(net, frags_lo.data(), frags_lo.size() * 2, &cur.wlo_fold))) return rc;
          if (net->x3_mx()) {
            std::vector<uint8_t> w8(tdnn_weight_mx8_bytes(cur.cout_pad, cur.cin_pad, 1));
            pack_tdnn_weight_mx8(wf.data(), out_ch, in_ch, 1, 0, &tap0, 1, cur.cout_pad, cur.cin_pad, cur.w_scale_fold, w8.data());
            if ((rc = dev_upload(net, w8.data(), w8.size(), &cur.w8_fold))) return rc;
          }
        } else {
          pack_tdnn_weight_frags(wf.data(), out_ch, in_ch, 1, 0, &tap0, 1, cur.cout_pad, cur.cin_pad, frags.data(), nullptr, net->frames_et());
        }
        if ((rc = dev_upload(net, frags.data(), frags.size() * 2, &cur.wfrag_fold))) return rc;
        void *bdev = nullptr;
        if ((rc = dev_upload(net, bf.data(), bf.size() * sizeof(float), &bdev))) return rc;
        cur.bias_fold = reinterpret_cast<float *>(bdev);
      }
    }
  }
  // f32m: a frame-level TDNN layer whose whole output buffer is read by exactly one other TDNN layer, as its input rows, may hand it over
  // as IMAGES (the split [hi halves | x_lo8 | x_hi8] the reader would otherwise make of the f32 rows per workgroup and chunk): recorded
  // here, decided per run (run_ops: both kernels must be the image-capable ones for that batch)
  if (net->x3_mx()) {
    for (size_t i = 0; i + 1 < net->ops.size(); ++i) {
      Op &o = net->ops[i];
      if (o.kind != OP_TDNN || o.utts || o.w8 == nullptr || o.fused_pool >= 0 || o.chain_last >= 0) continue;
      const auto &d = o.tdnn;
      const Buffer &b = net->bufs[d.out_buf];
      if (b.domain != ASV_DOMAIN_FRAMES || d.out_ch_off != 0 || d.out_ch != b.channels || b.channels % 32 != 0 || d.res_buf >= 0) continue;
      bool inside_chain = false;
      for (size_t h = 0; h < i; ++h) inside_chain |= net->ops[h].chain_last >= (int)i;
      if (inside_chain) continue;
      for (size_t j = i + 1; j < net->ops.size(); ++j) {
        const Op &r = net->ops[j];
        if (r.kind != OP_TDNN || r.tdnn.in_buf != d.out_buf) continue;
        const auto &rd = r.tdnn;
        const bool clean = rd.in_ch_off == 0 && rd.in_ch == b.channels && rd.in2_buf < 0 && rd.res_buf != d.out_buf && rd.seg_bias_buf != d.out_buf &&
                           rd.seg_scale_buf != d.out_buf && !r.utts && (r.chain_last >= 0 || (r.w8 != nullptr && r.fused_pool < 0));
        if (clean && sole_reader(d.out_buf, j)) o.image_reader = (int)j;
        break;
      }
    }
  }
  for (Op &o : net->ops) {                       // the host copies have served their purpose
    std::vector<float>().swap(o.host_w); std::vector<float>().swap(o.host_bias);
    std::vector<float>().swap(o.host_scale); std::vector<float>().swap(o.host_shift);
  }
  net->out_buf = out_buf; net->embed_dim = embed_dim; net->finalized = true;
  net->arena.resize(net->bufs.size());
  return ASV_OK;
}

int asv_net_embed_dim(const asv_net_t *net) { return net ? net->embed_dim : ASV_EINVAL; }

int asv_net_status(asv_net_t *net, unsigned *status, void *stream) {
  ASV_REQUIRE(net != nullptr && status != nullptr, "asv_net_status: null argument");
  ASV_ON_DEVICE(net->device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  uint32_t word = 0;
  ASV_HIP_CHECK(hipMemcpyAsync(&word, status_word(net), sizeof(word), hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemsetAsync(status_word(net), 0, sizeof(word), s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  *status = word;
  return ASV_OK;
}

int asv_net_status_async(asv_net_t *net, unsigned *host_status, void *stream) {
  ASV_REQUIRE(net != nullptr && host_status != nullptr, "asv_net_status_async: null argument");
  ASV_ON_DEVICE(net->device);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ASV_HIP_CHECK(hipMemcpyAsync(host_status, status_word(net), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemsetAsync(status_word(net), 0, sizeof(uint32_t), s));
  return ASV_OK;
}

unsigned long long asv_kernel_launch_count(int which) {
  return (which >= ASV_KERNEL_TDNN_P8 && which <= ASV_KERNEL_TDNN_X3M_IMAGE) ? g_kernel_launches[which].load() : 0ull;
}

size_t asv_net_device_bytes(const asv_net_t *net) {
  if (!net) return 0;
  size_t n = net->weight_bytes + net->meta_dev.cap + net->rowmeta_dev.cap;
  for (auto &m : net->arena) n += m.cap;
  return n;
}

int asv_net_describe(const asv_net_t *net, char *buf, size_t cap) {
  if (!net || !buf || cap == 0) return ASV_EINVAL;
  std::string s;
  char line[512];
  snprintf(line, sizeof(line), "asv_net precision=%s flags=%u feat_dim=%d buffers=%zu ops=%zu out=%d embed_dim=%d\n",
           net->precision == ASV_PREC_BF16 ? "bf16" : (net->precision == ASV_PREC_F16 ? "f16" : (net->precision == ASV_PREC_F32X ? "f32x" : "f32")), net->flags, net->feat_dim, net->bufs.size(), net->ops.size(), net->out_buf, net->embed_dim);
  s += line;
  for (size_t i = 0; i < net->bufs.size(); ++i) {
    const Domain &dm = net->domains[net->bufs[i].domain];
    char dname[64];
    if (dm.kind == 2) snprintf(dname, sizeof(dname), "grid(T/%d x %d, pitch %d)", 1 << dm.shift, dm.width, dm.pitch);
    else snprintf(dname, sizeof(dname), "%s", dm.kind == ASV_DOMAIN_FRAMES ? "frames" : "utts");
    snprintf(line, sizeof(line), "  buf %zu: %s channels=%d ld=%d\n", i, dname, net->bufs[i].channels, net->bufs[i].ld);
    s += line;
  }
  for (size_t i = 0; i < net->ops.size(); ++i) {
    const Op &op = net->ops[i];
    switch (op.kind) {
      case OP_TDNN: {
        const auto &d = op.tdnn;
        std::string taps;
        for (int t = 0; t < d.n_taps; ++t) { taps += (t ? "," : ""); taps += std::to_string(d.taps[t]); }
        snprintf(line, sizeof(line), "  op %zu: tdnn %d[%d:+%d]%s -> %d[%d:+%d] taps=[%s] act1=%d affine=%d first=%d act2=%d segbias=%d segscale=%d res=%d\n", i,
                 d.in_buf, d.in_ch_off, d.in_ch, d.in2_buf >= 0 ? "+in2" : "", d.out_buf, d.out_ch_off, d.out_ch, taps.c_str(), d.act1, (int)op.has_affine,
                 d.affine_first, d.act2, d.seg_bias_buf, d.seg_scale_buf, d.res_buf);
        break;
      }
      case OP_POOL:
        snprintf(line, sizeof(line), "  op %zu: stats_pool %d[%d:+%d] -> %d[%d] stddev=%d unbiased=%d var_mode=%d eps=%g\n", i, op.pool.in_buf, op.pool.in_ch_off,
                 op.pool.channels, op.pool.out_buf, op.pool.out_ch_off, op.pool.stddev, op.pool.unbiased, op.pool.var_mode, op.pool.eps);
        break;
      case OP_ATTPOOL:
        snprintf(line, sizeof(line), "  op %zu: attentive_pool x=%d logits=%d channels=%d -> %d[%d] eps=%g\n", i, op.att.x_buf, op.att.logit_buf, op.att.channels,
                 op.att.out_buf, op.att.out_ch_off, op.att.eps);
        break;
      case OP_LDE:
        snprintf(line, sizeof(line), "  op %zu: lde_pool x=%d channels=%d centres=%d -> %d[%d]\n", i, op.lde.x_buf, op.lde.channels, op.lde.n_centres, op.lde.out_buf,
                 op.lde.out_ch_off);
        break;
      case OP_RES2:
        snprintf(line, sizeof(line), "  op %zu: res2 %d[%d] -> %d[%d] branches=%d dilation=%d\n", i, op.res2.in_buf, op.res2.in_ch_off, op.res2.out_buf, op.res2.out_ch_off,
                 op.res2.branches, op.res2.dilation);
        break;
      case OP_GRID_INPUT:
        snprintf(line, sizeof(line), "  op %zu: grid_input %d -> %d\n", i, op.gin.in_buf, op.gin.out_buf);
        break;
      case OP_FLATTEN:
        snprintf(line, sizeof(line), "  op %zu: grid_flatten %d -> %d\n", i, op.flat.in_buf, op.flat.out_buf);
        break;
      case OP_IM2COL:
        snprintf(line, sizeof(line), "  op %zu: im2col %d -> %d taps=%d stride=%d channels=%d\n", i, op.i2c.in_buf, op.i2c.out_buf, op.i2c.n_taps, op.i2c.stride, op.i2c.channels);
        break;
      case OP_ELTWISE:
        snprintf(line, sizeof(line), "  op %zu: eltwise a=%d b=%d c=%d segscale=%d affine=%d channels=%d -> %d[%d]%s\n", i, op.elt.a_buf, op.elt.b_buf, op.elt.c_buf,
                 op.elt.seg_scale_buf, op.scale != nullptr, op.elt.channels, op.elt.out_buf, op.elt.out_ch_off, op.elt.out2_buf >= 0 ? " (+ second output)" : "");
        break;
    }
    s += line;
  }
  const size_t n = std::min(cap - 1, s.size());
  memcpy(buf, s.data(), n);
  buf[n] = 0;
  return (int)n;
}

int asv_net_set_profiling(asv_net_t *net, int enable) {
  ASV_REQUIRE(net != nullptr, "asv_net_set_profiling: null net");
  net->profiling = enable;
  return ASV_OK;
}

int asv_net_get_profile(asv_net_t *net, asv_kernel_time_t *rows, int cap, int *n_rows) {
  ASV_REQUIRE(net && rows && n_rows && cap >= 1, "asv_net_get_profile: bad arguments");
  ASV_ON_DEVICE(net->device);
  std::map<std::pair<int, int>, asv_kernel_time_t> agg;      // (kernel class, op or -1)
  for (auto &st : net->stamps) {
    ASV_HIP_CHECK(hipEventSynchronize(st.b));
    float ms = 0.0f;
    ASV_HIP_CHECK(hipEventElapsedTime(&ms, st.a, st.b));
    const int op = net->profiling == 2 ? st.op : -1;
    auto it = agg.find({st.kclass, op});
    if (it == agg.end()) {
      asv_kernel_time_t row;
      memset(&row, 0, sizeof(row));
      snprintf(row.name, sizeof(row.name), "%s", kKernelNames[st.kclass]);
      row.op_index = op;
      it = agg.emplace(std::make_pair(st.kclass, op), row).first;
    }
    it->second.launches += st.launches;
    it->second.total_ms += ms;
    it->second.flops += st.flops;
    net->event_pool.push_back(st.a);
    net->event_pool.push_back(st.b);
  }
  net->stamps.clear();
  int n = 0;
  for (auto &kv : agg)
    if (n < cap) rows[n++] = kv.second;
  *n_rows = n;
  return ASV_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// the launch sequence of one batch
namespace {

struct DomainRun {
  int rows_pad = 0;
  int32_t *seg_row0 = nullptr, *seg_len = nullptr, *row_seg = nullptr;
  uint32_t *row_valid = nullptr;
};

struct RunCtx {
  asv_net *net; hipStream_t s; BatchPlan bp;
  float *final_out = nullptr;      // asv_net_extract: the caller's result matrix (lets the last layer write it directly)
  bool final_written = false;
  int32_t *seg_src0, *seg_frames, *utt_seg0, *utt_nseg;     // device metadata
  std::vector<DomainRun> dom;
  std::vector<char> buf_image;     // per run: buffer holds images (TdnnKernelParams::y_image of its producer)
};

// what prepare() leaves behind for the next call with identical offsets
struct PlanCache {
  BatchPlan bp;
  int32_t *seg_src0, *seg_frames, *utt_seg0, *utt_nseg;
  std::vector<DomainRun> dom;
};
PlanCache &plan_cache_of(asv_net *net) {
  if (!net->plan_cache) {
    net->plan_cache = new PlanCache();
    net->plan_cache_free = [](void *p) { delete reinterpret_cast<PlanCache *>(p); };
  }
  return *reinterpret_cast<PlanCache *>(net->plan_cache);
}

int prepare(RunCtx &c, const int32_t *offsets, int n_utts, int max_chunk) {
  asv_net *net = c.net;
  int rc;
  ASV_REQUIRE(offsets != nullptr && n_utts >= 1, "extract: need at least one utterance");
  if (net->cache_valid && net->cached_max_chunk == max_chunk && (int)net->cached_offsets.size() == n_utts + 1 &&
      memcmp(net->cached_offsets.data(), offsets, (size_t)(n_utts + 1) * 4) == 0) {
    const PlanCache &pc = plan_cache_of(net);
    c.bp = pc.bp; c.seg_src0 = pc.seg_src0; c.seg_frames = pc.seg_frames; c.utt_seg0 = pc.utt_seg0; c.utt_nseg = pc.utt_nseg; c.dom = pc.dom;
    return ASV_OK;                                   // device tables, row maps and arena are still valid
  }
  net->cache_valid = false;
  if ((rc = make_plan(net, offsets, n_utts, max_chunk, c.bp))) return rc;
  const BatchPlan &bp = c.bp;
  const int S = bp.segments, B = bp.n_utts;
  const size_t ndom = net->domains.size();
  // ---- int32 metadata: one pinned staging buffer, one H2D copy
  size_t n_meta = (size_t)2 * S + 2 * B;
  for (auto &dp : bp.dom) n_meta += dp.seg_row0.size() + dp.seg_len.size();
  const size_t meta_bytes = n_meta * sizeof(int32_t);
  if (net->meta_inflight) { ASV_HIP_CHECK(hipEventSynchronize(net->meta_copied)); net->meta_inflight = false; }
  if (meta_bytes > net->meta_host_cap) {
    if (net->meta_host) ASV_HIP_CHECK(hipHostFree(net->meta_host));
    net->meta_host = nullptr;
    net->meta_host_cap = meta_bytes * 2 + 4096;
    ASV_HIP_CHECK(hipHostMalloc(&net->meta_host, net->meta_host_cap, hipHostMallocDefault));
  }
  if (!net->meta_copied) ASV_HIP_CHECK(hipEventCreateWithFlags(&net->meta_copied, hipEventDisableTiming));
  if ((rc = ensure(net->meta_dev, meta_bytes, c.s, false))) return rc;
  int32_t *h = reinterpret_cast<int32_t *>(net->meta_host);
  int32_t *d = reinterpret_cast<int32_t *>(net->meta_dev.ptr);
  size_t o = 0;
  auto put = [&](const std::vector<int32_t> &v, int32_t **dev) { memcpy(h + o, v.data(), v.size() * 4); *dev = d + o; o += v.size(); };
  put(bp.seg_src0, &c.seg_src0); put(bp.seg_frames, &c.seg_frames);
  put(bp.utt_seg0, &c.utt_seg0); put(bp.utt_nseg, &c.utt_nseg);
  c.dom.resize(ndom);
  size_t rm_words = 0;
  for (size_t i = 0; i < ndom; ++i) {
    put(bp.dom[i].seg_row0, &c.dom[i].seg_row0);
    put(bp.dom[i].seg_len, &c.dom[i].seg_len);
    c.dom[i].rows_pad = bp.dom[i].rows_pad;
    rm_words += (size_t)bp.dom[i].rows_pad + bp.dom[i].rows_pad / 32;
  }
  ASV_HIP_CHECK(hipMemcpyAsync(d, h, meta_bytes, hipMemcpyHostToDevice, c.s));
  ASV_HIP_CHECK(hipEventRecord(net->meta_copied, c.s));
  net->meta_inflight = true;
  // ---- row maps of every domain
  if ((rc = ensure(net->rowmeta_dev, rm_words * 4, c.s, false))) return rc;
  int32_t *r = reinterpret_cast<int32_t *>(net->rowmeta_dev.ptr);
  Prof prof{net, c.s};
  if ((rc = prof.begin(K_ROWMAP, 0))) return rc;
  for (size_t i = 0; i < ndom; ++i) {
    const Domain &dm = net->domains[i];
    c.dom[i].row_seg = r; r += c.dom[i].rows_pad;
    c.dom[i].row_valid = reinterpret_cast<uint32_t *>(r); r += c.dom[i].rows_pad / 32;
    const int nseg = (int)bp.dom[i].seg_row0.size();
    if ((rc = launch_rowmap(c.dom[i].seg_row0, c.dom[i].seg_len, nseg, c.dom[i].rows_pad, dm.kind == 2 ? dm.pitch : 1, dm.kind == 2 ? dm.width : 1,
                            c.dom[i].row_seg, c.dom[i].row_valid, c.s))) return rc;
  }
  if ((rc = prof.end())) return rc;
  // ---- activation arena
  for (size_t i = 0; i < net->bufs.size(); ++i) {
    const Buffer &b = net->bufs[i];
    if ((rc = ensure(net->arena[i], (size_t)c.dom[b.domain].rows_pad * b.ld * net->elem_size(b.domain), c.s, true))) return rc;
  }
  PlanCache &pc = plan_cache_of(net);
  pc.bp = c.bp; pc.seg_src0 = c.seg_src0; pc.seg_frames = c.seg_frames; pc.utt_seg0 = c.utt_seg0; pc.utt_nseg = c.utt_nseg; pc.dom = c.dom;
  net->cached_offsets.assign(offsets, offsets + n_utts + 1);
  net->cached_max_chunk = max_chunk;
  net->cache_valid = true;
  return ASV_OK;
}

unsigned char *view(RunCtx &c, int buf, int ch_off) {
  const Buffer &b = c.net->bufs[buf];
  return reinterpret_cast<unsigned char *>(c.net->arena[buf].ptr) + (size_t)ch_off * c.net->elem_size(b.domain);
}

int run_ops(RunCtx &c, size_t n_ops) {
  asv_net *net = c.net;
  const BatchPlan &bp = c.bp;
  Prof prof{net, c.s};
  int rc;
  const bool use_ref = (net->flags & ASV_FLAG_REF_KERNELS) != 0;
  c.buf_image.assign(net->bufs.size(), 0);
  // Kernel-selection switches (developer A/B aids; read once unless ASV_AMD_LIVE_TUNE is set - the in-process A/B tests switch that on
  // after the library's first launch, so THAT lookup cannot be cached: one getenv per run)
  const bool live_tune = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  auto tune = [&](const char *name, int dflt, int cached) { return live_tune ? (getenv(name) ? atoi(getenv(name)) : dflt) : cached; };
  static const int p8_env = getenv("ASV_AMD_P8") ? atoi(getenv("ASV_AMD_P8")) : 1;
  static const int p8x_env = getenv("ASV_AMD_P8X") ? atoi(getenv("ASV_AMD_P8X")) : 1;
  static const int x3m_env = getenv("ASV_AMD_X3M") ? atoi(getenv("ASV_AMD_X3M")) : 1;
  static const int img_env = getenv("ASV_AMD_X3M_IMAGE") ? atoi(getenv("ASV_AMD_X3M_IMAGE")) : 1;
  static const int chainm_rows_env = getenv("ASV_AMD_CHAINM_ROWS") != nullptr ? atoi(getenv("ASV_AMD_CHAINM_ROWS")) : 64;
  const int p8_on = tune("ASV_AMD_P8", 1, p8_env), p8x_on = tune("ASV_AMD_P8X", 1, p8x_env), x3m_on = tune("ASV_AMD_X3M", 1, x3m_env);
  const int img_on = tune("ASV_AMD_X3M_IMAGE", 1, img_env), chainm_rows = tune("ASV_AMD_CHAINM_ROWS", 64, chainm_rows_env);
  // "one round of tiles" = one 256 x 256 tile per CU of THIS device (the kernels size their persistent grids from the same count)
  static const long long cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return (long long)(n > 0 ? n : 256);
  }();
  // the kernel parameters of TDNN op k (views, weights, taps)
  auto fill_tdnn_params = [&](size_t k, TdnnKernelParams &p) {
    const Op &op = net->ops[k];
    const auto &d = op.tdnn;
    const int domid = net->bufs[d.in_buf].domain;
    const DomainRun &dr = c.dom[domid];
    const int et = net->dom_et(domid);
    memset(&p, 0, sizeof(p));
    p.et = et; p.x3_et = net->x3_et(); p.x3_terms = net->x3_terms(); p.w_unscale = 1.0f / op.w_scale;
    p.x3_tile = (net->flags & ASV_FLAG_X3_TILE128) ? 128 : 0;
    p.status = status_word(net);
    p.x = view(c, d.in_buf, d.in_ch_off); p.ldx = net->bufs[d.in_buf].ld;
    if (d.in2_buf >= 0) { p.x2 = view(c, d.in2_buf, d.in2_ch_off); p.ldx2 = net->bufs[d.in2_buf].ld; }
    p.w = op.w; p.bias = op.bias; p.scale = op.scale; p.shift = op.shift;
    if (d.seg_bias_buf >= 0) { p.seg_bias = reinterpret_cast<const float *>(net->arena[d.seg_bias_buf].ptr); p.ld_segbias = net->bufs[d.seg_bias_buf].ld; }
    if (d.seg_scale_buf >= 0) { p.seg_scale = reinterpret_cast<const float *>(net->arena[d.seg_scale_buf].ptr); p.ld_segscale = net->bufs[d.seg_scale_buf].ld; }
    if (d.res_buf >= 0) { p.res = view(c, d.res_buf, d.res_ch_off); p.ldres = net->bufs[d.res_buf].ld; }
    p.y = view(c, d.out_buf, d.out_ch_off); p.ldy = net->bufs[d.out_buf].ld;
    p.row_seg = dr.row_seg; p.row_valid = dr.row_valid; p.rows = dr.rows_pad;
    p.cin_pad = op.cin_pad; p.cout_store = op.cout_store;
    p.n_taps = d.n_taps;
    for (int t = 0; t < d.n_taps; ++t) { p.taps[t] = d.taps[t]; p.halo = std::max(p.halo, std::abs(d.taps[t])); }
    p.act1 = d.act1; p.act2 = d.act2; p.affine_first = d.affine_first;
    p.zero16 = net->zero_page;
    p.wfrag = op.wfrag; p.wlo = op.wlo; p.wconv = op.wconv; p.wx3p = op.wx3p; p.w8 = op.w8;
  };
  // f32x: the 8-phase three-product kernel takes the wide plain layers that fill whole rounds of 256 x 256 tiles (kernels_tdnn_p8x.hip; the
  // bits of tdnn_gemm_x3_kernel; ASV_AMD_P8X=0: off).  Production rule: at least one round of tiles AND a last round that is >= 85 % full -
  // the x-vector's tdnn2 at 256 utterances is 408 tiles = 1.6 rounds, where the finer 128-row tiles of tdnn_gemm_x3_kernel are as fast and
  // leave CUs to the other stream: -0.9 % on two streams, profiles/r5s_p8x_model_ab.txt; ECAPA's layers are 4.75 and 7.1 rounds
  auto p8x_rule = [&](const TdnnKernelParams &q, bool x3q, bool fuseq) {
    const long long tiles = (long long)(q.rows / 256) * (round_up(q.cout_store, 256) / 256);
    const bool fill = p8x_on > 1 ? tiles >= p8x_on : (tiles >= cus && tiles * 100 >= ((tiles + cus - 1) / cus) * cus * 85);
    return x3q && !fuseq && p8x_on != 0 && tdnn_p8x_supported(q) && fill;
  };
  // f32m: the 128-row kernel with its correction products on the scaled 8-bit instruction, where the three-product kernel would take its
  // 128-row tiles (ASV_AMD_X3M=0: off) and the 8-phase kernel does not (its fill rule: the x-vector's tdnn1 / tdnn2 - 163 us here against
  // 227 on the three-product kernel, profiles/r6f_*).  ECAPA's wide 1-tap layers stay on the 8-phase kernel: a new window (barrier +
  // conversion) per 32 channels makes this kernel only 4 - 12 % faster there on one stream and 8 % slower in the two-stream pipeline
  // (35.1 k against 38.2 k utterances/s, profiles/r6q_bench_line.json)
  auto x3m_rule = [&](const TdnnKernelParams &q, bool x3q, bool fuseq) {
    return x3q && !fuseq && !p8x_rule(q, x3q, fuseq) && x3m_on != 0 && net->x3_mx() && tdnn_x3m_supported(q) &&
           (long long)(q.rows / 128) * (round_up(q.cout_store, 256) / 256) >= (x3m_on > 1 ? x3m_on : 384);      // (384: where the three-product kernel takes its 128-row tiles)
  };
  auto x3_rule = [&](const Op &o, const TdnnKernelParams &q) {
    return !use_ref && !o.utts && net->x3() && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && tdnn_x3_supported(q);
  };
  // the chain kernels' fused pooling: utterances per block of `block_rows` rows, worst block (a crowd of tiny utterances -> per-layer path)
  auto chain_slots_uniform = [&](int block_rows, int n_blocks, int *min_len) {
    const DomainPlan &fp = bp.dom[ASV_DOMAIN_FRAMES];
    std::vector<int> per_block((size_t)n_blocks + 1, 0);
    int slots = 1, ml = 1 << 30;
    for (size_t sidx = 0; sidx < fp.seg_len.size(); ++sidx) ml = std::min(ml, (int)fp.seg_len[sidx]);
    for (size_t sidx = 0; sidx < fp.seg_row0.size(); ++sidx)
      for (int h = fp.seg_row0[sidx] / block_rows; h <= (fp.seg_row0[sidx] + fp.seg_len[sidx] - 1) / block_rows; ++h) slots = std::max(slots, ++per_block[h]);
    if (min_len != nullptr) *min_len = ml;
    return slots;
  };
  auto chain_is_mx = [&](size_t head) {
    bool mx = net->x3() && net->x3_mx();
    for (size_t k = head; k <= (size_t)net->ops[head].chain_last && mx; ++k)
      mx = (net->ops[k].wfrag_fold != nullptr ? net->ops[k].w8_fold : net->ops[k].w8) != nullptr;
    return mx;
  };
  auto chain_branch = [&](const Op &o, const TdnnKernelParams &q, bool rows16) {      // the condition under which op o runs as the head of a chain kernel
    return o.chain_last >= 0 && !use_ref && (rows16 || net->x3()) && q.halo <= kHalo && (net->flags & ASV_FLAG_SMALL_TILES) == 0 &&
           (net->x3() || (unsigned long long)q.rows * (unsigned long long)q.ldx * 2ull < (1ull << 32));
  };
  // f32m, op j reads ONE buffer as its input rows (Op::image_reader of the producer): will it, in THIS run, go to a kernel that reads
  // images?  Follows the selection below step by step - the chain branch first, then the per-layer rules - and the launches check it.
  auto reader_takes_image = [&](size_t j) {
    const Op &o = net->ops[j];
    TdnnKernelParams q;
    fill_tdnn_params(j, q);
    if (chain_branch(o, q, false)) {
      const bool mx = chain_is_mx(j), mx96 = mx && chainm_rows == 96 && q.rows >= 96;
      const int slots = mx96 ? chain_slots_uniform(96, chainm96_tiles(q.rows), nullptr) : chain_slots_uniform(64, q.rows >> 6, nullptr);
      if (slots <= 16) return mx && !mx96 && q.cin_pad % 32 == 0;
    }
    return o.fused_pool < 0 && x3m_rule(q, x3_rule(o, q), false) && tdnn_x3m_image_in_supported(q);
  };
  for (size_t i = 0; i < n_ops; ++i) {
    Op &op = net->ops[i];
    switch (op.kind) {
      case OP_TDNN: {
        const auto &d = op.tdnn;
        const int domid = net->bufs[d.in_buf].domain;
        const DomainRun &dr = c.dom[domid];
        const int et = net->dom_et(domid);
        const bool bf16 = et != ET_F32;              // 16-bit rows (bf16 or half)
        TdnnKernelParams p;
        fill_tdnn_params(i, p);
        p.x_image = c.buf_image[d.in_buf];
        const bool chain_x3 = net->x3();             // f32x: the split-product chain on 64-row tiles (kernels_tdnn_chainx.hip)
        if (chain_branch(op, p, bf16)) {
          // tdnn -> [1-tap]* -> 1-tap + pooling in one kernel, if the batch allows the fused pooling (no crowd of tiny utterances)
          const DomainPlan &fp = bp.dom[ASV_DOMAIN_FRAMES];
          const int tshift = chain_x3 ? 6 : 7;         // rows per pooling partial: the kernel's tile
          // the 16-bit chain runs batches of less than one round of workgroups in 96- / 64-frame tiles (ChainTilePlan); developer
          // runs with phase stamps keep 128-frame tiles throughout
          static const bool chain_dbg_on = getenv("ASV_AMD_CHAIN_DBG") != nullptr;
          ChainTilePlan plan;
          if (!chain_x3) plan = chain_tile_plan(p.rows, !chain_dbg_on);
          // f32m: the chain with its correction products on the scaled 8-bit instruction, when every layer of it has 8-bit fragments - in
          // 64-frame tiles (kernels_tdnn_chainm.hip).  ASV_AMD_CHAINM_ROWS=96 selects the 96-frame kernel (kernels_tdnn_chainm96.hip: the same
          // results, measured 13 % SLOWER - 507.6 against 450.1 us on one box, profiles/r6r_*: kept as the measurement it is)
          const bool chain_mx = chain_x3 && chain_is_mx(i);
          const bool mx96 = chain_mx && chainm_rows == 96 && p.rows >= 96;
          const int n_blocks = mx96 ? chainm96_tiles(p.rows) : (chain_x3 ? (p.rows >> tshift) : plan.tiles());
          int slots = 1, min_len = 1 << 30;
          if (chain_x3) {
            slots = chain_slots_uniform(mx96 ? 96 : 64, n_blocks, &min_len);
          } else {
            std::vector<int> per_half((size_t)n_blocks + 1, 0);
            for (size_t sidx = 0; sidx < fp.seg_len.size(); ++sidx) min_len = std::min(min_len, (int)fp.seg_len[sidx]);
            for (size_t sidx = 0; sidx < fp.seg_row0.size(); ++sidx)
              for (int h = plan.tile_of(fp.seg_row0[sidx]); h <= plan.tile_of(fp.seg_row0[sidx] + fp.seg_len[sidx] - 1); ++h) slots = std::max(slots, ++per_half[h]);
          }
          if (slots <= 16) {
            ASV_REQUIRE(!p.x_image || (chain_mx && !mx96), "tdnn(chain): image rows reached a chain kernel that cannot read them (internal)");
            const size_t l = (size_t)op.chain_last;
            Op &lo = net->ops[l];
            TdnnChainParams cp;
            memset(&cp, 0, sizeof(cp));
            cp.x = p.x; cp.ldx = p.ldx; cp.rows = p.rows; cp.cin_pad = p.cin_pad; cp.n_taps = p.n_taps;
            for (int t = 0; t < p.n_taps; ++t) cp.taps[t] = p.taps[t];
            // a layer whose consumer holds folded weights (wfrag_fold) stores ReLU(acc) only: its scale / shift are not passed
            auto layer_of = [&](size_t k) {
              const Op &o = net->ops[k];
              const bool folded_in = o.wfrag_fold != nullptr;                                  // the previous layer's BN sits in these weights
              const bool folded_out = k < l && net->ops[k + 1].wfrag_fold != nullptr;          // this layer's BN sits in the next layer's
              TdnnChainLayer L;
              L.wfrag = folded_in ? o.wfrag_fold : o.wfrag; L.bias = folded_in ? o.bias_fold : o.bias;
              L.wlo = folded_in ? o.wlo_fold : o.wlo; L.w_scale = folded_in ? o.w_scale_fold : o.w_scale;
              L.w8 = folded_in ? o.w8_fold : o.w8;
              L.scale = folded_out ? nullptr : o.scale; L.shift = folded_out ? nullptr : o.shift;
              L.relu = o.tdnn.act1 == ASV_ACT_RELU; L.cout_pad = o.cout_pad;
              return L;
            };
            cp.first = layer_of(i);
            cp.n_mid = (int)(l - i - 1);
            for (size_t k = i + 1; k < l; ++k) cp.mid[k - i - 1] = layer_of(k);
            cp.last = layer_of(l);
            cp.pool_slots = slots; cp.ld_partial = lo.cout_pad; cp.row_seg = dr.row_seg;
            cp.et = chain_x3 ? net->x3_et() : et;
            cp.min_seg_len = min_len;
            cp.status = status_word(net);
            static const bool chainm_group_off = getenv("ASV_AMD_CHAINM_GROUP") != nullptr && atoi(getenv("ASV_AMD_CHAINM_GROUP")) == 0;      // measuring aid (same results)
            cp.x_image = p.x_image ? (chainm_group_off ? 2 : 1) : 0;
            static const int chainm_abl = getenv("ASV_AMD_CHAINM_ABL") != nullptr ? atoi(getenv("ASV_AMD_CHAINM_ABL")) : 0;       // developer aid, read once
            cp.abl = (chainm_abl & ~8) != 0 && getenv("ASV_AMD_CHAIN_DBG") == nullptr ? (chainm_abl & 8) : chainm_abl;               // the garbage-result bits only under ASV_AMD_CHAIN_DBG
            cp.n128 = plan.n128; cp.n_tail = plan.n_tail; cp.tail_rows = plan.tail_rows;
            if ((rc = ensure(net->poolpart_dev, (size_t)n_blocks * slots * 2 * 3 * cp.ld_partial * 4, c.s, false))) return rc;
            cp.pool_partial = reinterpret_cast<float *>(net->poolpart_dev.ptr);
            double fl = 0.0;
            for (size_t k = i; k <= l; ++k) fl += 2.0 * (double)bp.frames * net->ops[k].tdnn.in_ch * net->ops[k].tdnn.out_ch * net->ops[k].tdnn.n_taps;
            if ((rc = prof.begin(K_TDNN, fl, (int)i))) return rc;
            static const int chain_dbg = getenv("ASV_AMD_CHAIN_DBG") != nullptr ? std::max(1, atoi(getenv("ASV_AMD_CHAIN_DBG"))) : 0;   // developer aid: phase durations to stderr
            DevMem dbg;
            const size_t dbg_wgs = mx96 ? (size_t)n_blocks : (chain_mx ? (size_t)(p.rows / 64) : (size_t)(p.rows / 128));
            if (chain_dbg && (!chain_x3 || chain_mx)) {
              if ((rc = ensure(dbg, dbg_wgs * 8 * 32 * 8, c.s, true))) return rc;
              cp.dbg = reinterpret_cast<unsigned long long *>(dbg.ptr);
              cp.dbg_fine = chain_dbg >= 3;

            }
            if (chain_mx) ++g_kernel_launches[ASV_KERNEL_TDNN_CHAINM];
            if ((rc = mx96 ? launch_tdnn_chainm96(cp, c.s) : (chain_mx ? launch_tdnn_chainm(cp, c.s) : (chain_x3 ? launch_tdnn_chainx(cp, c.s) : launch_tdnn_chain(cp, c.s))))) return rc;
            if ((rc = prof.end())) return rc;
            if (chain_dbg && (!chain_x3 || chain_mx)) {
              const size_t nwg = dbg_wgs;
              std::vector<unsigned long long> h(nwg * 8 * 32);
              ASV_HIP_CHECK(hipStreamSynchronize(c.s));
              ASV_HIP_CHECK(hipMemcpy(h.data(), dbg.ptr, h.size() * 8, hipMemcpyDeviceToHost));
              ASV_HIP_CHECK(hipFree(dbg.ptr));
              double sum[32] = {0}; size_t cnt = 0; double cyc = 0, rt = 0;
              for (size_t w = 0; w < nwg * 8; ++w) {
                const unsigned long long *t = &h[w * 32];
                if (t[0] == 0) continue;
                for (int k = 1; k < 14; ++k) if (t[k] > t[k - 1]) sum[k] += (double)(t[k] - t[k - 1]);      // 13: the 4-wave kernel's drain
                if (chain_dbg >= 3) {
                  if (t[16] > t[7]) sum[16] += (double)(t[16] - t[7]);                                   // unit's K loop end -> epilogue body
                  for (int k = 17; k < 22; ++k) if (t[k] > t[k - 1]) sum[k] += (double)(t[k] - t[k - 1]);
                }
                const unsigned long long t_end = t[13] > t[12] ? t[13] : t[12];
                if (t_end > t[0] && t[15] > t[14]) { cyc += (double)(t_end - t[0]); rt += (double)(t[15] - t[14]); }
                ++cnt;
              }
              fprintf(stderr, "[chain dbg] %zu waves, mean cycles per phase:", cnt);
              fprintf(stderr, " [shader clock %.0f MHz, %.1f us per workgroup]", rt > 0 ? 100.0 * cyc / rt : 0.0, cnt ? rt / 100.0 / (double)cnt : 0.0);
              for (int k = 1; k < 14; ++k) if (k < 13 || sum[k] > 0) fprintf(stderr, " %d:%.0f", k, sum[k] / (double)std::max<size_t>(cnt, 1));
              if (chain_dbg >= 3) {
                fprintf(stderr, " | first epilogue: entry %.0f, fragments", sum[16] / (double)std::max<size_t>(cnt, 1));
                for (int k = 17; k < 21; ++k) fprintf(stderr, " %.0f", sum[k] / (double)std::max<size_t>(cnt, 1));
                fprintf(stderr, ", publish %.0f", sum[21] / (double)std::max<size_t>(cnt, 1));
              }
              fprintf(stderr, "\n");
              if (chain_mx && chain_dbg >= 3) {                      // layer A, steps 0 .. 14: mean cycles per step, by wave
                for (int wv = 0; wv < 8; ++wv) {
                  double d[15] = {0}; size_t cn = 0;
                  for (size_t wg = 0; wg < nwg; ++wg) {
                    const unsigned long long *t = &h[(wg * 8 + wv) * 32];
                    if (t[16] == 0 || t[31] <= t[16]) continue;
                    for (int k = 0; k < 15; ++k) d[k] += (double)(t[17 + k] - t[16 + k]);
                    ++cn;
                  }
                  fprintf(stderr, "[chainm dbg] wave %d, cycles of layer A's steps 0..14:", wv);
                  for (int k = 0; k < 15; ++k) fprintf(stderr, " %.0f", d[k] / (double)std::max<size_t>(cn, 1));
                  fprintf(stderr, "\n");
                }
              }
              {                                                       // 4-wave kernel, ablation 8: fine stamps of wave 0's third unit (slots of wave 4..7)
                double fs[25] = {0}; size_t fc = 0;
                for (size_t wg = 0; wg < nwg; ++wg) {
                  const unsigned long long *t = &h[(wg * 8 + 4) * 32];
                  if (t[0] == 0 || t[24] <= t[0]) continue;
                  for (int k = 1; k < 25; ++k) fs[k] += (double)(t[k] - t[k - 1]);
                  ++fc;
                }
                if (fc) {
                  fprintf(stderr, "[chain dbg] third unit of wave 0, %zu workgroups, per chunk: pre | 32 matrix instructions | post:", fc);
                  for (int c = 0; c < 8; ++c) fprintf(stderr, "  %.0f | %.0f | %.0f", fs[3 * c + 1] / fc, fs[3 * c + 2] / fc, fs[3 * c + 3] / fc);
                  fprintf(stderr, "\n");
                }
              }
              if (chain_dbg == 2 || chain_dbg >= 4) {                 // raw timelines of two workgroups: waves w and w + 4 share a SIMD
                const size_t picks[2] = {nwg / 4, nwg / 2 + 1};
                for (size_t wg : picks) {
                  if (wg >= nwg) continue;
                  unsigned long long t0 = ~0ull;
                  for (int w = 0; w < 8; ++w) if (h[(wg * 8 + w) * 32] != 0) t0 = std::min(t0, h[(wg * 8 + w) * 32]);
                  for (int w = 0; w < 8; ++w) {
                    fprintf(stderr, "[chain dbg] workgroup %zu wave %d stamps (cycles since the first wave's stamp 0):", wg, w);
                    for (int k = 0; k < 13; ++k) fprintf(stderr, " %lld", (long long)(h[(wg * 8 + w) * 32 + k] - t0));
                    if (chain_dbg >= 4) { fprintf(stderr, " |"); for (int k = 16; k < 22; ++k) fprintf(stderr, " %lld", (long long)(h[(wg * 8 + w) * 32 + k] - t0)); }
                    fprintf(stderr, "\n");
                  }
                }
              }
            }
            Op &po = net->ops[lo.fused_pool];
            po.skipped = true;
            const auto &q = po.pool;
            PoolFinishParams f;
            f.partial = cp.pool_partial; f.ld_partial = cp.ld_partial; f.pool_slots = slots; f.lh_split = 1; f.tile_shift = tshift;
            f.rows_shift = plan.rows128(); f.tail_rows = plan.n_tail > 0 ? plan.tail_rows : 0; f.n_shift = plan.n128;
            if (mx96) { f.rows_shift = 0; f.tail_rows = 96; f.n_shift = 0; }       // uniform 96-row blocks (pool_finish_kernel's tail form from row 0 on)
            f.row_seg = dr.row_seg; f.rows = dr.rows_pad; f.seg_row0 = dr.seg_row0; f.seg_len = dr.seg_len;
            f.shift = lo.shift;
            f.out = reinterpret_cast<float *>(net->arena[q.out_buf].ptr) + q.out_ch_off; f.ld_out = net->bufs[q.out_buf].ld; f.channels = q.channels;
            f.stddev = q.stddev; f.unbiased = q.unbiased; f.var_mode = q.var_mode; f.eps = q.eps;
            if ((rc = prof.begin(K_POOL, 0, lo.fused_pool))) return rc;
            if ((rc = launch_pool_finish(f, bp.segments, c.s))) return rc;
            if ((rc = prof.end())) return rc;
            i = l;                                       // the chain's other layers ran inside the kernel
            break;
          }
        }
        const bool narrow = p.halo <= kHalo;
        // fused statistics pooling: needs few enough segments per 128-row half-tile (i.e. no tiny utterances)
        int pool_slots = 0;
        if (op.fused_pool >= 0 && !use_ref && (net->flags & ASV_FLAG_SMALL_TILES) == 0) {
          const DomainPlan &fp = bp.dom[ASV_DOMAIN_FRAMES];
          std::vector<int> per_half((size_t)fp.rows_pad / 128 + 1, 0);
          int worst = 1;
          for (size_t sidx = 0; sidx < fp.seg_row0.size(); ++sidx)
            for (int h = fp.seg_row0[sidx] >> 7; h <= (fp.seg_row0[sidx] + fp.seg_len[sidx] - 1) >> 7; ++h) worst = std::max(worst, ++per_half[h]);
          pool_slots = worst;
          if (pool_slots > 16) pool_slots = 0;               // many tiny utterances: use the separate pooling kernel
        }
        const bool big3 = !use_ref && narrow && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && tdnn_big3_supported(p, et, !bf16);
        const bool utts_kernel = !use_ref && op.utts;
        const bool x3 = x3_rule(op, p);
        const bool c1_conv = !use_ref && net->domains[domid].kind == 2 && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && grid_conv_c1_supported(p, et, d.in_ch);
        const bool narrow_conv = !use_ref && net->domains[domid].kind == 2 && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && grid_conv_narrow_supported(p, et);
        const bool wide_conv = !use_ref && net->domains[domid].kind == 2 && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && grid_conv_wide_supported(p, et);
        const bool s2d_conv = !use_ref && net->domains[domid].kind == 2 && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && grid_conv_s2d_supported(p, et);
        const bool x3_conv = !use_ref && !op.utts && net->x3() && !x3 && (net->flags & ASV_FLAG_SMALL_TILES) == 0 && grid_conv_x3_supported(p);
        if (!use_ref && !big3 && op.utts && !utts_kernel) {
          // pooled-domain layers have one row per utterance (M is tiny, K is large): slice K over more
          // workgroups.  The slice count depends on K only, never on the batch, so an utterance's
          // embedding is bit-identical whatever batch it is extracted in.
          const int nchunks = (p.cin_pad + (bf16 ? 64 : 32) - 1) / (bf16 ? 64 : 32);
          p.ksplit = std::min(nchunks / 4, 24);
          if (p.ksplit > 1) {
            p.ld_partial = round_up(p.cout_store, 64);
            if ((rc = ensure(net->splitk_dev, (size_t)p.ksplit * p.rows * p.ld_partial * 4, c.s, false))) return rc;
            p.partial = reinterpret_cast<float *>(net->splitk_dev.ptr);
          } else {
            p.ksplit = 0;
          }
        }
        double valid_rows = op.utts ? (double)bp.segments : (double)bp.frames;
        if (net->domains[domid].kind == 2) {
          valid_rows = 0;
          for (int32_t len : bp.dom[domid].seg_len) valid_rows += (double)(len / net->domains[domid].pitch) * net->domains[domid].width;
        }
        if ((rc = prof.begin(op.utts ? K_UTTS : K_TDNN, 2.0 * valid_rows * d.in_ch * d.out_ch * d.n_taps * (d.alg_fraction > 0.0f ? (double)d.alg_fraction : 1.0), (int)i))) return rc;
        const bool fuse = pool_slots > 0 && (big3 || (x3 && tdnn_x3_pool_supported(p)));
        if (fuse) {
          p.pool_slots = pool_slots;
          p.ld_partial = op.cout_pad;
          if ((rc = ensure(net->poolpart_dev, (size_t)(p.rows / 128) * pool_slots * 3 * p.ld_partial * 4, c.s, false))) return rc;
          p.pool_partial = reinterpret_cast<float *>(net->poolpart_dev.ptr);
        }
        // Round 5: the layers of the variant-3 kernel with the plain epilogue, whole 64-channel chunks and at least one round of 256 x 256
        // tiles on the chip's CUs go to the 8-phase kernel (kernels_tdnn_p8.hip: both operands through LDS-DMA, staggered wave rows;
        // bit-identical outputs, 1.03 - 1.15 x the rate: profiles/r5e_p8_shapes.txt).  ASV_AMD_P8=0: the variant-3 kernel everywhere.
        const bool p8 = big3 && !fuse && p8_on != 0 && tdnn_p8_supported(p, et, !bf16) &&
                        (long long)(p.rows / 256) * (round_up(p.cout_store, 256) / 256) >= (p8_on > 1 ? p8_on : cus);
        // ... and the f32x / f32m forms (the rules: p8x_rule, x3m_rule above)
        const bool p8x = p8x_rule(p, x3, fuse);
        const bool x3m = x3m_rule(p, x3, fuse);
        ASV_REQUIRE(!p.x_image || (x3m && tdnn_x3m_image_in_supported(p)), "tdnn: image rows reached a kernel that cannot read them (internal)");
        // f32m: the output rows as images, when their one reader will take them as such in this run (Op::image_reader; ASV_AMD_X3M_IMAGE=0: off)
        if (x3m && img_on != 0 && op.image_reader >= 0 && (size_t)op.image_reader < n_ops && tdnn_x3m_image_out_supported(p) &&
            reader_takes_image((size_t)op.image_reader)) {
          p.y_image = 1;
          c.buf_image[d.out_buf] = 1;
          ++g_kernel_launches[ASV_KERNEL_TDNN_X3M_IMAGE];
        }
        if (use_ref) rc = launch_tdnn_ref(p, et, !bf16, c.s);
        else if (utts_kernel) {
          // last layer, every utterance a single chunk: the kernel also produces the caller's [utterance][embed_dim] result
          if (c.final_out != nullptr && i + 1 == net->ops.size() && d.out_buf == net->out_buf && d.out_ch_off == 0 && d.out_ch == net->embed_dim &&
              bp.segments == bp.n_utts) {
            p.final_out = c.final_out; p.final_ld = net->embed_dim; p.final_len = c.seg_frames;
            c.final_written = true;
          }
          // pooled-domain layers: split-bf16 products (f32-grade to ~2^-17, twice the rate of the f32-input MFMA) in the 16-bit
          // throughput modes; the exact f32-input MFMA in the parity modes - with IEEE-half operand halves in the frame layers
          // the bf16 split here would be the largest error left in the f32x mode (< 0.3 % of the FLOPs: +1 % of an f32x step)
          rc = launch_utts_gemm(p, bp.segments, net->frames_h16(), c.s);
        }
        else if (narrow_conv) rc = launch_grid_conv_narrow(p, c.s);
        else if (wide_conv) rc = launch_grid_conv_wide(p, c.s);
        else if (s2d_conv) rc = launch_grid_conv_s2d(p, c.s);
        else if (c1_conv) rc = launch_grid_conv_c1(p, c.s);
        else if (x3m) { rc = launch_tdnn_x3m(p, c.s); ++g_kernel_launches[ASV_KERNEL_TDNN_X3M]; }
        else if (p8x) { rc = launch_tdnn_p8x(p, c.s); ++g_kernel_launches[ASV_KERNEL_TDNN_P8X]; }
        else if (x3) rc = launch_tdnn_x3(p, c.s);
        else if (x3_conv) rc = launch_grid_conv_x3(p, c.s);
        else if (p8) { rc = launch_tdnn_p8(p, c.s); ++g_kernel_launches[ASV_KERNEL_TDNN_P8]; }
        else if (big3) { rc = launch_tdnn_big3(p, c.s); ++g_kernel_launches[ASV_KERNEL_TDNN_BIG3]; }
        else {
          rc = launch_tdnn_mfma(p, et, !bf16, c.s);
          if (!rc && p.ksplit > 1) rc = launch_splitk_epilogue(p, et, !bf16, c.s);
        }
        if (rc) return rc;
        if ((rc = prof.end())) return rc;
        if (op.fused_pool >= 0) {
          Op &po = net->ops[op.fused_pool];
          po.skipped = fuse;
          if (fuse) {
            const auto &q = po.pool;
            PoolFinishParams f;
            f.partial = p.pool_partial; f.ld_partial = p.ld_partial; f.pool_slots = pool_slots; f.lh_split = 0; f.tile_shift = 7;
            f.rows_shift = 0; f.tail_rows = 0; f.n_shift = 0;
            f.row_seg = dr.row_seg; f.rows = dr.rows_pad; f.seg_row0 = dr.seg_row0; f.seg_len = dr.seg_len;
            f.shift = op.shift;
            f.out = reinterpret_cast<float *>(net->arena[q.out_buf].ptr) + q.out_ch_off; f.ld_out = net->bufs[q.out_buf].ld; f.channels = q.channels;
            f.stddev = q.stddev; f.unbiased = q.unbiased; f.var_mode = q.var_mode; f.eps = q.eps;
            if ((rc = prof.begin(K_POOL, 0, op.fused_pool))) return rc;
            if ((rc = launch_pool_finish(f, bp.segments, c.s))) return rc;
            if ((rc = prof.end())) return rc;
          }
        }
        break;
      }
      case OP_POOL: {
        if (op.skipped) break;                       // folded into the producing layer's epilogue
        const auto &d = op.pool;
        const int domid = net->bufs[d.in_buf].domain;
        const Domain &dm = net->domains[domid];
        PoolKernelParams p;
        p.x = view(c, d.in_buf, d.in_ch_off); p.ldx = net->bufs[d.in_buf].ld; p.channels = d.channels;
        p.seg_row0 = c.dom[domid].seg_row0; p.seg_len = c.dom[domid].seg_len;
        p.out = reinterpret_cast<float *>(net->arena[d.out_buf].ptr) + d.out_ch_off; p.ld_out = net->bufs[d.out_buf].ld;
        p.stddev = d.stddev; p.unbiased = d.unbiased; p.var_mode = d.var_mode; p.eps = d.eps;
        p.row_stride = d.per_bin ? dm.pitch : 1;
        p.groups = d.per_bin ? dm.width : 1;
        p.chunk_partial = nullptr; p.chunks = 0; p.ld_chunk = 0;
        if (!d.stddev && !d.per_bin && dm.kind == 2 && dm.pitch >= 32 && (net->flags & ASV_FLAG_NO_FUSE) == 0) {      // (narrow maps: the second launch costs more than it saves)
          // mean over a 2-D map (SE squeeze): long segments, few of them - kPoolChunkRows rows per workgroup + a finish kernel
          int max_len = 0;
          for (int32_t len : bp.dom[domid].seg_len) max_len = std::max(max_len, (int)len);
          const int chunks = (max_len + kPoolChunkRows - 1) / kPoolChunkRows;
          if (chunks >= 1) {                                       // always this form for these ops: the arithmetic must not depend on the batch
            p.chunks = chunks; p.ld_chunk = round_up(d.channels, 64);
            if ((rc = ensure(net->poolpart_dev, (size_t)bp.segments * chunks * p.ld_chunk * 4, c.s, false))) return rc;
            p.chunk_partial = reinterpret_cast<float *>(net->poolpart_dev.ptr);
          }
        }
        if ((rc = prof.begin(K_POOL, 0, (int)i))) return rc;
        if ((rc = launch_stats_pool(p, bp.segments, net->dom_et(domid), c.s))) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_ATTPOOL: {
        const auto &d = op.att;
        const DomainRun &dr = c.dom[net->bufs[d.x_buf].domain];          // the frames domain, or a sequence domain behind the 2-D trunk
        if ((rc = prof.begin(K_ATT, 0, (int)i))) return rc;
        const int group = d.logit_group > 1 ? d.logit_group : (d.shared_logits ? d.channels : 1);
        rc = launch_attentive_pool(view(c, d.x_buf, d.x_ch_off), net->bufs[d.x_buf].ld, view(c, d.logit_buf, d.logit_ch_off), net->bufs[d.logit_buf].ld,
                                   d.channels, dr.seg_row0, dr.seg_len, bp.segments, d.eps,
                                   reinterpret_cast<float *>(net->arena[d.out_buf].ptr) + d.out_ch_off, net->bufs[d.out_buf].ld, net->frames_et(), group, d.logit_softplus2 != 0, op.scale, op.shift, c.s);
        if (rc) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_LDE: {
        const auto &d = op.lde;
        const DomainRun &dr = c.dom[net->bufs[d.x_buf].domain];
        if ((rc = ensure(net->lde_dev, (size_t)dr.rows_pad * 64 * 4, c.s, false))) return rc;
        if ((rc = prof.begin(K_ATT, 0, (int)i))) return rc;
        rc = launch_lde_pool(view(c, d.x_buf, d.x_ch_off), net->bufs[d.x_buf].ld, d.channels, dr.rows_pad, op.scale, op.shift, d.n_centres,
                             reinterpret_cast<float *>(net->lde_dev.ptr), dr.seg_row0, dr.seg_len, bp.segments,
                             reinterpret_cast<float *>(net->arena[d.out_buf].ptr) + d.out_ch_off, net->bufs[d.out_buf].ld, net->frames_et(), c.s);
        if (rc) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_ELTWISE: {
        const auto &d = op.elt;
        const int domid = net->bufs[d.a_buf].domain;
        EltwiseKernelParams p;
        memset(&p, 0, sizeof(p));
        p.a = view(c, d.a_buf, d.a_ch_off); p.lda = net->bufs[d.a_buf].ld;
        if (d.b_buf >= 0) { p.b = view(c, d.b_buf, d.b_ch_off); p.ldb = net->bufs[d.b_buf].ld; }
        if (d.c_buf >= 0) { p.c = view(c, d.c_buf, d.c_ch_off); p.ldc = net->bufs[d.c_buf].ld; }
        p.out = view(c, d.out_buf, d.out_ch_off); p.ldo = net->bufs[d.out_buf].ld;
        p.channels = d.channels;
        p.scale = op.scale; p.shift = op.shift;
        p.act = d.act;
        if (d.seg_scale_buf >= 0) { p.seg_scale = reinterpret_cast<const float *>(net->arena[d.seg_scale_buf].ptr); p.ld_segscale = net->bufs[d.seg_scale_buf].ld; }
        if (d.seg_norm_buf >= 0) { p.seg_norm = reinterpret_cast<const float *>(net->arena[d.seg_norm_buf].ptr); p.ld_segnorm = net->bufs[d.seg_norm_buf].ld; p.seg_norm_mode = d.seg_norm_mode; }
        if (d.out2_buf >= 0) {
          p.d = view(c, d.d_buf, d.d_ch_off); p.ldd = net->bufs[d.d_buf].ld;
          p.out2 = view(c, d.out2_buf, d.out2_ch_off); p.ldo2 = net->bufs[d.out2_buf].ld;
        }
        if (op.utts) { p.rows = bp.segments; }
        else { p.rows = c.dom[domid].rows_pad; p.row_seg = c.dom[domid].row_seg; p.row_valid = c.dom[domid].row_valid; }
        if ((rc = prof.begin(K_ELT, 0, (int)i))) return rc;
        if ((rc = launch_eltwise(p, net->dom_et(domid), c.s))) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_RES2: {
        const auto &d = op.res2;
        const DomainRun &dr = c.dom[ASV_DOMAIN_FRAMES];
        Res2KernelParams p;
        memset(&p, 0, sizeof(p));
        p.x = view(c, d.in_buf, d.in_ch_off); p.ldx = net->bufs[d.in_buf].ld;
        p.y = view(c, d.out_buf, d.out_ch_off); p.ldy = net->bufs[d.out_buf].ld;
        p.rows = dr.rows_pad; p.wfrag = op.wfrag; p.bias = op.bias; p.scale = op.scale; p.shift = op.shift; p.row_valid = dr.row_valid;
        p.branches = d.branches; p.dilation = d.dilation; p.et = net->frames_et();
        if ((rc = prof.begin(K_TDNN, 2.0 * (double)bp.frames * kRes2Width * kRes2Width * 3 * d.branches, (int)i))) return rc;
        static const bool res2_dbg = getenv("ASV_AMD_RES2_DBG") != nullptr;          // developer aid: phase durations to stderr
        DevMem dbg;
        if (res2_dbg) {
          if ((rc = ensure(dbg, (size_t)(p.rows / 128) * 8 * 16 * 8, c.s, true))) return rc;
          p.dbg = reinterpret_cast<unsigned long long *>(dbg.ptr);
        }
        if ((rc = launch_res2_chain(p, c.s))) return rc;
        if ((rc = prof.end())) return rc;
        if (res2_dbg) {
          const size_t nwg = (size_t)(p.rows / 128);
          std::vector<unsigned long long> h(nwg * 8 * 16);
          ASV_HIP_CHECK(hipStreamSynchronize(c.s));
          ASV_HIP_CHECK(hipMemcpy(h.data(), dbg.ptr, h.size() * 8, hipMemcpyDeviceToHost));
          ASV_HIP_CHECK(hipFree(dbg.ptr));
          double sum[16] = {0}; size_t cnt = 0; double cyc = 0, rt = 0;
          for (size_t w = 0; w < nwg * 8; ++w) {
            const unsigned long long *t = &h[w * 16];
            if (t[0] == 0) continue;
            for (int k = 1; k < 14; ++k) if (t[k] > t[k - 1]) sum[k] += (double)(t[k] - t[k - 1]);
            if (t[13] > t[0] && t[15] > t[14]) { cyc += (double)(t[13] - t[0]); rt += (double)(t[15] - t[14]); }
            ++cnt;
          }
          fprintf(stderr, "[res2 dbg] %zu waves [shader clock %.0f MHz, %.1f us per workgroup] 1 = first windows; per branch (1..3): K loop, wait + barrier, epilogue, "
                  "barrier; 13 = branches 4.. + end:", cnt, rt > 0 ? 100.0 * cyc / rt : 0.0, cnt ? rt / 100.0 / (double)cnt : 0.0);
          for (int k = 1; k < 14; ++k) fprintf(stderr, " %d:%.0f", k, sum[k] / (double)std::max<size_t>(cnt, 1));
          fprintf(stderr, "\n");
        }
        break;
      }
      case OP_GRID_INPUT: {
        const int domid = net->bufs[op.gin.out_buf].domain;
        const DomainRun &dg = c.dom[domid];
        if ((rc = prof.begin(K_GATHER, 0, (int)i))) return rc;
        rc = launch_grid_from_frames(net->arena[op.gin.in_buf].ptr, net->bufs[op.gin.in_buf].ld, net->feat_dim, c.dom[ASV_DOMAIN_FRAMES].seg_row0, dg.seg_row0, dg.row_seg, dg.row_valid,
                                     dg.rows_pad, net->domains[domid].pitch, net->arena[op.gin.out_buf].ptr, net->bufs[op.gin.out_buf].ld, net->frames_et(), c.s);
        if (rc) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_FLATTEN: {
        const auto &d = op.flat;
        const int din = net->bufs[d.in_buf].domain, dout = net->bufs[d.out_buf].domain;
        if ((rc = prof.begin(K_GATHER, 0, (int)i))) return rc;
        rc = launch_grid_flatten(net->arena[d.in_buf].ptr, net->bufs[d.in_buf].ld, net->bufs[d.in_buf].channels, net->domains[din].width, net->domains[din].pitch,
                                 c.dom[din].seg_row0, c.dom[dout].seg_row0, c.dom[dout].row_seg, c.dom[dout].row_valid, c.dom[dout].rows_pad,
                                 net->arena[d.out_buf].ptr, net->bufs[d.out_buf].ld, net->frames_et(), c.s);
        if (rc) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
      case OP_IM2COL: {
        const auto &d = op.i2c;
        const int din = net->bufs[d.in_buf].domain, dout = net->bufs[d.out_buf].domain;
        Im2colParams p;
        memset(&p, 0, sizeof(p));
        p.in = net->arena[d.in_buf].ptr; p.out = net->arena[d.out_buf].ptr;
        p.ldi = net->bufs[d.in_buf].ld; p.ldo = net->bufs[d.out_buf].ld;
        p.channels = d.channels; p.n_taps = d.n_taps; p.stride = d.stride;
        for (int t = 0; t < d.n_taps; ++t) { p.dt[t] = d.dt[t]; p.df[t] = d.df[t]; }
        p.in_row0 = c.dom[din].seg_row0; p.in_len = c.dom[din].seg_len;
        p.out_row0 = c.dom[dout].seg_row0; p.out_row_seg = c.dom[dout].row_seg; p.out_row_valid = c.dom[dout].row_valid;
        p.in_pitch = net->domains[din].pitch; p.in_width = net->domains[din].width;
        p.out_pitch = net->domains[dout].pitch; p.out_rows = c.dom[dout].rows_pad;
        if (d.b_buf >= 0) { p.b = net->arena[d.b_buf].ptr; p.ldb = net->bufs[d.b_buf].ld; }
        if (d.seg_scale_buf >= 0) { p.seg_scale = reinterpret_cast<const float *>(net->arena[d.seg_scale_buf].ptr); p.ld_segscale = net->bufs[d.seg_scale_buf].ld; }
        p.act = d.act;
        if ((rc = prof.begin(K_GATHER, 0, (int)i))) return rc;
        if ((rc = launch_im2col(p, net->frames_et(), c.s))) return rc;
        if ((rc = prof.end())) return rc;
        break;
      }
    }
  }
  return prof.close_span();
}

int pack_features(RunCtx &c, const float *feats) {
  Prof prof{c.net, c.s};
  int rc;
  const DomainRun &dr = c.dom[ASV_DOMAIN_FRAMES];
  if ((rc = prof.begin(K_PACK, 0))) return rc;
  if ((rc = launch_pack_input(feats, c.net->feat_dim, c.seg_src0, dr.seg_row0, dr.row_seg, dr.rows_pad, c.net->arena[0].ptr, c.net->bufs[0].ld,
                              c.net->frames_et(), c.s))) return rc;
  return prof.end();
}

}  // namespace

extern "C" {

int asv_net_extract(asv_net_t *net, const float *feats, const int32_t *offsets, int n_utts, float *out, int max_chunk, void *stream) {
  ASV_REQUIRE(net && net->finalized, "asv_net_extract: net is null or not finalized");
  ASV_REQUIRE(feats && out, "asv_net_extract: null feature or output pointer");
  ASV_ON_DEVICE(net->device);
  RunCtx c;
  c.net = net; c.s = reinterpret_cast<hipStream_t>(stream);
  int rc;
  if ((rc = prepare(c, offsets, n_utts, max_chunk))) return rc;
  if ((rc = pack_features(c, feats))) return rc;
  c.final_out = out;
  if ((rc = run_ops(c, net->ops.size()))) return rc;
  if (c.final_written) return ASV_OK;               // the last layer wrote `out` itself (single-chunk utterances)
  Prof prof{net, c.s};
  if ((rc = prof.begin(K_COMBINE, 0))) return rc;
  if ((rc = launch_combine(reinterpret_cast<const float *>(net->arena[net->out_buf].ptr), net->bufs[net->out_buf].ld, c.utt_seg0, c.utt_nseg, c.seg_frames, n_utts,
                           net->embed_dim, out, c.s))) return rc;
  return prof.end();
}

int asv_tdnn_forward(const asv_tdnn_desc_t *d, int precision, unsigned flags, const float *x, const int32_t *offsets, int n_utts, float *y, void *stream) {
  ASV_REQUIRE(d && x && y && offsets, "asv_tdnn_forward: null argument");
  int dev = 0;
  ASV_HIP_CHECK(hipGetDevice(&dev));
  asv_net_t *net = nullptr;
  int rc = asv_net_create(&net, dev, precision, flags, d->in_ch);
  if (rc) return rc;
  std::unique_ptr<asv_net_t, void (*)(asv_net_t *)> guard(net, asv_net_destroy);
  const int ob = asv_net_new_buffer(net, ASV_DOMAIN_FRAMES, d->out_ch);
  if (ob < 0) return ob;
  asv_tdnn_desc_t dd = *d;
  dd.in_buf = 0; dd.in_ch_off = 0; dd.in2_buf = -1; dd.out_buf = ob; dd.out_ch_off = 0;
  dd.seg_bias_buf = -1; dd.seg_scale_buf = -1; dd.res_buf = -1;
  if ((rc = asv_net_add_tdnn(net, &dd))) return rc;
  net->arena.resize(net->bufs.size());
  RunCtx c;
  c.net = net; c.s = reinterpret_cast<hipStream_t>(stream);
  if ((rc = prepare(c, offsets, n_utts, 1 << 30))) return rc;
  if ((rc = pack_features(c, x))) return rc;
  if ((rc = run_ops(c, 1))) return rc;
  const DomainRun &dr = c.dom[ASV_DOMAIN_FRAMES];
  if ((rc = launch_unpack_rows(net->arena[ob].ptr, net->bufs[ob].ld, d->out_ch, c.seg_src0, dr.seg_row0, dr.row_seg, dr.rows_pad, y, net->frames_et(), c.s))) return rc;
  ASV_HIP_CHECK(hipStreamSynchronize(c.s));
  return ASV_OK;
}

int asv_stats_pool_forward(const float *x, int channels, const int32_t *offsets, int n_utts, int stddev, int unbiased, int var_mode, float eps, float *y,
                           void *stream) {
  ASV_REQUIRE(x && y && offsets && channels >= 1, "asv_stats_pool_forward: bad argument");
  int dev = 0;
  ASV_HIP_CHECK(hipGetDevice(&dev));
  asv_net_t *net = nullptr;
  int rc = asv_net_create(&net, dev, ASV_PREC_F32, 0, channels);
  if (rc) return rc;
  std::unique_ptr<asv_net_t, void (*)(asv_net_t *)> guard(net, asv_net_destroy);
  const int out_ch = channels * (stddev ? 2 : 1);
  const int ob = asv_net_new_buffer(net, ASV_DOMAIN_UTTS, out_ch);
  if (ob < 0) return ob;
  asv_pool_desc_t pd;
  memset(&pd, 0, sizeof(pd));
  pd.struct_size = sizeof(pd); pd.in_buf = 0; pd.channels = channels; pd.out_buf = ob; pd.stddev = stddev; pd.unbiased = unbiased; pd.var_mode = var_mode; pd.eps = eps;
  if ((rc = asv_net_add_stats_pool(net, &pd))) return rc;
  net->arena.resize(net->bufs.size());
  RunCtx c;
  c.net = net; c.s = reinterpret_cast<hipStream_t>(stream);
  if ((rc = prepare(c, offsets, n_utts, 1 << 30))) return rc;
  if ((rc = pack_features(c, x))) return rc;
  if ((rc = run_ops(c, 1))) return rc;
  ASV_HIP_CHECK(hipMemcpy2DAsync(y, (size_t)out_ch * 4, net->arena[ob].ptr, (size_t)net->bufs[ob].ld * 4, (size_t)out_ch * 4, n_utts, hipMemcpyDeviceToDevice, c.s));
  ASV_HIP_CHECK(hipStreamSynchronize(c.s));
  return ASV_OK;
}

}  // extern "C"
