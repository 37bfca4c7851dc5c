// f32-grade TDNN / 1x1-conv implicit GEMM on the 16-bit matrix cores ("f32x" precision mode).
//
// Activations stay f32 in HBM (every other kernel of the f32 mode works on them unchanged).  This kernel
// splits both operands into two 16-bit halves, x = xh + xl, w = wh + wl, and accumulates
//     wh*xh + wh*xl + wl*xh                       (the dropped wl*xl term is ~2^-16 of a product for bf16 halves, ~2^-22 for
//                                                  IEEE-half ones - the default since round 3: asv_amd.h ASV_FLAG_X3_SPLIT_*)
// in the f32 accumulators of v_mfma_f32_32x32x16_{f16,bf16}: three matrix instructions per product instead of the
// exact-f32 v_mfma_f32_32x32x2_f32, whose rate is 1/16 of the bf16 one (MI355X_MICROARCH.md: 157 vs 2500 TFLOP/s).
// The same split is what kernels_utts.hip does for the pooled-domain layers.  Replaces, for the wide frame
// layers of the parity mode, F.conv1d + ReLU + eval BN of components.py:107-149, 410-431.
//
// Structure = kernels_tdnn_v3.hip with smaller pieces:
//   * the f32 feature window of a 32-channel chunk (72 frames x 128 B, shared by all taps) goes through a 4-stage
//     LDS ring by LDS-DMA (global_load_lds_dwordx4, XOR-swizzled 16-byte slots, one barrier per chunk);
//   * a lane's 8 consecutive k values are two ds_read_b128; the hi / lo split runs on the VALU
//     (v_cvt_pk_bf16_f32: hi; x - hi is exact in f32; v_cvt_pk_bf16_f32 again: lo) next to the MFMAs of the
//     previous k-group - the matrix and vector pipes issue independently;
//   * the weights are split once on the host into two fragment-ordered bf16 arrays (the layout of
//     pack_tdnn_weight_frags) and come straight from L2 as 1 KiB wave loads;
//   * 64 frames x 256 channels per workgroup, 4 waves (64 x 64 each = 2 x 2 accumulators), <= 168 VGPRs:
//     three workgroups per CU.
// Tried and dropped (r2p): splitting the f32 window ONCE per workgroup and chunk into a [hi | lo] bf16 image in a second LDS
// buffer (no VALU work in front of the k-groups instead of 48 operations per 12 MFMAs): 237.6 k vs 241.4 k utt/s on C2 - the
// kernel is not bound by the split but by what it streams: 4 KiB of weight fragments per 12 MFMAs and wave (64-row waves,
// hi + lo halves: 341 B per MFMA against 256 B in the bf16 kernel with 128-row waves) and f32 windows / outputs (the 1-tap
// 512 -> 1500 layer moves 1.05 GB per launch).  The next step for this mode is the 128-row wave tile of kernels_tdnn_v3.hip.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int XBN = 256;            // channels per workgroup
constexpr int XBK = 32;             // channels per chunk (128-byte f32 rows in LDS)
constexpr int XROWB = 128;
constexpr int XSTAGES = 4;
constexpr int XSPITCH = 68;                    // floats per epilogue scratch row
// MF = 32-frame accumulator fragments per wave:
//   MF = 2:  64 frames per workgroup, <= 168 VGPRs, 39 KiB of LDS: three workgroups per CU (round 2's geometry; kept for batches
//            whose 128-row tiles would not fill the chip);
//   MF = 4: 128 frames per workgroup (round 3), 72 KiB of LDS, two workgroups per CU: every weight fragment pair (hi, lo) a wave
//            fetches from L2 feeds 12 instead of 6 matrix instructions (171 instead of 341 bytes per instruction - the 64-row
//            kernel sat at the L1 fill rate), and the wave's 128 rows x 64 channels are what the fused statistics pooling
//            epilogue (POOL; the layout of kernels_tdnn_v3.hip's) works on.
template <int MF> struct X3Geom {
  static constexpr int BM = MF * 32;                  // frames per workgroup
  static constexpr int WIN = BM + 2 * kHalo;          // 72 | 136 window rows
  static constexpr int GROUPS = WIN / 8;              // 9 | 17 eight-row DMA pieces
  static constexpr int PIECES = (GROUPS + 3) / 4;     // 3 | 5 per wave
  static constexpr int STAGE = WIN * XROWB;           // 9216 | 17408 B
  static constexpr int RING = XSTAGES * STAGE;        // 36864 | 69632 B
  static_assert(4 * 32 * XSPITCH * 4 <= RING, "epilogue scratch must fit in the ring");
};
static_assert(XBN == kBigTileN, "weight padding must match the N tile");

typedef __attribute__((address_space(3))) unsigned char x3_lds_byte;

__device__ __forceinline__ int xswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void x3_glds16(const void *gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// X3Frag / x3_split (8 f32 -> the hi and lo halves in the 16-bit type): device_utils.h, shared with kernels_conv2d_x3.hip

// ET: the 16-bit type of the operand halves; TERMS: bit 0 = w_hi x_hi, bit 1 = w_hi x_lo, bit 2 = w_lo x_hi
// SHARED (round 3, 128-row geometry): the f32 window of a chunk is split into its [hi | lo] image ONCE per workgroup - every
//   thread converts two or three 8-value pieces - instead of by every wave for every tap inside the K loop (the four waves of a
//   workgroup multiply the SAME 128 rows by their own 64 channels: 4 waves x n_taps redundant splits, ~4 VALU operations per
//   matrix instruction - the K loop was VALU-bound at ~50 % matrix-pipe utilisation).  LDS: two f32 stages (LDS-DMA targets) +
//   two image buffers = the four stages of the other form; per chunk c, behind its one barrier: LDS-DMA of window c + 2 into
//   the stage window c has just been converted out of, conversion of window c + 1, K loop on image c (8 ds_read_b128 + 4 weight
//   fetches per 24 matrix instructions, no VALU).  Round 2 tried this on the 64-row geometry, whose K loop was bound by its
//   weight-fragment traffic instead, and saw nothing (profiles/r2p_*).
template <bool GENERIC, int ET, int TERMS, int MF = 2, bool POOL = false, bool SHARED = false>
__global__ __launch_bounds__(256, MF == 2 ? 3 : 2) void tdnn_gemm_x3_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  using Geo = X3Geom<MF>;
  constexpr int XBM = Geo::BM, XGROUPS = Geo::GROUPS, XPIECES = Geo::PIECES, XSTAGE = Geo::STAGE, XRING = Geo::RING;
  static_assert(!POOL || (MF == 4 && !GENERIC), "the fused pooling epilogue works on 128-row tiles with the plain epilogue");
  static_assert(!SHARED || (MF == 4 && TERMS == 7), "the shared split exists for the 128-row geometry with all three products");
  __shared__ __attribute__((aligned(16))) unsigned char lds[XRING + 3 * 256 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * XBM;
  const int n0 = (tile % n_tiles) * XBN;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *zero = reinterpret_cast<const unsigned char *>(p.zero16);
  const size_t x_pitch = (size_t)p.ldx * 4;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(x3_lds_byte *)lds);
  const int g_row = lane >> 3, g_slot = lane & 7;

  const int nchunks = (p.cin_pad + XBK - 1) / XBK;
  const int n_taps = p.n_taps;
  const int nkg = ((p.cin_pad + 63) / 64) * 4;                  // 16-channel k-groups per tap in the fragment arrays

  float *lds_par = reinterpret_cast<float *>(lds + XRING);
  if (tid < 192) {
    const int which = tid >> 6, idx = (tid & 63) * 4;
    float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = (which == 0) ? p.bias : (which == 1 ? p.scale : p.shift);
    if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + n0 + idx);
    *reinterpret_cast<float4 *>(lds_par + which * 256 + idx) = v;
  }

  // feature window of chunk c -> ring stage st (see kernels_tdnn_v3.hip: clamped rows are zero gap rows)
  size_t a_off[XPIECES];
#pragma unroll
  for (int i = 0; i < XPIECES; ++i) {
    const int grp = min(wn + i * 4, XGROUPS - 1);
    const int w = grp * 8 + g_row;
    const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
    a_off[i] = (size_t)row * x_pitch + (size_t)xswz(w, g_slot) * 16u;
  }
  auto issue_A = [&](int c, int st) {
    const unsigned char *base = xg + (size_t)c * XROWB;
    const bool tail = (c + 1) * XBK > p.cin_pad;
#pragma unroll
    for (int i = 0; i < XPIECES; ++i) {
      const int grp = min(wn + i * 4, XGROUPS - 1);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + st * XSTAGE + grp * 1024);
      const int w = grp * 8 + g_row;
      const bool ok = !tail || (c * XBK + xswz(w, g_slot) * 4 < p.cin_pad);
      x3_glds16(ok ? base + a_off[i] : zero, dst);
    }
  };

  const size_t frag_stride = (size_t)n_taps * nkg * 1024;      // bytes per 32-channel fragment
  const size_t wf_off0 = (size_t)((n0 + wn * 64) / 32) * frag_stride + (size_t)lane * 16;
  const unsigned char *wh_base = reinterpret_cast<const unsigned char *>(p.wfrag) + wf_off0;
  const unsigned char *wl_base = reinterpret_cast<const unsigned char *>(p.wlo) + wf_off0;

  uint32_t range = 0u;                       // range watch of the half split (device_utils.h)
  struct WFrags { uint4 h[2], l[2]; };
  struct XFrags { X3Frag f[MF]; };
  // k-group index g runs over (chunk, tap, half): g = (c * n_taps + t) * 2 + kg
  auto load_w = [&](int c, int t, int kg, WFrags &w) {
    const size_t off = ((size_t)t * nkg + (size_t)c * 2 + kg) * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      w.h[j] = *reinterpret_cast<const uint4 *>(wh_base + j * frag_stride + off);
      if constexpr ((TERMS & 4) != 0) w.l[j] = *reinterpret_cast<const uint4 *>(wl_base + j * frag_stride + off);
      else w.l[j] = make_uint4(0, 0, 0, 0);
    }
  };
  // SHARED: window c -> image buffer c & 1: row w = [hi of channels 0..31 (4 slots) | lo (4 slots)], slots swizzled like the window's
  auto convert = [&](int c) {
    const unsigned char *src = lds + (c & 1) * XSTAGE;
    unsigned char *dst = lds + (2 + (c & 1)) * XSTAGE;
#pragma unroll
    for (int it = 0; it < (Geo::WIN * 4 + 255) / 256; ++it) {
      const int item = it * 256 + tid;
      if (item < Geo::WIN * 4) {
        const int w = item >> 2, q = item & 3;
        const uint4 a = *reinterpret_cast<const uint4 *>(src + w * XROWB + xswz(w, 2 * q) * 16);
        const uint4 b = *reinterpret_cast<const uint4 *>(src + w * XROWB + xswz(w, 2 * q + 1) * 16);
        const X3Frag f = x3_split<ET, true>(a, b, range);
        *reinterpret_cast<uint4 *>(dst + w * XROWB + xswz(w, q) * 16) = f.hi;
        *reinterpret_cast<uint4 *>(dst + w * XROWB + xswz(w, 4 + q) * 16) = f.lo;
      }
    }
  };
  auto load_x = [&](int c, int d, int kg, XFrags &x) {
    if constexpr (SHARED) {
      const unsigned char *Ib = lds + (2 + (c & 1)) * XSTAGE;
      const int sh = kg * 2 + lh;
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int w = i * 32 + lr + kHalo + d;
        x.f[i].hi = *reinterpret_cast<const uint4 *>(Ib + w * XROWB + xswz(w, sh) * 16);
        x.f[i].lo = *reinterpret_cast<const uint4 *>(Ib + w * XROWB + xswz(w, 4 + sh) * 16);
      }
      return;
    }
    const unsigned char *Ab = lds + (c % XSTAGES) * XSTAGE;
    const int s0 = kg * 4 + lh * 2;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int w = i * 32 + lr + kHalo + d;
      const uint4 a = *reinterpret_cast<const uint4 *>(Ab + w * XROWB + xswz(w, s0) * 16);
      const uint4 b = *reinterpret_cast<const uint4 *>(Ab + w * XROWB + xswz(w, s0 + 1) * 16);
      x.f[i] = x3_split<ET, (TERMS & 2) != 0>(a, b, range);
    }
  };

  f32x16_t acc[MF][2];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto mma = [&](const XFrags &x, const WFrags &w) {
    // term-major order: every accumulator is touched once per 4 MFMAs (no back-to-back dependency)
#pragma unroll
    for (int term = 0; term < 3; ++term) {
      if (((TERMS >> term) & 1) == 0) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          const uint4 a = (term == 2) ? w.l[j] : w.h[j];
          const uint4 b = (term == 1) ? x.f[i].lo : x.f[i].hi;
          acc[i][j] = mfma16<ET>(a, b, acc[i][j]);
        }
    }
  };

  // ---- prologue: windows 0..2 in flight, fragments of k-group 0; wait for window 0 only
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  issue_A(0, 0);
  WFrags w0, w1;
  XFrags x0, x1;
  load_w(0, 0, 0, w0);
  if (nchunks > 1) issue_A(1, 1);
  if constexpr (SHARED) {
    // window 0 landed -> image 0; its stage then takes window 2.  Window 1 is converted at the top of chunk 0 (below).
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    convert(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // window 1 (this wave's pieces) landed, image 0 written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunks > 2) issue_A(2, 0);
    if (nchunks > 1) convert(1);
  } else {
    if (nchunks > 2) issue_A(2, 2);
    if (nchunks > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  load_x(0, __builtin_amdgcn_readlane(v_taps, 0), 0, x0);

  // ---- main loop over k-groups, two per trip (register sets 0 / 1 alternate: static indices only)
  const int per_chunk = n_taps * 2;
  const int G = nchunks * per_chunk;
  int c = 0, t = 0;                      // position of the k-group in flight (kg = g & 1)
  auto advance = [&](int g, XFrags &xn, WFrags &wn_) {
    // fetch k-group g + 1 (if any) into (xn, wn_); crossing into a new chunk first meets the workgroup
    const int kg = g & 1;
    int c2 = c, t2 = t, kg2 = kg + 1;
    if (kg2 == 2) { kg2 = 0; t2 = t + 1; if (t2 == n_taps) { t2 = 0; c2 = c + 1; } }
    if (g + 1 < G) {
      if (c2 != c) {
        if constexpr (SHARED) {
          // entering chunk c + 1: its image was written at the top of chunk c (by every wave: lgkmcnt), window c + 2 was
          // issued there too and is older than the youngest 4 operations (the fragments of k-group g).  Behind the barrier
          // nobody reads image c any more and window c + 1's stage is free: it takes window c + 3; window c + 2 becomes
          // image (c + 2) & 1 = c & 1.
          asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (c + 3 < nchunks) issue_A(c + 3, (c + 1) & 1);
          if (c + 2 < nchunks) convert(c + 2);
        } else {
          // VMEM retires in order: everything but the youngest 4 operations (the fragments of k-group g, fetched one
          // k-group ago) has landed, in particular this wave's pieces of window c+1 (>= 7 operations old: the pieces of
          // window c+2 and at least one group of fragments were issued behind them)
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (c + 3 < nchunks) issue_A(c + 3, (c + 3) % XSTAGES);       // nobody reads window c-1 any more
        }
      }
      load_w(c2, t2, kg2, wn_);
      load_x(c2, __builtin_amdgcn_readlane(v_taps, t2), kg2, xn);
    }
    c = c2; t = t2;
  };
  if constexpr (SHARED) {
    // Pinned schedule: the 12 fetches of k-group g + 1 (4 weight fragments from L2, 8 image fragments from LDS) are threaded one in
    // front of every pair of the 24 matrix instructions of k-group g (sched_barrier after each pair, as kernels_tdnn_v3.hip does):
    // left to itself hipcc puts a group's fetches and their waits in front of its matrix instructions, and with two waves per SIMD
    // both then sit in the same wait (measured: the unpinned SHARED form was 10 % SLOWER than the per-wave split, r3c).
    auto step = [&](const XFrags &xc, const WFrags &wc, XFrags &xn, WFrags &wnx, int g) {
      const int kg = g & 1;
      int c2 = c, t2 = t, kg2 = kg + 1;
      if (kg2 == 2) { kg2 = 0; t2 = t + 1; if (t2 == n_taps) { t2 = 0; c2 = c + 1; } }
      const bool more = g + 1 < G;
      if (!more) { c2 = c; t2 = t; kg2 = kg; }                     // the last k-group re-fetches itself (valid memory, never used)
      const bool enter = more && c2 != c;
      if (enter) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nchunks) convert(c + 2);
        // window c + 3 is issued BEHIND this step's weight fetches (below): the vector-memory counter retires in order, a fetch
        // issued behind the DMA could not be consumed before the window has landed
      }
      const size_t woff = ((size_t)t2 * nkg + (size_t)c2 * 2 + kg2) * 1024;
      const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t2);
      const int sw = (wrow >> 1) & 7;                               // the swizzle term is blind to + 32 rows: one address per half image
      const unsigned char *ib = lds + (2 + (c2 & 1)) * XSTAGE + wrow * XROWB;
      const unsigned char *ih = ib + (((kg2 * 2 + lh) ^ sw) << 4), *il = ib + (((4 + kg2 * 2 + lh) ^ sw) << 4);
#pragma unroll
      for (int pr = 0; pr < 12; ++pr) {
        if (pr == 4 && enter && c + 3 < nchunks) issue_A(c + 3, (c + 1) & 1);
        if (pr < 2) wnx.h[pr] = *reinterpret_cast<const uint4 *>(wh_base + pr * frag_stride + woff);
        else if (pr < 4) wnx.l[pr - 2] = *reinterpret_cast<const uint4 *>(wl_base + (pr - 2) * frag_stride + woff);
        else if (((pr - 4) & 1) == 0) xn.f[(pr - 4) >> 1].hi = *reinterpret_cast<const uint4 *>(ih + ((pr - 4) >> 1) * 32 * XROWB);
        else xn.f[(pr - 4) >> 1].lo = *reinterpret_cast<const uint4 *>(il + ((pr - 4) >> 1) * 32 * XROWB);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int idx = 2 * pr + e, term = idx >> 3, j = (idx >> 2) & 1, i = idx & 3;     // term-major: an accumulator recurs every 8th instruction
          const uint4 a = (term == 2) ? wc.l[j] : wc.h[j];
          const uint4 b = (term == 1) ? xc.f[i].lo : xc.f[i].hi;
          acc[i][j] = mfma16<ET>(a, b, acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      c = c2; t = t2;
    };
    for (int g = 0; g < G; g += 2) {
      step(x0, w0, x1, w1, g);
      if (g + 1 < G) step(x1, w1, x0, w0, g + 1);
    }
  } else {
    for (int g = 0; g < G; g += 2) {
      advance(g, x1, w1);
      mma(x0, w0);
      if (g + 1 < G) {
        advance(g + 1, x0, w0);
        mma(x1, w1);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // the ring becomes epilogue scratch
  x3_publish_range(range, p.status);
  asm volatile("" ::: "memory");

  // ---- epilogue: acc[i][j][r]: frame = m0 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  float *scr = reinterpret_cast<float *>(lds) + wn * (32 * XSPITCH);
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;           // 1 / the power of two the host multiplied the weights by (exact; 1 for the bf16 split)
  if constexpr (POOL) {
    // Fused statistics pooling (pooling.py:58-67 folded into the producing layer; the f32 form of kernels_tdnn_v3.hip's POOL
    // epilogue, same partial layout): the layer's output - 783 MB per C2 step for tdnn5 in f32, written and read back by the
    // pooling kernel - never reaches HBM.  Per 32-frame fragment the wave writes u = max(acc + b, lo) * scale (the output
    // minus the BN shift) to its scratch tile, then lane = channel sums the rows per utterance about a pivot (the utterance's
    // first value in this tile): sum (u - pv), sum (u - pv)^2, pv -> P[tile of 128 rows][segment slot][3][channel];
    // pool_finish_kernel merges a segment's tiles in row order (Chan et al.) and adds the shift to the mean.
    const int half = m0 >> 7;
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && m0 + k < p.rows) first_seg = p.row_seg[m0 + k];
    const int ch_l = n0 + wn * 64 + lane;
    const int rowseg_lo = p.row_seg[m0 + lane], rowseg_hi = p.row_seg[m0 + 64 + lane];
    float ps = 0.0f, pq = 0.0f, pv = 0.0f;
    int cur_seg = -1;
    auto flush = [&]() {
      const int slot = cur_seg - first_seg;
      if (slot >= 0 && slot < p.pool_slots && ch_l < p.ld_partial) {
        float *dst = p.pool_partial + ((size_t)(half * p.pool_slots + slot) * 3) * p.ld_partial + ch_l;
        dst[0] = ps;
        dst[p.ld_partial] = pq;
        dst[2 * p.ld_partial] = pv;
      }
    };
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const bool valid = (p.row_valid[(m0 + i * 32) >> 5] >> lr) & 1u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;
          const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
          const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
          float4 u;
          u.x = valid ? fmaxf(fmaf(acc[i][j][q * 4 + 0], unscale, b4.x), act_lo) * sc4.x : 0.0f;
          u.y = valid ? fmaxf(fmaf(acc[i][j][q * 4 + 1], unscale, b4.y), act_lo) * sc4.y : 0.0f;
          u.z = valid ? fmaxf(fmaf(acc[i][j][q * 4 + 2], unscale, b4.z), act_lo) * sc4.z : 0.0f;
          u.w = valid ? fmaxf(fmaf(acc[i][j][q * 4 + 3], unscale, b4.w), act_lo) * sc4.w : 0.0f;
          *reinterpret_cast<float4 *>(scr + lr * XSPITCH + j * 32 + 8 * q + 4 * lh) = u;
        }
      // rows are consumed in order, segments are contiguous in rows; rowseg_lo / hi hold row_seg of the tile's 128 rows
      const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
      const unsigned long long in_frag = 0xffffffffull << ((i & 1) * 32);
      const unsigned long long m_valid = __builtin_amdgcn_ballot_w64(rs_vec >= 0) & in_frag;
      if (m_valid == 0) continue;                                                  // gap rows only
      const int sg0 = __builtin_amdgcn_readlane(rs_vec, __builtin_ctzll(m_valid));
      const unsigned long long m_same = __builtin_amdgcn_ballot_w64(rs_vec == sg0) & in_frag;
      if (m_same == in_frag) {                                                     // one utterance, no gap row: branch-free
        if (sg0 != cur_seg) {
          if (cur_seg >= 0) flush();
          cur_seg = sg0; ps = 0.0f; pq = 0.0f; pv = scr[lane];
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const float d = scr[r * XSPITCH + lane] - pv;
          ps += d;
          pq = fmaf(d, d, pq);
        }
      } else {
#pragma unroll 1
        for (int r = 0; r < 32; ++r) {
          const int sg = __builtin_amdgcn_readlane(rs_vec, (i & 1) * 32 + r);          // wave-uniform
          if (sg < 0) continue;                                                          // gap row
          const float v = scr[r * XSPITCH + lane];
          if (sg != cur_seg) {
            if (cur_seg >= 0) flush();
            cur_seg = sg; ps = 0.0f; pq = 0.0f; pv = v;
          }
          const float d = v - pv;
          ps += d;
          pq = fmaf(d, d, pq);
        }
      }
    }
    if (cur_seg >= 0) flush();
    return;
  }
  float *yg = reinterpret_cast<float *>(p.y);
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const bool valid = (p.row_valid[(m0 + i * 32) >> 5] >> lr) & 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
        const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
        const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 512 + chl);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) {
            float z = fmaf(acc[i][j][q * 4 + e], unscale, b[e]);
            z = p.affine_first ? apply_act(z * sc[e] + sh[e], p.act1) : apply_act(z, p.act1) * sc[e] + sh[e];
            z = apply_act(z, p.act2);
            y[e] = valid ? z : 0.0f;
          } else {
            const float z = fmaxf(fmaf(acc[i][j][q * 4 + e], unscale, b[e]), act_lo) * sc[e] + sh[e];      // = tdnn_epilogue_fast for unscale = 1
            y[e] = valid ? z : 0.0f;
          }
        }
        *reinterpret_cast<float4 *>(scr + lr * XSPITCH + j * 32 + 8 * q + 4 * lh) = make_float4(y[0], y[1], y[2], y[3]);
      }
    // the scratch tile is wave-private: LDS operations of one wave complete in order
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int frow = it * 4 + (lane >> 4), slot = lane & 15;
      const float4 v = *reinterpret_cast<const float4 *>(scr + frow * XSPITCH + slot * 4);
      const int ch = n0 + wn * 64 + slot * 4;
      const int row = m0 + i * 32 + frow;
      if (ch < p.cout_store) *reinterpret_cast<float4 *>(yg + (size_t)row * p.ldy + ch) = v;
    }
  }
}

}  // namespace

// tile rows the launcher will use for this layer: 128 unless the batch is too small to fill the chip with 128-row tiles
// (ASV_AMD_X3_TILE = 64 | 128 overrides: A/B)
static int x3_tile_rows(const TdnnKernelParams &p) {
  static const int forced = getenv("ASV_AMD_X3_TILE") != nullptr ? atoi(getenv("ASV_AMD_X3_TILE")) : 0;
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  const int f = live ? (getenv("ASV_AMD_X3_TILE") != nullptr ? atoi(getenv("ASV_AMD_X3_TILE")) : 0) : forced;
  if (f == 64 || f == 128) return (f == 128 && p.rows % 128 != 0) ? 64 : f;
  if (p.x3_tile == 128 && p.rows % 128 == 0) return 128;
  const long long t128 = (long long)(p.rows / 128) * (round_up(p.cout_store, XBN) / XBN);
  return (p.rows % 128 == 0 && t128 >= 384) ? 128 : 64;
}

bool tdnn_x3_supported(const TdnnKernelParams &p) {
  const bool fits32 = (unsigned long long)p.rows * (unsigned long long)p.ldx * 4ull < (1ull << 40);
  return p.wfrag != nullptr && p.wlo != nullptr && p.x2 == nullptr && p.seg_bias == nullptr && p.seg_scale == nullptr && p.res == nullptr &&
         p.zero16 != nullptr && p.rows % 64 == 0 && p.cout_store % 4 == 0 && p.cout_store >= 192 && p.cin_pad >= 32 &&
         p.halo <= kHalo && p.ksplit <= 1 && fits32;
}

// the fused statistics pooling needs the 128-row geometry and the plain epilogue
bool tdnn_x3_pool_supported(const TdnnKernelParams &p) {
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  return tdnn_x3_supported(p) && fast && p.rows % 128 == 0 && (p.x3_terms & 7) == 7;
}

int launch_tdnn_x3(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows % 64 == 0, "tdnn(x3): rows %d not a multiple of 64", p.rows);
  ASV_REQUIRE(p.wfrag != nullptr && p.wlo != nullptr, "tdnn(x3): split fragment-packed weights missing");
  for (int t = 0; t < p.n_taps; ++t)
    ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(x3): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const bool pool = p.pool_partial != nullptr;
  const int terms = (p.x3_terms & 7) | 1;
  const int bm = pool ? 128 : (terms != 7 ? 64 : x3_tile_rows(p));        // the reduced-product variants: 64-row geometry only
  const int m_tiles = p.rows / bm, n_tiles = round_up(p.cout_store, XBN) / XBN;
  const dim3 grid(m_tiles * n_tiles), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  ASV_REQUIRE(p.x3_et == ET_BF16 || p.x3_et == ET_F16, "tdnn(x3): split type %d", p.x3_et);
  ASV_REQUIRE(p.w_unscale > 0.0f, "tdnn(x3): weight scale missing");
  ASV_REQUIRE(!pool || (tdnn_x3_pool_supported(p) && p.row_seg != nullptr && p.pool_slots >= 1), "tdnn(x3): fused pooling needs the plain epilogue, all three products, 128-row tiles and a row map");
  // ASV_AMD_X3_SHARED=0: the per-wave split inside the K loop on the 128-row geometry too (A/B; read once, or per launch with LIVE_TUNE)
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  static const int shared0 = getenv("ASV_AMD_X3_SHARED") != nullptr ? atoi(getenv("ASV_AMD_X3_SHARED")) : 1;
  const bool shared = (live ? (getenv("ASV_AMD_X3_SHARED") != nullptr ? atoi(getenv("ASV_AMD_X3_SHARED")) : 1) : shared0) != 0;
#define ASV_X3(GENV, ETV, TV, MFV, POOLV, SHV) hipLaunchKernelGGL((tdnn_gemm_x3_kernel<GENV, ETV, TV, MFV, POOLV, SHV>), grid, block, 0, s, p, m_tiles, n_tiles)
#define ASV_X3_128(GENV, ETV, POOLV) do { if (shared) ASV_X3(GENV, ETV, 7, 4, POOLV, true); else ASV_X3(GENV, ETV, 7, 4, POOLV, false); } while (0)
  // the reduced-product measurement variants exist for the plain epilogue and the 64-row geometry only (a layer with another
  // epilogue runs all three products)
#define ASV_X3_ET(ETV) do { if (pool) ASV_X3_128(false, ETV, true); \
                            else if (!fast) { if (bm == 128) ASV_X3_128(true, ETV, false); else ASV_X3(true, ETV, 7, 2, false, false); } \
                            else if (terms == 7) { if (bm == 128) ASV_X3_128(false, ETV, false); else ASV_X3(false, ETV, 7, 2, false, false); } \
                            else if (terms == 3) ASV_X3(false, ETV, 3, 2, false, false); \
                            else if (terms == 5) ASV_X3(false, ETV, 5, 2, false, false); else ASV_X3(false, ETV, 1, 2, false, false); } while (0)
  if (p.x3_et == ET_F16) ASV_X3_ET(ET_F16);
  else ASV_X3_ET(ET_BF16);
#undef ASV_X3_ET
#undef ASV_X3_128
#undef ASV_X3
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
