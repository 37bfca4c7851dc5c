// The 2-D ResNet trunk's convolutions in the f32x precision mode: f32 rows in HBM, every product as THREE 16-bit matrix
// instructions on hi / lo operand halves (w_hi x_hi + w_hi x_lo + w_lo x_hi, f32 accumulate) - the scheme of kernels_tdnn_x3.hip
// for the grid domain.  Replaces conv3x3 / conv1x1 + eval BatchNorm (+ ReLU) of libs/nnet/resnet.py:12-20, 23-110, 352-368 in
// the parity-grade mode, where until round 4 every grid-domain layer ran on the exact f32-input matrix instruction
// (157 TFLOP/s peak: ResNet34-SE at 1.7 k utterances/s against 18 k in bf16).
//
// A [B, C, F, T] map is a [rows][C] matrix, rows = (time, frequency) positions, frequency fastest (runtime.hip): a
// stride-1 3 x 3 convolution is nine row-offset taps dt * pitch + df, the stride-2 ones arrive as 4 backward taps over a
// space-to-depth tensor or as 1-tap layers over an im2col tensor.  One kernel template serves all of them:
//   * workgroup = BM output rows x BN output channels (BN = the layer's whole width up to 256), 4 waves as WM x WN, a wave owns
//     MF x NFW accumulator fragments of 32 x 32;
//   * K walks 32-channel chunks; per chunk the window (HLO + BM + HHI rows, all taps read it shifted) sits in LDS as ONE image
//     [row][hi: 32 halves | lo: 32 halves] = 128 bytes per row, 16-byte slots XOR-swizzled by (row >> 1) & 7 (conflict-free
//     ds_read_b128 for any tap shift).  The f32 rows come from HBM into registers, are split there (v_cvt_pk_f16_f32,
//     v_fma_mix_f32: x - hi exactly, v_cvt_pk_f16_f32 again) and written to the image once per workgroup: nine taps share one
//     split.  Two images: the loads of chunk c + 1 are issued at the top of chunk c's K loop - BEHIND the first weight prefetch,
//     the vector-memory counter retires in order - and converted behind it; one barrier per chunk;
//   * weights: hi / lo halves in fragment order [chunk][tap][k-group][n-fragment][hi | lo][lane][8], scaled by a power of two
//     per layer on the host (runtime.hip x3_weight_scale), 1 KiB wave loads from L2 one (tap, k-group) step ahead;
//   * a unit = one (NFW = 2) or two (NFW = 1) row fragments: 2 or 4 ds_read_b128 feed 6 matrix instructions; the next unit's
//     reads are issued in front of them;
//   * epilogue: acc / scale + bias -> [ReLU] -> folded BN, through a per-wave LDS tile to 16-byte stores of whole row pieces.
// Every output row is a fixed-order sum over its own window: bit-identical whatever batch the utterance is extracted in.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"
#include "host_convert.h"

namespace asv {
namespace {

constexpr int QROWB = 128;          // image row: 32 channels as [hi 64 B | lo 64 B]
constexpr int QCH = 32;             // channels per chunk

__device__ __forceinline__ int qswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

template <int WM_, int WN_, int MF_, int NFW_, int HLO_, int HHI_, int NBUF_>
struct QGeom {
  static constexpr int WM = WM_, WN = WN_, MF = MF_, NFW = NFW_, HLO = HLO_, HHI = HHI_, NBUF = NBUF_;
  static constexpr int BM = WM * MF * 32, BN = WN * NFW * 32;
  static constexpr int WIN = HLO + BM + HHI;
  static constexpr int IMG = WIN * QROWB;
  static constexpr int NP = (WIN * 4 + 255) / 256;          // 8-channel pieces per thread and chunk
  static constexpr int SPITCH = NFW * 32 + 4;               // floats per row of the epilogue's per-wave tile
  static constexpr int SCR = 4 * 32 * SPITCH * 4;
  static constexpr int MAIN = NBUF * IMG > SCR ? NBUF * IMG : SCR;
  static constexpr int LDS = MAIN + 3 * BN * 4;
  static constexpr int UI = NFW == 1 ? 2 : 1;               // row fragments per unit (6 matrix instructions either way)
  static constexpr int NU = MF / UI;                        // units per (tap, k-group) step
  // Halo-free geometries serve 1-tap layers: a chunk's K loop is 2 steps (~0.4 - 0.8 us) - shorter than the round trip of the next
  // chunk's rows - so they keep TWO register stages and fetch chunk c + 2 at the top of chunk c (measured with one stage: the
  // 1536 -> 128 layer of ECAPA's attention at 284 us, the 1152 -> 256 im2col GEMM of the ResNet at 166 us)
  static constexpr bool PF2 = HLO == 0 && HHI == 0 && NBUF == 2;
  static_assert(WM * WN == 4 && MF % UI == 0 && 2 * LDS <= 163840, "grid conv (f32x) geometry");
};

// MODE 0: one input, the plain epilogue (bias -> [ReLU] -> folded BN); 1: + a second input added while the rows are staged (Res2Net's
// sp + x_i), plain epilogue; 2: the long epilogue of device_utils.h tdnn_epilogue (tanh / sigmoid / "bn-relu" order / per-segment bias
// and scale / residual), with or without a second input - kept ROLLED (one copy of its code per kernel, applied to the rows on their
// way out of the per-wave tile): unrolled over the 64 accumulator pieces it was measured 2.3 x slower than the whole K loop
template <typename G, int ET, int MODE = 0>
__global__ __launch_bounds__(256, 2) void grid_conv_x3_kernel(const TdnnKernelParams p, const int n_tiles, const int nft) {
  constexpr bool GENERIC = MODE != 0;         // (registers for a second input)
  constexpr bool LONG_EPI = MODE == 2;
  constexpr int WN = G::WN, MF = G::MF, NFW = G::NFW, HLO = G::HLO, BM = G::BM, BN = G::BN, WIN = G::WIN, NP = G::NP, UI = G::UI, NU = G::NU;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
  const int n_taps = p.n_taps;
  const int nchunks = (p.cin_pad + QCH - 1) / QCH;

  // the epilogue's per-channel constants (bias | scale | shift of this tile's BN channels) -> LDS now, read behind the K loops
  float *lds_par = reinterpret_cast<float *>(lds + G::MAIN);
  for (int e = tid; e < 3 * BN; e += 256) {
    const int which = e / BN, c = e % BN;
    const float *src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
    lds_par[e] = src != nullptr ? src[n0 + c] : (which == 1 ? 1.0f : 0.0f);
  }

  // ---- staging: f32 rows -> registers -> [hi | lo] image
  const float *xg = reinterpret_cast<const float *>(p.x);
  const float *x2g = reinterpret_cast<const float *>(p.x2);       // optional second input, added while the rows are staged (Res2Net's sp + x_i)
  // The loads are issued UNCONDITIONALLY (rows / channels outside the matrix: a valid address, the value zeroed at conversion, `ok`
  // bit per piece): behind per-lane branches hipcc cannot count them and waits with vmcnt(0) in front of the conversion - for the
  // loads issued a chunk ago AND the ones just issued (the two-stage prefetch of the 1-tap geometries then hides nothing).
  struct Stage { uint4 a[NP], b[NP], a2[GENERIC ? NP : 1], b2[GENERIC ? NP : 1]; uint32_t ok; };
  uint32_t range = 0u;                                           // range watch of the half split (device_utils.h)
  auto gload = [&](int c, Stage &st) {
    uint32_t okbits = 0u;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int item = it * 256 + tid, w = item >> 2, q = item & 3;
      const int row = m0 - HLO + w, ch = c * QCH + q * 8;
      const bool ok = item < WIN * 4 && row >= 0 && row < p.rows && ch < p.cin_pad;
      okbits |= ok ? (1u << it) : 0u;
      const size_t off = ok ? (size_t)row * p.ldx + ch : 0;
      const float *src = xg + off;
      st.a[it] = *reinterpret_cast<const uint4 *>(src);
      st.b[it] = *reinterpret_cast<const uint4 *>(src + 4);
      if constexpr (GENERIC) {
        // the second input stays in registers of its own until the rows are converted: adding here would wait for both loads
        // in front of the K loop they are meant to hide behind (measured: 95 instead of 43 us for a Res2Net branch)
        if (x2g != nullptr) {                                     // (wave-uniform)
          const float *src2 = x2g + (ok ? (size_t)row * p.ldx2 + ch : 0);
          st.a2[it] = *reinterpret_cast<const uint4 *>(src2);
          st.b2[it] = *reinterpret_cast<const uint4 *>(src2 + 4);
        }
      }
    }
    st.ok = okbits;
  };
  auto sstore = [&](int buf, Stage &st) {
    unsigned char *img = lds + buf * G::IMG;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int item = it * 256 + tid, w = item >> 2, q = item & 3;
      if (item < WIN * 4) {
        if constexpr (GENERIC) {
          if (x2g != nullptr) { st.a[it] = add_f32x4(st.a[it], st.a2[it]); st.b[it] = add_f32x4(st.b[it], st.b2[it]); }
        }
        const bool ok = (st.ok >> it) & 1u;
        const uint4 zero = make_uint4(0, 0, 0, 0);
        const X3Frag f = x3_split<ET, true>(ok ? st.a[it] : zero, ok ? st.b[it] : zero, range);
        *reinterpret_cast<uint4 *>(img + w * QROWB + qswz(w, q) * 16) = f.hi;
        *reinterpret_cast<uint4 *>(img + w * QROWB + qswz(w, 4 + q) * 16) = f.lo;
      }
    }
  };

  // ---- operands
  struct WF { uint4 h[NFW], l[NFW]; };
  struct XF { uint4 h[UI], l[UI]; };
  const unsigned char *wq = reinterpret_cast<const unsigned char *>(p.wconv) + (size_t)(n0 / 32 + wn * NFW) * 2048 + (size_t)lane * 16;
  const size_t step_stride = (size_t)nft * 2048;               // bytes per (chunk, tap, k-group) step: nft fragments x (hi, lo)
  auto load_w = [&](int step, WF &w) {                         // step = (c * n_taps + t) * 2 + kg
    const unsigned char *b = wq + (size_t)step * step_stride;
#pragma unroll
    for (int j = 0; j < NFW; ++j) {
      w.h[j] = *reinterpret_cast<const uint4 *>(b + j * 2048);
      w.l[j] = *reinterpret_cast<const uint4 *>(b + j * 2048 + 1024);
    }
  };
  const int wrow0 = HLO + wm * (MF * 32) + lr;                 // window row of this lane in row fragment 0 at tap offset 0
  auto load_x = [&](const unsigned char *img, int d, int kg, int u, XF &x) {
#pragma unroll
    for (int k = 0; k < UI; ++k) {
      const int w = wrow0 + (u * UI + k) * 32 + d;
      x.h[k] = *reinterpret_cast<const uint4 *>(img + w * QROWB + qswz(w, kg * 2 + lh) * 16);
      x.l[k] = *reinterpret_cast<const uint4 *>(img + w * QROWB + qswz(w, 4 + kg * 2 + lh) * 16);
    }
  };

  f32x16_t acc[MF][NFW];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NFW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto mma = [&](const WF &w, const XF &x, int u) {
    // product-major: an accumulator recurs every UI * NFW = 2 instructions
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int k = 0; k < UI; ++k)
#pragma unroll
        for (int j = 0; j < NFW; ++j) {
          const uint4 a = term == 2 ? w.l[j] : w.h[j];
          const uint4 b = term == 1 ? x.l[k] : x.h[k];
          acc[u * UI + k][j] = mfma16<ET>(a, b, acc[u * UI + k][j]);
        }
  };

  // ---- prologue: window of chunk 0 -> image 0, weights of step 0
  const int v_taps = p.taps[lane < ASV_MAX_TAPS ? lane : 0];
  WF wa, wb;
  Stage s0;
  Stage s1;                                                        // (PF2 only; dead otherwise)
  gload(0, s0);
  load_w(0, wa);
  sstore(0, s0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (G::PF2) {
    if (nchunks > 1) gload(1, s1);
  }

  // one chunk: its K loop on image c & 1; the rows of chunk `c_ld` go into flight (-> `ld`) behind the first weight prefetch, the
  // rows in `st` (chunk c + 1, landed by now) are converted into the other image behind the loop
  auto chunk = [&](const int c, Stage &ld, const int c_ld, Stage &st) {
    const unsigned char *img = lds + (G::NBUF == 2 ? (c & 1) : 0) * G::IMG;
    const bool more_chunks = c + 1 < nchunks;
    XF xa, xb;
    load_x(img, __builtin_amdgcn_readlane(v_taps, 0), 0, 0, xa);
    // k-group 1's fragments of the first tap -> wb, then the next rows go into flight BEHIND them (the vector-memory counter retires
    // in order: a fragment fetch issued behind the rows could not be consumed before they land).  Both sit in front of the tap loop,
    // which runs at least once (do-while): hipcc can then count the fragment fetches that are younger than the rows and waits for
    // the rows alone in front of their conversion (with a for loop it had to assume zero trips: vmcnt(0), every chunk).
    int step = c * n_taps * 2;
    load_w(step + 1, wb);
    if constexpr (G::NBUF == 2) {
      if (c_ld < nchunks) gload(c_ld, ld);
    }
    __builtin_amdgcn_sched_barrier(0);
    // one tap: k-group 0 on wa, k-group 1 on wb, the fragments of the next two k-groups fetched one k-group ahead
    auto tap = [&](const int t) {
      const int d = __builtin_amdgcn_readlane(v_taps, t);
      const bool last_tap = t + 1 == n_taps;
      const int dn = __builtin_amdgcn_readlane(v_taps, last_tap ? t : t + 1);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        XF &xc = (u & 1) ? xb : xa;
        XF &xn = (u & 1) ? xa : xb;
        if (u + 1 < NU) load_x(img, d, 0, u + 1, xn);
        else load_x(img, d, 1, 0, xn);
        mma(wa, xc, u);
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next step's fragments (next tap, or the next chunk's first step; the very last step re-fetches itself: valid memory,
      // never used) -> wa
      {
        const int next = (last_tap && !more_chunks) ? step + 1 : step + 2;
        load_w(next, wa);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        // units alternate the two fragment sets; NU units per k-group: the set in use at the start of k-group 1 is (NU & 1)
        XF &xc = ((NU + u) & 1) ? xb : xa;
        XF &xn = ((NU + u) & 1) ? xa : xb;
        if (u + 1 < NU) load_x(img, d, 1, u + 1, xn);
        else if (!last_tap) load_x(img, dn, 0, 0, xn);          // (2 NU units per tap: the next tap starts on xa again)
        mma(wb, xc, u);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!last_tap) load_w(step + 3, wb);                       // k-group 1 of the next tap
      __builtin_amdgcn_sched_barrier(0);
      step += 2;
    };
    // The FIRST tap stands outside the loop: its waits for wa / wb are then counted against what this chunk has issued so far -
    // inside the loop hipcc merges them with the back edge's (only 4 younger fetches) and the chunk would start by waiting for the
    // rows it has just requested.  (Halo-free geometries have exactly one tap: taps ascend strictly and |tap| <= 0.)
    tap(0);
    if constexpr (!G::PF2) {
#pragma unroll 1
      for (int t = 1; t < n_taps; ++t) tap(t);
    }
    if constexpr (G::NBUF == 2) {
      if (more_chunks) {
        sstore((c + 1) & 1, st);                                 // nobody reads that image: it was chunk c - 1's
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
  };
  if constexpr (G::PF2) {
    // even chunks: chunk c + 2 -> s0 (chunk c left it at the end of chunk c - 1), chunk c + 1 waits in s1; odd chunks: the other way
#pragma unroll 1
    for (int c = 0; c < nchunks; c += 2) {
      chunk(c, s0, c + 2, s1);
      if (c + 1 < nchunks) chunk(c + 1, s1, c + 3, s0);
    }
  } else {
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) chunk(c, s0, c + 1, s0);
  }

  // ---- epilogue: acc[i][j][r] = row m0 + wm*MF*32 + i*32 + lr, channel n0 + (wn*NFW + j)*32 + 8*(r>>2) + 4*lh + (r&3)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                   // every wave is through with the images: they become scratch
  x3_publish_range(range, p.status);
  asm volatile("" ::: "memory");
  float *scr = reinterpret_cast<float *>(lds) + wave * (32 * G::SPITCH);
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;
  float *yg = reinterpret_cast<float *>(p.y);
#pragma unroll
  for (int i = 0; i < MF; ++i) {
    const int rbase = m0 + wm * (MF * 32) + i * 32;
    const uint32_t vbits = p.row_valid[rbase >> 5];
    const bool valid = (vbits >> lr) & 1u;
#pragma unroll
    for (int j = 0; j < NFW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chl = (wn * NFW + j) * 32 + 8 * q + 4 * lh;
        float y[4];
        if constexpr (LONG_EPI) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = acc[i][j][q * 4 + e] * unscale;          // the epilogue itself: on the way out, below
        } else {
          const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
          const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + BN + chl);
          const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 2 * BN + chl);
          const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = fmaxf(fmaf(acc[i][j][q * 4 + e], unscale, b[e]), act_lo) * sc[e] + sh[e];      // = tdnn_epilogue_fast for unscale = 1
            y[e] = valid ? z : 0.0f;
          }
        }
        *reinterpret_cast<float4 *>(scr + lr * G::SPITCH + j * 32 + 8 * q + 4 * lh) = make_float4(y[0], y[1], y[2], y[3]);
      }
    // the tile is wave-private: LDS operations of one wave complete in order
    constexpr int SLOTS = NFW * 8;                                // float4 per row of the tile
    constexpr int RPI = 64 / SLOTS;                               // rows per store instruction: 8 | 4
    if constexpr (LONG_EPI) {
#pragma unroll 1
      for (int it = 0; it < 32 / RPI; ++it) {
        const int frow = it * RPI + lane / SLOTS, slot = lane % SLOTS;
        const float4 v = *reinterpret_cast<const float4 *>(scr + frow * G::SPITCH + slot * 4);
        const int chl = wn * (NFW * 32) + slot * 4;
        const bool ok = (vbits >> frow) & 1u;
        const float a[4] = {v.x, v.y, v.z, v.w};
        float y[4];
#pragma unroll 1
        for (int e = 0; e < 4; ++e)
          y[e] = tdnn_epilogue<ET_F32>(p, a[e], rbase + frow, n0 + chl + e, lds_par[chl + e], lds_par[BN + chl + e], lds_par[2 * BN + chl + e], ok);
        *reinterpret_cast<float4 *>(yg + (size_t)(rbase + frow) * p.ldy + n0 + chl) = make_float4(y[0], y[1], y[2], y[3]);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int frow = it * RPI + lane / SLOTS, slot = lane % SLOTS;
        const float4 v = *reinterpret_cast<const float4 *>(scr + frow * G::SPITCH + slot * 4);
        *reinterpret_cast<float4 *>(yg + (size_t)(rbase + frow) * p.ldy + n0 + wn * (NFW * 32) + slot * 4) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The 32 -> 32 convolutions of the trunk's first stage (6 of 32 layers, the widest maps: 17 % of a ResNet34-SE f32x step) as a
// PERSISTENT kernel with a sliding window - the f32x form of kernels_conv2d.hip grid_conv_narrow_pers_kernel.  The one-tile kernel
// above fetches and splits 420 window rows for 256 output rows (halo = pitch + 1 = 82 rows on either side: 1.64 x the reads and the
// conversion work), one phase after the other - SQ counters (profiles/r4_mfma_util_resnet_f32x.json): matrix pipe 0.29 busy, waves
// parked at a wait 0.44 of their time.  Here a workgroup walks a contiguous run of 128-row tiles:
//   * the window lives in an LDS ring of [hi | lo] image rows addressed by the matrix row modulo the ring; a tile adds its 128 new
//     rows only (read and split once: amplification 1.0);
//   * those rows are fetched into registers TWO tiles ahead (top of the tile's K loop), split and written to the ring one tile
//     ahead, behind the tile's stores - into ring rows nobody reads any more (ring = window + one tile, 448 rows): one barrier per tile;
//   * a wave = 32 rows x the 32 output channels with ALL weight fragments (9 taps x 2 k-groups x hi / lo = 36 x 16 bytes per lane)
//     in registers for the whole run: the K loop is LDS reads and matrix instructions only;
//   * the same (tap, k-group, term) accumulation order as the one-tile kernel: bit-identical outputs (tests/test_gpu_grid_conv_x3.py).
struct QPers32 {
  static constexpr int BM = 128, HALO = 84;                 // >= pitch + 1 of an 80-bin grid (82), a multiple of 4
  static constexpr int WIN = BM + 2 * HALO;                 // 296
  static constexpr int RING = 448;                          // >= WIN + BM = 424, a multiple of 16 (the slot swizzle continues across the wrap)
  static constexpr int SPITCH = 36;
  static constexpr int SCR_OFF = RING * QROWB;              // 57344
  static constexpr int PAR_OFF = SCR_OFF + 4 * 32 * SPITCH * 4;
  static constexpr int LDS = PAR_OFF + 3 * 32 * 4;          // 76160: two workgroups per CU
  static_assert(RING >= WIN + BM && RING % 16 == 0 && 2 * LDS <= 163840, "sliding-window geometry (f32x, 32 channels)");
};

template <int ET>
__global__ __launch_bounds__(256, 2) void grid_conv_x3_pers32_kernel(const TdnnKernelParams p, const int tiles_per_wg) {
  using G = QPers32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int n_tiles = p.rows / G::BM;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(t_begin + tiles_per_wg, n_tiles);
  if (t_begin >= t_end) return;

  float *lds_par = reinterpret_cast<float *>(lds + G::PAR_OFF);
  if (tid < 96) {
    const int which = tid / 32, c = tid % 32;
    const float *src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
    lds_par[tid] = src != nullptr ? src[c] : (which == 1 ? 1.0f : 0.0f);
  }

  // ---- rows -> registers -> ring.  A stage = the 128 new rows of one tile: 2 pieces (row, 8 channels) per thread
  const float *xg = reinterpret_cast<const float *>(p.x);
  struct Stage { uint4 a[2], b[2]; uint32_t ok; };
  uint32_t range = 0u;
  const int q = tid & 3;                                      // this thread's 8-channel group, in every piece
  auto gload = [&](int row0, Stage &st) {                     // matrix rows row0 .. row0 + 127 (unconditional loads, see the kernel above)
    uint32_t okbits = 0u;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = row0 + it * 64 + (tid >> 2);
      const bool ok = row >= 0 && row < p.rows;
      okbits |= ok ? (1u << it) : 0u;
      const float *src = xg + (ok ? (size_t)row * p.ldx + q * 8 : 0);
      st.a[it] = *reinterpret_cast<const uint4 *>(src);
      st.b[it] = *reinterpret_cast<const uint4 *>(src + 4);
    }
    st.ok = okbits;
  };
  auto sstore = [&](int row0, const Stage &st) {              // (row0 + HALO >= 0: virtual rows)
    const int v0 = (row0 + G::HALO) % G::RING;                // wave-uniform
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int rr = v0 + it * 64 + (tid >> 2);
      rr = rr >= G::RING ? rr - G::RING : rr;
      const bool ok = (st.ok >> it) & 1u;
      const uint4 zero = make_uint4(0, 0, 0, 0);
      const X3Frag f = x3_split<ET, true>(ok ? st.a[it] : zero, ok ? st.b[it] : zero, range);
      *reinterpret_cast<uint4 *>(lds + rr * QROWB + qswz(rr, q) * 16) = f.hi;
      *reinterpret_cast<uint4 *>(lds + rr * QROWB + qswz(rr, 4 + q) * 16) = f.lo;
    }
  };

  // ---- weights: every fragment of the layer, for the whole run
  uint4 wh[9][2], wl[9][2];
  {
    const unsigned char *wq = reinterpret_cast<const unsigned char *>(p.wconv) + (size_t)lane * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        wh[t][kg] = *reinterpret_cast<const uint4 *>(wq + (size_t)(t * 2 + kg) * 2048);
        wl[t][kg] = *reinterpret_cast<const uint4 *>(wq + (size_t)(t * 2 + kg) * 2048 + 1024);
      }
  }
  const int v_taps = p.taps[lane < 9 ? lane : 0];

  // ---- the first window (rows m - HALO .. m + BM + HALO of the first tile), then the next tile's rows into registers
  {
    const int first = t_begin * G::BM - G::HALO;
    Stage st;
#pragma unroll 1
    for (int r0 = 0; r0 < G::WIN; r0 += 128) {              // 3 x 128 rows (the last 88 beyond the window belong to the next tile: rewritten below, harmless)
      gload(first + r0, st);
      sstore(first + r0, st);
    }
  }
  Stage sa, sb;                                               // sa: tile + 1's rows (landed), sb: tile + 2's rows (in flight)
  gload((t_begin + 1) * G::BM + G::HALO, sa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;
  float *yg = reinterpret_cast<float *>(p.y);
  float *scr = reinterpret_cast<float *>(lds + G::SCR_OFF) + wave * (32 * G::SPITCH);

  // Every vector-memory operation of the loop is UNCONDITIONAL (past the end of the run: rows that are fetched / split and never
  // read) and the validity bits travel one tile ahead: hipcc then counts what is younger than the value it needs and waits for
  // that value alone - behind a branch, or consumed in the tile that fetched them, each wait became vmcnt(0): the row fetch of the
  // next tiles and the previous tile's stores drained in front of every K loop.
  const int last_word = (p.rows >> 5) - 1;
  auto one_tile = [&](const int tile, Stage &next, Stage &next2, const uint32_t vbits, uint32_t &vbits_next) {
    const int m0 = tile * G::BM;
    vbits_next = p.row_valid[min((m0 + G::BM + wave * 32) >> 5, last_word)];
    // the rows tile + 2 adds -> next2 (nobody holds it: its previous content went to the ring at the end of the previous tile)
    gload((tile + 2) * G::BM + G::HALO, next2);
    const int wb = __builtin_amdgcn_readfirstlane((m0 + G::HALO + wave * 32) % G::RING);      // ring row of this wave's first output row
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // the operands of step (t, kg) + 1 are read in front of step (t, kg)'s three matrix instructions (two waves per SIMD: without
    // this the ~100+ cycles of every ds_read round trip stand between two 96-cycle instruction groups of a wave)
    uint4 xh[2], xl[2];
    auto read_x = [&](const int t, const int kg, uint4 &h, uint4 &l) {
      const int d = __builtin_amdgcn_readlane(v_taps, t);
      int rr = wb + lr + d;
      rr = rr < 0 ? rr + G::RING : (rr >= G::RING ? rr - G::RING : rr);
      const unsigned char *rowp = lds + rr * QROWB;
      h = *reinterpret_cast<const uint4 *>(rowp + qswz(rr, kg * 2 + lh) * 16);
      l = *reinterpret_cast<const uint4 *>(rowp + qswz(rr, 4 + kg * 2 + lh) * 16);
    };
    read_x(0, 0, xh[0], xl[0]);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int t = st >> 1, kg = st & 1;
      if (st + 1 < 18) read_x((st + 1) >> 1, (st + 1) & 1, xh[(st + 1) & 1], xl[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                       // (hipcc otherwise sinks the reads behind the step's second instruction)
      acc = mfma16<ET>(wh[t][kg], xh[st & 1], acc);
      acc = mfma16<ET>(wh[t][kg], xl[st & 1], acc);
      acc = mfma16<ET>(wl[t][kg], xh[st & 1], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: acc[r] = row m0 + wave*32 + lr, channel 8*(r>>2) + 4*lh + (r&3)
    const int rbase = m0 + wave * 32;
    const bool valid = (vbits >> lr) & 1u;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int chl = 8 * qq + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
      const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 32 + chl);
      const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 64 + chl);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = fmaxf(fmaf(acc[qq * 4 + e], unscale, b[e]), act_lo) * sc[e] + sh[e];
        y[e] = valid ? z : 0.0f;
      }
      *reinterpret_cast<float4 *>(scr + lr * G::SPITCH + chl) = make_float4(y[0], y[1], y[2], y[3]);
    }
    // the tile is wave-private: the LDS operations of one wave complete in order
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int frow = it * 8 + (lane >> 3), slot = lane & 7;
      const float4 v = *reinterpret_cast<const float4 *>(scr + frow * G::SPITCH + slot * 4);
      *reinterpret_cast<float4 *>(yg + (size_t)(rbase + frow) * p.ldy + slot * 4) = v;
    }
    // tile + 1's rows: split and into the ring rows behind the window (nobody reads them during this tile)
    sstore((tile + 1) * G::BM + G::HALO, next);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  uint32_t va = p.row_valid[min((t_begin * G::BM + wave * 32) >> 5, last_word)], vb = 0u;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; tile += 2) {
    one_tile(tile, sa, sb, va, vb);
    if (tile + 1 < t_end) one_tile(tile + 1, sb, sa, vb, va);
  }
  x3_publish_range(range, p.status);
}

// ---------------------------------------------------------------------------------------------------------------
// The 64 -> 64 convolutions of the second stage (7 layers; one-tile kernel: 282 us against ~100 us of matrix work or HBM time - every
// 128-row tile streams the layer's 147 KiB of hi / lo weight fragments from L2 once per wave column, re-fetches 96 halo rows and
// converts 1.75 x its rows) in the same persistent form.  The weights of a 64-channel layer do not fit one wave's registers, so K is
// split over TWO waves: workgroup = 512 threads = 8 waves = 2 row fragments x 2 output fragments x 2 INPUT halves; a wave keeps the
// 36 fragments of (its 32 output channels, its 32 input channels, 9 taps) for the whole run and reads only its half of the ring rows;
// the two partial accumulators meet in LDS once per tile - each wave of the pair finishes HALF of the 32 channels (partner's part +
// its own: one f32 addition, the same whichever wave performs it) and runs that half's epilogue and stores.  64-row tiles, ring of 224
// rows x 256 B ([chunk][hi | lo], 16-byte slots XOR-swizzled by row & 15), one workgroup per CU.
// Measured (profiles/r4m_x3_pers64_ab.txt, r4n_pers64_abl.txt, r4o_pers_skew.txt): 282 -> 214 us per layer.  Ablations of the developer
// build: 230 us as is on that box, 121 without the K loop, 200 without exchange + epilogue, 220 without row fetch / split, 70 without
// K loop and fetch, 33 with nothing but the loop skeleton - the K loop costs its full 109 us (= the matrix pipe's time for the layer at
// the clock it runs at) ON TOP of the other phases: the two waves a SIMD holds run the same phase at the same time.  Tried without
// effect on that sum: 64-row tiles / one 8-wave workgroup per CU (=5), LDS operands one step ahead, two accumulators per wave (kept:
// they cost nothing), starting the odd workgroup of a CU one K loop late; and (template parameter PIPE, the default) tile i's exchange read + epilogue + stores and the row split placed BETWEEN the steps of tile i + 1's K loop: 220 -> 217 us (profiles/r4p_pers64_pipe_ab.txt) - correct, kept, not the answer either.
// Working explanation (instruction count of the loop body, tools/r5_pers64_sq.sh is the check): a wave issues ~660 instructions per 32-row tile - 54 matrix (one dependent chain), ~300 VALU (ring / swizzle addresses of 36 reads, split, epilogue), ~80 SALU, 30 s_nop - IN ORDER, two waves per SIMD: ~11 cycles each = the 7.2 k cycles per tile measured; the pipe is idle half the time because the wave is busy elsewhere, not because it waits for data.
// Not the accumulation order of the one-tile kernel (there: chunk 0's taps, then chunk 1's, in one accumulator; here four partial
// sums - (chunk, k-group) - are added at the end): equal to f32 rounding (~1e-7), not bit for bit; every row still has ONE fixed
// order whatever the batch.
// WM = row fragments per tile: 1 -> 32-row tiles, 4 waves, 67 KiB: TWO workgroups per CU (the default: one workgroup's barrier,
// exchange and epilogue phases run under the other's K loop); 2 -> 64-row tiles, 8 waves, 109 KiB: one workgroup per CU (measured
// first: 282 -> 214 us per layer).
template <int WM_> struct QPers64 {
  static constexpr int WM = WM_, NT = WM * 256, PAIRS = WM * 2;
  static constexpr int BM = WM * 32, HALO = 44;             // >= pitch + 1 of a 40-bin grid (42), a multiple of 4
  static constexpr int WIN = BM + 2 * HALO;                 // 120 | 152
  static constexpr int RING = WM == 1 ? 160 : 224;          // >= WIN + BM = 152 | 216, a multiple of 16
  static constexpr int ROWB = 256;
  static constexpr int RED_OFF = RING * ROWB;               // partial accumulators, [parity][pair][half][2][lane] float4
  static constexpr int SCR_OFF = RED_OFF + 2 * PAIRS * 4096;    // per wave: 32 rows x 16 channels (+4) on their way to 16-byte stores
  static constexpr int SPITCH = 20;
  static constexpr int PAR_OFF = SCR_OFF + 4 * WM * 32 * SPITCH * 4;
  static constexpr int LDS = PAR_OFF + 3 * 64 * 4;          // 68352 | 111360
  static_assert(RING >= WIN + BM && RING % 16 == 0 && (3 - WM) * LDS <= 163840, "sliding-window geometry (f32x, 64 channels)");
};

template <int ET, int WM, bool PIPE = false>
__global__ __launch_bounds__(WM * 256, 3 - WM) void grid_conv_x3_pers64_kernel(const TdnnKernelParams p, const int tiles_per_wg) {
  using G = QPers64<WM>;
#ifdef ASV_WITH_ABLATION
  const int abl = p.tune;           // developer build only (results are garbage): 1 no output stores, 2 no row fetch / split, 4 no K loop, 8 no exchange + epilogue
#else
  constexpr int abl = 0;
#endif
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM == 1 ? 0 : (wave >> 2), wn = (wave >> 1) & 1, kh = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int n_tiles = p.rows / G::BM;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(t_begin + tiles_per_wg, n_tiles);
  if (t_begin >= t_end) return;

  float *lds_par = reinterpret_cast<float *>(lds + G::PAR_OFF);
  if (tid < 192) {
    const int which = tid / 64, c = tid % 64;
    const float *src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
    lds_par[tid] = src != nullptr ? src[c] : (which == 1 ? 1.0f : 0.0f);
  }

  // ---- rows -> registers -> ring.  A stage = the BM new rows of one tile: ONE piece (row, 8 channels) per thread
  const float *xg = reinterpret_cast<const float *>(p.x);
  struct Stage { uint4 a, b; uint32_t ok; };
  uint32_t range = 0u;
  const int q8 = tid & 7, srow = tid >> 3;                    // this thread's 8-channel group (chunk q8 >> 2, piece q8 & 3) and row of a stage
  auto gload = [&](int row0, Stage &st) {
    const int row = row0 + srow;
    const bool ok = row >= 0 && row < p.rows;
    const float *src = xg + (ok ? (size_t)row * p.ldx + q8 * 8 : 0);
    st.a = *reinterpret_cast<const uint4 *>(src);
    st.b = *reinterpret_cast<const uint4 *>(src + 4);
    st.ok = ok ? 1u : 0u;
  };
  auto sstore = [&](int row0, const Stage &st) {
    int rr = (row0 + G::HALO) % G::RING + srow;
    rr = rr >= G::RING ? rr - G::RING : rr;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    const X3Frag f = x3_split<ET, true>(st.ok ? st.a : zero, st.ok ? st.b : zero, range);
    const int s0 = (q8 >> 2) * 8 + (q8 & 3);
    *reinterpret_cast<uint4 *>(lds + rr * G::ROWB + ((s0 ^ (rr & 15)) << 4)) = f.hi;
    *reinterpret_cast<uint4 *>(lds + rr * G::ROWB + (((s0 + 4) ^ (rr & 15)) << 4)) = f.lo;
  };

  // ---- weights: the fragments of (output fragment wn, input chunk kh), for the whole run
  uint4 wh[9][2], wl[9][2];
  {
    const unsigned char *wq = reinterpret_cast<const unsigned char *>(p.wconv) + (size_t)lane * 16;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const size_t frag = ((((size_t)kh * 9 + t) * 2 + kg) * 2 + wn) * 2;
        wh[t][kg] = *reinterpret_cast<const uint4 *>(wq + frag * 1024);
        wl[t][kg] = *reinterpret_cast<const uint4 *>(wq + frag * 1024 + 1024);
      }
  }
  const int v_taps = p.taps[lane < 9 ? lane : 0];

  {
    const int first = t_begin * G::BM - G::HALO;
    Stage st;
#pragma unroll 1
    for (int r0 = 0; r0 < G::WIN; r0 += G::BM) {             // whole stages (the rows past the window belong to the next tile: rewritten below, harmless)
      gload(first + r0, st);
      sstore(first + r0, st);
    }
  }
  Stage sa, sb;
  gload((t_begin + 1) * G::BM + G::HALO, sa);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;
  float *yg = reinterpret_cast<float *>(p.y);
  float *scr = reinterpret_cast<float *>(lds + G::SCR_OFF) + wave * (32 * G::SPITCH);
  const int last_word = (p.rows >> 5) - 1;
  const int cbase = wn * 32 + kh * 16;                        // the 16 output channels this wave finishes

  auto one_tile = [&](const int tile, Stage &next, Stage &next2, const uint32_t vbits, uint32_t &vbits_next) {
    const int m0 = tile * G::BM;
    vbits_next = p.row_valid[min((m0 + G::BM + wm * 32) >> 5, last_word)];
    if (!(abl & 2)) gload((tile + 2) * G::BM + G::HALO, next2);
    const int wb = __builtin_amdgcn_readfirstlane((m0 + G::HALO + wm * 32) % G::RING);
    // TWO accumulators, k-group 0's steps in one and k-group 1's in the other: a step's three matrix instructions stand back to
    // back on one accumulator (nothing between them: the lo half is read FIRST, so the wait in front of the first instruction
    // covers both operands), the reads of the next step follow, and the next step's instructions use the OTHER accumulator.  With one
    // accumulator every ds_read / s_waitcnt between two dependent matrix instructions cost the ~43-cycle same-accumulator penalty
    // (MI355X_MICROARCH.md, instruction table): the pipe sat at 0.42 of its peak whatever the tile shape or the read-ahead.
    f32x16_t acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acc2[r] = 0.0f; }
    uint4 xh[2], xl[2];
    auto read_x = [&](const int t, const int kg, uint4 &h, uint4 &l) {
      const int d = __builtin_amdgcn_readlane(v_taps, t);
      int rr = wb + lr + d;
      rr = rr < 0 ? rr + G::RING : (rr >= G::RING ? rr - G::RING : rr);
      const unsigned char *rowp = lds + rr * G::ROWB;
      const int sw = rr & 15;
      l = *reinterpret_cast<const uint4 *>(rowp + (((kh * 8 + 4 + kg * 2 + lh) ^ sw) << 4));
      h = *reinterpret_cast<const uint4 *>(rowp + (((kh * 8 + kg * 2 + lh) ^ sw) << 4));
    };
    read_x(0, 0, xh[0], xl[0]);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      if (abl & 4) break;
      const int t = st >> 1, kg = st & 1;
      if (st + 1 < 18) read_x((st + 1) >> 1, (st + 1) & 1, xh[(st + 1) & 1], xl[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                       // (hipcc otherwise sinks the reads behind the step's second instruction)
      if (kg == 0) {
        acc = mfma16<ET>(wh[t][kg], xh[st & 1], acc);
        acc = mfma16<ET>(wh[t][kg], xl[st & 1], acc);
        acc = mfma16<ET>(wl[t][kg], xh[st & 1], acc);
      } else {
        acc2 = mfma16<ET>(wh[t][kg], xh[st & 1], acc2);
        acc2 = mfma16<ET>(wh[t][kg], xl[st & 1], acc2);
        acc2 = mfma16<ET>(wl[t][kg], xh[st & 1], acc2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = __fadd_rn(acc[r], acc2[r]);       // this wave's part: k-group 0's sum + k-group 1's
    // ---- the pair's accumulators meet: this wave hands over the half the partner finishes (registers 8 (1 - kh) .. + 8)
    float4 *red = reinterpret_cast<float4 *>(lds + G::RED_OFF) + ((tile & 1) * G::PAIRS + (wave >> 1)) * 256;
    if (!(abl & 8)) {
      const int o = (1 - kh) * 8;
      red[((1 - kh) * 2 + 0) * 64 + lane] = make_float4(acc[o + 0], acc[o + 1], acc[o + 2], acc[o + 3]);
      red[((1 - kh) * 2 + 1) * 64 + lane] = make_float4(acc[o + 4], acc[o + 5], acc[o + 6], acc[o + 7]);
    }
    // tile + 1's rows: split and into the ring rows behind the window (nobody reads them during this tile)
    if (!(abl & 2)) sstore((tile + 1) * G::BM + G::HALO, next);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (abl & 8) return;
    // ---- epilogue of this wave's 16 channels: rows m0 + wm*32 + lr, channels cbase + 8*j + 4*lh + e (accumulator register 8*kh + 4*j + e)
    const int rbase = m0 + wm * 32;
    const bool valid = (vbits >> lr) & 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 o4 = red[(kh * 2 + j) * 64 + lane];
      const float other[4] = {o4.x, o4.y, o4.z, o4.w};
      const int chl = cbase + 8 * j + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
      const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 64 + chl);
      const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 128 + chl);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sum = __fadd_rn(acc[kh * 8 + j * 4 + e], other[e]);          // chunk 0's part + chunk 1's part (commutative: the same in either wave)
        const float z = fmaxf(fmaf(sum, unscale, b[e]), act_lo) * sc[e] + sh[e];
        y[e] = valid ? z : 0.0f;
      }
      *reinterpret_cast<float4 *>(scr + lr * G::SPITCH + 8 * j + 4 * lh) = make_float4(y[0], y[1], y[2], y[3]);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int frow = it * 16 + (lane >> 2), slot = lane & 3;
      const float4 v = *reinterpret_cast<const float4 *>(scr + frow * G::SPITCH + slot * 4);
      if (!(abl & 1)) *reinterpret_cast<float4 *>(yg + (size_t)(rbase + frow) * p.ldy + cbase + slot * 4) = v;
    }
  };
  uint32_t va = p.row_valid[min((t_begin * G::BM + wm * 32) >> 5, last_word)], vb = 0u;
  if constexpr (!PIPE) {
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += 2) {
      one_tile(tile, sa, sb, va, vb);
      if (tile + 1 < t_end) one_tile(tile + 1, sb, sa, vb, va);
    }
  } else {
    // PIPE: the phases of a tile that are not matrix work - the exchange's second half + epilogue + stores of tile i, the split of tile
    // i + 2's rows... of tile i + 1's rows - are placed INSIDE tile i + 1's K loop (between its steps): the two waves a SIMD holds walk
    // the same phases at the same time (ablations above), so what a wave issues between its matrix instructions runs while the other
    // wave's matrix instructions occupy the pipe, and the stretch outside the K loop shrinks to the exchange write and the barrier.
    // The first pass of the loop finishes a tile of zeros onto the first tile's own rows (no branch: every vector-memory operation
    // of the loop stays unconditional); the real values follow from the same wave, one tile later.
    struct Pend { float mine[8]; uint32_t vbits; int m0, par; };
    auto epi_a = [&](const Pend &pd) {
      const float4 *red = reinterpret_cast<const float4 *>(lds + G::RED_OFF) + (pd.par * G::PAIRS + (wave >> 1)) * 256;
      const bool valid = (pd.vbits >> lr) & 1u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float4 o4 = red[(kh * 2 + j) * 64 + lane];
        const float other[4] = {o4.x, o4.y, o4.z, o4.w};
        const int chl = cbase + 8 * j + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
        const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 64 + chl);
        const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 128 + chl);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sum = __fadd_rn(pd.mine[j * 4 + e], other[e]);
          const float z = fmaxf(fmaf(sum, unscale, b[e]), act_lo) * sc[e] + sh[e];
          y[e] = valid ? z : 0.0f;
        }
        *reinterpret_cast<float4 *>(scr + lr * G::SPITCH + 8 * j + 4 * lh) = make_float4(y[0], y[1], y[2], y[3]);
      }
    };
    auto epi_b = [&](const Pend &pd) {
      const int rbase = pd.m0 + wm * 32;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int frow = it * 16 + (lane >> 2), slot = lane & 3;
        const float4 v = *reinterpret_cast<const float4 *>(scr + frow * G::SPITCH + slot * 4);
        *reinterpret_cast<float4 *>(yg + (size_t)(rbase + frow) * p.ldy + cbase + slot * 4) = v;
      }
    };
    Pend pd;
#pragma unroll
    for (int j = 0; j < 8; ++j) pd.mine[j] = 0.0f;
    pd.vbits = 0u; pd.m0 = t_begin * G::BM; pd.par = 0;
    auto tile_pipe = [&](const int tile, Stage &next, Stage &next2, const uint32_t vbits, uint32_t &vbits_next) {
      const int m0 = tile * G::BM;
      vbits_next = p.row_valid[min((m0 + G::BM + wm * 32) >> 5, last_word)];
      gload((tile + 2) * G::BM + G::HALO, next2);
      const int wb = __builtin_amdgcn_readfirstlane((m0 + G::HALO + wm * 32) % G::RING);
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      uint4 xh[2], xl[2];
      auto read_x = [&](const int t, const int kg, uint4 &h, uint4 &l) {
        const int d = __builtin_amdgcn_readlane(v_taps, t);
        int rr = wb + lr + d;
        rr = rr < 0 ? rr + G::RING : (rr >= G::RING ? rr - G::RING : rr);
        const unsigned char *rowp = lds + rr * G::ROWB;
        const int sw = rr & 15;
        l = *reinterpret_cast<const uint4 *>(rowp + (((kh * 8 + 4 + kg * 2 + lh) ^ sw) << 4));
        h = *reinterpret_cast<const uint4 *>(rowp + (((kh * 8 + kg * 2 + lh) ^ sw) << 4));
      };
      read_x(0, 0, xh[0], xl[0]);
#pragma unroll
      for (int st = 0; st < 18; ++st) {
        const int t = st >> 1, kg = st & 1;
        if (st + 1 < 18) read_x((st + 1) >> 1, (st + 1) & 1, xh[(st + 1) & 1], xl[(st + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma16<ET>(wh[t][kg], xh[st & 1], acc);
        acc = mfma16<ET>(wh[t][kg], xl[st & 1], acc);
        acc = mfma16<ET>(wl[t][kg], xh[st & 1], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (st == 2) { epi_a(pd); __builtin_amdgcn_sched_barrier(0); }             // the previous tile: partner's part + own, epilogue -> the wave's LDS tile
        if (st == 7) { epi_b(pd); __builtin_amdgcn_sched_barrier(0); }             // ... -> 16-byte stores
        if (st == 12) { sstore((tile + 1) * G::BM + G::HALO, next); __builtin_amdgcn_sched_barrier(0); }      // the next tile's rows -> ring
      }
      float4 *red = reinterpret_cast<float4 *>(lds + G::RED_OFF) + ((tile & 1) * G::PAIRS + (wave >> 1)) * 256;
      const int o = (1 - kh) * 8;
      red[((1 - kh) * 2 + 0) * 64 + lane] = make_float4(acc[o + 0], acc[o + 1], acc[o + 2], acc[o + 3]);
      red[((1 - kh) * 2 + 1) * 64 + lane] = make_float4(acc[o + 4], acc[o + 5], acc[o + 6], acc[o + 7]);
#pragma unroll
      for (int j = 0; j < 8; ++j) pd.mine[j] = acc[kh * 8 + j];
      pd.vbits = vbits; pd.m0 = m0; pd.par = tile & 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; tile += 2) {
      tile_pipe(tile, sa, sb, va, vb);
      if (tile + 1 < t_end) tile_pipe(tile + 1, sb, sa, vb, va);
    }
    epi_a(pd);
    epi_b(pd);
  }
  x3_publish_range(range, p.status);
}

// the nine-tap geometries (halo >= pitch + 1 of the stage's grid) and the halo-free ones (1-tap layers over im2col / space-to-depth
// tensors); two image buffers unless the layer is a single chunk
using Q32 = QGeom<4, 1, 2, 1, 82, 82, 1>;      // 32 -> 32, grids of <= 80 bins (halo = pitch + 1 <= 82): 256 rows, one chunk, 52.9 KiB: three workgroups per CU
using Q64 = QGeom<2, 2, 2, 1, 48, 48, 2>;      // 64 channels out, 40-bin grid: 128 rows, 2 x 28 KiB
using Q64P = QGeom<2, 2, 2, 1, 0, 0, 2>;
using Q128 = QGeom<2, 2, 4, 2, 24, 24, 2>;     // 128 out, 20-bin grid: 256 rows, 2 x 38 KiB
using Q128P = QGeom<2, 2, 2, 2, 0, 0, 2>;     // (128-row tiles: two register stages of rows, see QGeom::PF2)
using Q256 = QGeom<1, 4, 4, 2, 16, 16, 2>;     // 256 out (per n tile), 10-bin grid: 128 rows, 2 x 20 KiB
using Q256P = QGeom<1, 4, 4, 2, 0, 0, 2>;
// frames-domain layers the wide f32x kernel (kernels_tdnn_x3.hip: >= 192 output channels, no second input, plain epilogue) does not
// take: ECAPA's Res2Net branches (128 -> 128, 3 dilated taps, sp + x_i as second input) and attention bottleneck (1536 -> 128 + tanh
// + per-utterance bias); halo = kHalo rounded up to 8
using Q64F = QGeom<2, 2, 2, 1, 8, 8, 2>;
using Q128F = QGeom<2, 2, 4, 2, 8, 8, 2>;
using Q256F = QGeom<1, 4, 4, 2, 8, 8, 2>;
// the 128-channel geometries with the long epilogue / a second input: 128-row tiles (64 accumulator registers instead of 128: the
// 256-row forms spill ~20 registers there)
using Q128g = QGeom<2, 2, 2, 2, 24, 24, 2>;
using Q128Fg = QGeom<2, 2, 2, 2, 8, 8, 2>;
using Q128Pg = QGeom<2, 2, 2, 2, 0, 0, 2>;
using Q256g = QGeom<1, 4, 2, 2, 16, 16, 2>;     // (64-row tiles)
using Q256Fg = QGeom<1, 4, 2, 2, 8, 8, 2>;
using Q256Pg = QGeom<1, 4, 2, 2, 0, 0, 2>;

struct QPick { int bm, bn, hlo, hhi, id; };
template <typename G> QPick qpick(int id) { return QPick{G::BM, G::BN, G::HLO, G::HHI, id}; }
// the geometry with the smallest window that holds the layer's taps
bool plain_epilogue(const TdnnKernelParams &p) {
  return (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr && p.seg_scale == nullptr &&
         p.res == nullptr && p.x2 == nullptr;
}
QPick pick_geom(const TdnnKernelParams &p) {
  const bool fast = plain_epilogue(p);
  int lo = 0, hi = 0;
  for (int t = 0; t < p.n_taps; ++t) { lo = std::max(lo, -p.taps[t]); hi = std::max(hi, p.taps[t]); }
  const int halo = std::max(lo, hi);
  const QPick none{0, 0, 0, 0, -1};
  if (p.cout_store == 32) return (p.cin_pad == 32 && halo <= Q32::HLO) ? qpick<Q32>(0) : none;
  if (p.cout_store == 64) return halo == 0 ? qpick<Q64P>(2) : (halo <= 8 ? qpick<Q64F>(7) : (halo <= Q64::HLO ? qpick<Q64>(1) : none));
  if (p.cout_store == 128) {
    if (halo == 0) return fast ? qpick<Q128P>(4) : qpick<Q128Pg>(12);
    if (halo <= 8) return fast ? qpick<Q128F>(8) : qpick<Q128Fg>(11);
    return halo <= Q128::HLO ? (fast ? qpick<Q128>(3) : qpick<Q128g>(10)) : none;
  }
  if (p.cout_store % 256 == 0) {
    if (halo == 0) return fast ? qpick<Q256P>(6) : qpick<Q256Pg>(15);
    if (halo <= 8) return fast ? qpick<Q256F>(9) : qpick<Q256Fg>(14);
    return halo <= Q256::HLO ? (fast ? qpick<Q256>(5) : qpick<Q256g>(13)) : none;
  }
  return none;
}

}  // namespace

// can the layer be packed for / run by the f32x grid kernel?  (shape only: the weights come in p.wconv)
bool grid_conv_x3_shape_ok(int cin_pad, int cout_store) {
  if (cin_pad % QCH != 0 || cin_pad < QCH) return false;
  if (cout_store == 32) return cin_pad == 32;
  return cout_store == 64 || cout_store == 128 || (cout_store > 0 && cout_store % 256 == 0);
}

size_t grid_conv_x3_frag_elems(int cin_pad, int cout_store, int n_taps) {
  return (size_t)(cin_pad / QCH) * n_taps * 2 * (cout_store / 32) * 2 * 512;
}

// [chunk][tap][k-group][n-fragment][hi | lo][lane = (k half lh, channel lr)][8]: channel nf * 32 + lr, k = chunk * 32 + kg * 16 + lh * 8 + e;
// every weight multiplied by `scale` (a power of two) first; lo = w * scale - hi
void pack_grid_conv_x3_frags(const float *w, int out_ch, int in_ch, int tot_ctx, int left_ctx, const int *taps, int n_taps, int cin_pad, int cout_store,
                             int et, float scale, uint16_t *dst) {
  const int nft = cout_store / 32;
  const size_t n = grid_conv_x3_frag_elems(cin_pad, cout_store, n_taps);
  for (size_t i = 0; i < n; ++i) dst[i] = 0;
  for (int co = 0; co < out_ch; ++co)
    for (int t = 0; t < n_taps; ++t) {
      const int k = taps[t] - left_ctx;
      for (int ci = 0; ci < in_ch; ++ci) {
        const int c = ci / QCH, kg = (ci % QCH) / 16, lh = (ci % 16) / 8, e = ci % 8, nf = co / 32, lr = co % 32;
        const size_t frag = ((((size_t)c * n_taps + t) * 2 + kg) * nft + nf) * 2;
        const size_t at = (size_t)(lh * 32 + lr) * 8 + e;
        const float v = w[((size_t)co * in_ch + ci) * tot_ctx + k] * scale;
        const uint16_t h = et == ET_F16 ? f32_to_f16_host(v) : f32_to_bf16_host(v);
        const float hv = et == ET_F16 ? f16_to_f32_host(h) : bf16_to_f32_host(h);
        dst[frag * 512 + at] = h;
        dst[(frag + 1) * 512 + at] = et == ET_F16 ? f32_to_f16_host(v - hv) : f32_to_bf16_host(v - hv);
      }
    }
}

bool grid_conv_x3_supported(const TdnnKernelParams &p) {
  if (p.wconv == nullptr || p.ksplit > 1 || !grid_conv_x3_shape_ok(p.cin_pad, p.cout_store)) return false;
  if ((p.x3_terms & 7) != 7 || !(p.w_unscale > 0.0f)) return false;
  const QPick g = pick_geom(p);
  if (g.id < 0 || p.rows % g.bm != 0 || p.ldx % 4 != 0 || p.ldy % 4 != 0 || (p.x2 != nullptr && p.ldx2 % 4 != 0)) return false;
  if (g.id == 0 && p.cin_pad != QCH) return false;                 // the one-image geometry holds one chunk
  return true;
}

int launch_grid_conv_x3(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(grid_conv_x3_supported(p), "grid conv (f32x): unsupported layer");
  ASV_REQUIRE(p.x3_et == ET_BF16 || p.x3_et == ET_F16, "grid conv (f32x): split type %d", p.x3_et);
  const QPick g = pick_geom(p);
  {
    // the 32 -> 32 and 64 -> 64 nine-tap layers: persistent sliding-window forms (ASV_AMD_X3_PERS=0: the one-tile kernels; =3: the 32-channel
    // form only - it gives the one-tile kernel's bits, the 64-channel form its values to f32 rounding)
    static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
    static const int pers0 = getenv("ASV_AMD_X3_PERS") != nullptr ? atoi(getenv("ASV_AMD_X3_PERS")) : 1;
    const int pers = live && getenv("ASV_AMD_X3_PERS") != nullptr ? atoi(getenv("ASV_AMD_X3_PERS")) : pers0;
    int halo = 0;
    for (int t = 0; t < p.n_taps; ++t) halo = std::max(halo, std::abs(p.taps[t]));
    if (pers && g.id == 0 && plain_epilogue(p) && p.n_taps == 9 && p.cin_pad == 32 && p.cout_store == 32 && halo <= QPers32::HALO && p.rows % QPers32::BM == 0 &&
        p.rows >= QPers32::RING && (pers & 8) == 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      const int n_tiles = p.rows / QPers32::BM, wgs = std::min(n_tiles, cus * 2), per_wg = (n_tiles + wgs - 1) / wgs;
      const dim3 pgrid((n_tiles + per_wg - 1) / per_wg), pblock(256);
      if (p.x3_et == ET_F16) hipLaunchKernelGGL((grid_conv_x3_pers32_kernel<ET_F16>), pgrid, pblock, 0, s, p, per_wg);
      else hipLaunchKernelGGL((grid_conv_x3_pers32_kernel<ET_BF16>), pgrid, pblock, 0, s, p, per_wg);
      ASV_HIP_CHECK(hipGetLastError());
      return ASV_OK;
    }
    if (pers && g.id == 1 && plain_epilogue(p) && p.n_taps == 9 && p.cin_pad == 64 && p.cout_store == 64 && halo <= QPers64<1>::HALO && p.rows % 64 == 0 &&
        p.rows >= QPers64<2>::RING && (pers & 2) == 0) {
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      const int wm = (pers & 4) ? 2 : 1;                               // ASV_AMD_X3_PERS=5: the 64-row / 8-wave form
#ifdef ASV_WITH_ABLATION
      TdnnKernelParams pa = p;
      pa.tune = live && getenv("ASV_AMD_X3_PERS_ABL") != nullptr ? atoi(getenv("ASV_AMD_X3_PERS_ABL")) : 0;
      const TdnnKernelParams &p = pa;
#endif
      const int n_tiles = p.rows / (32 * wm), wgs = std::min(n_tiles, cus * (3 - wm)), per_wg = (n_tiles + wgs - 1) / wgs;
      const dim3 pgrid((n_tiles + per_wg - 1) / per_wg), pblock(256 * wm);
      if (wm == 1 && (pers & 16) == 0) {                               // (default) the tile loop with the non-matrix phases inside the K loop
        if (p.x3_et == ET_F16) hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_F16, 1, true>), pgrid, pblock, 0, s, p, per_wg);
        else hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_BF16, 1, true>), pgrid, pblock, 0, s, p, per_wg);
      } else if (wm == 1) {                                            // ASV_AMD_X3_PERS=17: phase after phase
        if (p.x3_et == ET_F16) hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_F16, 1>), pgrid, pblock, 0, s, p, per_wg);
        else hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_BF16, 1>), pgrid, pblock, 0, s, p, per_wg);
      } else {
        if (p.x3_et == ET_F16) hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_F16, 2>), pgrid, pblock, 0, s, p, per_wg);
        else hipLaunchKernelGGL((grid_conv_x3_pers64_kernel<ET_BF16, 2>), pgrid, pblock, 0, s, p, per_wg);
      }
      ASV_HIP_CHECK(hipGetLastError());
      return ASV_OK;
    }
  }
  const int m_tiles = p.rows / g.bm, n_tiles = p.cout_store / g.bn, nft = p.cout_store / 32;
  const dim3 grid(m_tiles * n_tiles), block(256);
  // the hot instantiations carry the short epilogue only; tanh / sigmoid / per-segment terms / residual / "bn-relu" order / a second
  // input go to the GENERIC ones
  const bool fast = plain_epilogue(p);
  const bool long_epi = !((p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                          p.seg_scale == nullptr && p.res == nullptr);
#define ASV_QCONV1(GEO, MODEV) do { if (p.x3_et == ET_F16) hipLaunchKernelGGL((grid_conv_x3_kernel<GEO, ET_F16, MODEV>), grid, block, 0, s, p, n_tiles, nft); \
                                    else hipLaunchKernelGGL((grid_conv_x3_kernel<GEO, ET_BF16, MODEV>), grid, block, 0, s, p, n_tiles, nft); } while (0)
  // FAST: MODE 0; GEN: MODE 1 (second input, plain epilogue) or 2 (the long epilogue)
#define ASV_QCONV_FAST(GEO) ASV_QCONV1(GEO, 0)
#define ASV_QCONV_GEN(GEO) do { if (long_epi) ASV_QCONV1(GEO, 2); else ASV_QCONV1(GEO, 1); } while (0)
  // the small-tile geometries (ids 0 - 2, 7) carry both epilogues; the 128- / 256-channel ones have a geometry per epilogue (the long one
  // and the second input's registers do not fit beside 128 accumulator registers)
  (void)fast;
  switch (g.id) {
    case 0: if (fast) ASV_QCONV_FAST(Q32); else ASV_QCONV_GEN(Q32); break;
    case 1: if (fast) ASV_QCONV_FAST(Q64); else ASV_QCONV_GEN(Q64); break;
    case 2: if (fast) ASV_QCONV_FAST(Q64P); else ASV_QCONV_GEN(Q64P); break;
    case 7: if (fast) ASV_QCONV_FAST(Q64F); else ASV_QCONV_GEN(Q64F); break;
    case 3: ASV_QCONV_FAST(Q128); break;
    case 4: ASV_QCONV_FAST(Q128P); break;
    case 8: ASV_QCONV_FAST(Q128F); break;
    case 5: ASV_QCONV_FAST(Q256); break;
    case 6: ASV_QCONV_FAST(Q256P); break;
    case 9: ASV_QCONV_FAST(Q256F); break;
    case 10: ASV_QCONV_GEN(Q128g); break;
    case 11: ASV_QCONV_GEN(Q128Fg); break;
    case 12: ASV_QCONV_GEN(Q128Pg); break;
    case 13: ASV_QCONV_GEN(Q256g); break;
    case 14: ASV_QCONV_GEN(Q256Fg); break;
    default: ASV_QCONV_GEN(Q256Pg); break;
  }
#undef ASV_QCONV_FAST
#undef ASV_QCONV_GEN
#undef ASV_QCONV1
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
