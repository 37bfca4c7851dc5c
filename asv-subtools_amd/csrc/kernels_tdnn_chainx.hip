// The layer chain of kernels_tdnn_chain.hip for the f32x precision mode: tdnn -> [1-tap 512 -> 512]* -> 1-tap + fused statistics
// pooling in ONE kernel with f32-grade products (model/xvector.py:77-98: tdnn3 -> tdnn4 -> tdnn5 -> StatisticsPooling;
// components.py:107-149, 410-431; pooling.py:58-67).
//
// Every product runs as three matrix instructions on operand halves, w_hi x_hi + w_hi x_lo + w_lo x_hi (kernels_tdnn_x3.hip).  What
// the chain adds for this mode: the 512-channel output tile of a layer stays in LDS as its [hi | lo] half images - split ONCE,
// by the producing layer's epilogue - and is the B operand of the next layer directly.  The 1-tap layers, which as separate
// launches paid a window DMA, a conversion pass and a workgroup barrier for every 48 matrix instructions plus a prologue, an
// epilogue and an f32 round trip through HBM per tile (0.27 - 0.28 of the mode's matrix peak against 0.40 for the 3-tap layers,
// profiles/r3d_*), run from LDS without any of them; the 1500-channel f32 tensor (783 MB per C2 step) never exists.
//
// Geometry: 64 frames per workgroup (two half images of 64 x 512 = 128 KiB of LDS), 8 waves, wave w = channels 64 w .. 64 w + 63
// of the resident tile (2 x 2 accumulators); last layer: 64-channel units, unit = pass * 8 + wave, operands swapped (lane =
// channel) and the register-only pooling epilogue of kernels_tdnn_chain.hip.  Layer A's f32 window goes through two LDS-DMA
// stages and two converted images (the SHARED form of kernels_tdnn_x3.hip) that live inside the not yet written Y region.
// The fetches of a k-group (4 weight fragments from L2, 4 image fragments from LDS) are pinned between the 12 matrix
// instructions of the previous one.  One workgroup (512 threads, 160 KiB of LDS) per CU.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int XM = 64;                     // frames per workgroup
constexpr int XN = kChainWidth;            // channels of the resident tile (512)
constexpr int XROW = 128;                  // window / image row: 32 f32 = [hi 32 | lo 32] halves
constexpr int XWINR = XM + 2 * kHalo;      // 72 window rows
constexpr int XSTG = XWINR * XROW;         // 9216 B per f32 stage / per image
constexpr int XPCS = XWINR / 8;            // 9 eight-row DMA pieces per window
constexpr int YROW = XN * 2;               // 1024 B per row of a half image
constexpr int YIMG = XM * YROW;            // 65536 B per half image: Yh at 0, Yl at YIMG
constexpr int XPAR = 2 * YIMG;             // bias | scale | shift of the layer in flight (6 KiB)
constexpr int CHAINX_LDS = 2 * YIMG + 8192;
static_assert(4 * XSTG <= YIMG, "layer A's stages and images live inside the Y region");
static_assert(CHAINX_LDS <= 163840, "160 KiB of LDS per CU");

typedef __attribute__((address_space(3))) unsigned char chainx_lds_byte;
struct XTrNo { static constexpr bool value = false; };
struct XTrYes { static constexpr bool value = true; };

__device__ __forceinline__ int wswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void chainx_glds16(const void *gsrc, uint32_t lds_dst) {
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

// two f32 -> hi pair + lo pair of the 16-bit type (kernels_tdnn_x3.hip x3_split)
template <int ET>
__device__ __forceinline__ void split2(float v0, float v1, uint32_t &hi, uint32_t &lo, uint32_t &range) {
  hi = pack_h16x2<ET>(v0, v1);
  if constexpr (ET == ET_F16) range |= h16_range_bits(hi);         // range watch of the half split (device_utils.h)
  float r0, r1;
  if constexpr (ET == ET_F16) {
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
  } else {
    float h0, h1;
    unpack_h16x2<ET>(hi, h0, h1);
    r0 = v0 - h0; r1 = v1 - h1;
  }
  lo = pack_h16x2<ET>(r0, r1);
}

template <int ET>
__global__ __launch_bounds__(512, 2) void tdnn_chainx_kernel(const TdnnChainParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CHAINX_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * XM;
  uint32_t range = 0u;                      // range watch of the half split (device_utils.h): layer A's window and the resident tiles
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(chainx_lds_byte *)lds);
  float *par = reinterpret_cast<float *>(lds + XPAR);
  const uint32_t lane16 = (uint32_t)lane * 16u;

  auto stage_params = [&](const TdnnChainLayer &L) {
    if (tid < 384) {
      const int which = tid >> 7, idx = (tid & 127) * 4;
      float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = (which == 0) ? L.bias : (which == 1 ? L.scale : L.shift);
      if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + idx);
      *reinterpret_cast<float4 *>(par + which * XN + idx) = v;
    }
  };

  struct WF { uint4 h[2], l[2]; };          // this wave's two 32-channel weight fragments of a k-group: hi and lo halves
  struct XF { uint4 h[2], l[2]; };          // the two 32-frame fragments of a k-group: hi and lo halves
  f32x16_t acc[2][2];
  // the accumulators start from bias * w_scale (the weights carry the power of two w_scale; the epilogues multiply by 1 / w_scale):
  // TR = false: acc[i][j][4 q + e] = channel j * 32 + 8 q + 4 lh + e of the wave's slice; TR = true: lane = channel j * 32 + lr
  auto init_acc = [&](const float *bias64, float w_scale, auto tr) {
    if constexpr (decltype(tr)::value) {
      const float b0 = bias64[lr] * w_scale, b1 = bias64[32 + lr] * w_scale;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4 *>(bias64 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i][j][q * 4 + 0] = b4.x * w_scale; acc[i][j][q * 4 + 1] = b4.y * w_scale;
            acc[i][j][q * 4 + 2] = b4.z * w_scale; acc[i][j][q * 4 + 3] = b4.w * w_scale;
          }
        }
    }
  };
  // matrix instruction idx (0..11) of a k-group, term-major (an accumulator recurs every 4th instruction): term 0 w_hi x_hi,
  // 1 w_hi x_lo, 2 w_lo x_hi
  auto mma1 = [&](const XF &x, const WF &w, int idx, auto tr) {
    const int term = idx >> 2, j = (idx >> 1) & 1, i = idx & 1;
    const uint4 a = (term == 2) ? w.l[j] : w.h[j];
    const uint4 b = (term == 1) ? x.l[i] : x.h[i];
    if constexpr (decltype(tr)::value) acc[i][j] = mfma16<ET>(b, a, acc[i][j]);
    else acc[i][j] = mfma16<ET>(a, b, acc[i][j]);
  };

  // ================================ phase 1: layer A, f32 window -> image -> products ================================
  stage_params(p.first);
  {
    const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
    const size_t x_pitch = (size_t)p.ldx * 4;
    const int g_row = lane >> 3, g_slot = lane & 7;
    const int nchunks = p.cin_pad / 32;
    const int n_taps = p.n_taps;
    const int nkg = (p.cin_pad / 64) * 4;                       // 16-channel k-groups per tap in the fragment arrays
    // window of chunk c -> f32 stage c & 1: piece w by wave w, the ninth piece by wave 0 (rows beyond the matrix ends are clamped
    // onto zero gap rows)
    auto piece_off = [&](int grp) -> size_t {
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      return (size_t)row * x_pitch + (size_t)wswz(w, g_slot) * 16u;
    };
    const size_t off_a = piece_off(wave), off_b = piece_off(8);
    auto issue_A = [&](int c) {
      const unsigned char *base = xg + (size_t)c * XROW;
      chainx_glds16(base + off_a, __builtin_amdgcn_readfirstlane(lds_base + (c & 1) * XSTG + wave * 1024));
      if (wave == 0) chainx_glds16(base + off_b, __builtin_amdgcn_readfirstlane(lds_base + (c & 1) * XSTG + 8 * 1024));
    };
    // f32 stage c & 1 -> image c & 1 (at 2 * XSTG): row = [hi of 32 channels (4 slots) | lo (4 slots)]
    auto convert = [&](int c) {
      if (tid < XWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        const unsigned char *src = lds + (c & 1) * XSTG + w * XROW;
        const uint4 a = *reinterpret_cast<const uint4 *>(src + wswz(w, 2 * q) * 16);
        const uint4 b = *reinterpret_cast<const uint4 *>(src + wswz(w, 2 * q + 1) * 16);
        uint4 hi, lo;
        split2<ET>(__uint_as_float(a.x), __uint_as_float(a.y), hi.x, lo.x, range);
        split2<ET>(__uint_as_float(a.z), __uint_as_float(a.w), hi.y, lo.y, range);
        split2<ET>(__uint_as_float(b.x), __uint_as_float(b.y), hi.z, lo.z, range);
        split2<ET>(__uint_as_float(b.z), __uint_as_float(b.w), hi.w, lo.w, range);
        unsigned char *dst = lds + (2 + (c & 1)) * XSTG + w * XROW;
        *reinterpret_cast<uint4 *>(dst + wswz(w, q) * 16) = hi;
        *reinterpret_cast<uint4 *>(dst + wswz(w, 4 + q) * 16) = lo;
      }
    };
    const size_t frag_stride = (size_t)n_taps * nkg * 1024;
    const unsigned char *wh = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 2) * frag_stride + lane16;
    const unsigned char *wl = reinterpret_cast<const unsigned char *>(p.first.wlo) + (size_t)(wave * 2) * frag_stride + lane16;
    auto load_w = [&](int c, int t, int kg, WF &w) {
      const size_t off = ((size_t)t * nkg + (size_t)c * 2 + kg) * 1024;
      w.h[0] = *reinterpret_cast<const uint4 *>(wh + off); w.h[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + off);
      w.l[0] = *reinterpret_cast<const uint4 *>(wl + off); w.l[1] = *reinterpret_cast<const uint4 *>(wl + frag_stride + off);
    };
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    auto x_addr = [&](int c, int t, int kg, uint32_t &ah, uint32_t &al) {
      const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t);
      const int sw = (wrow >> 1) & 7;
      const uint32_t base = (uint32_t)((2 + (c & 1)) * XSTG + wrow * XROW);
      ah = base + (uint32_t)(((kg * 2 + lh) ^ sw) << 4);
      al = base + (uint32_t)(((4 + kg * 2 + lh) ^ sw) << 4);
    };
    issue_A(0);
    WF w0, w1;
    XF x0, x1;
    load_w(0, 0, 0, w0);
    if (nchunks > 1) issue_A(1);
    // window 0 landed (everything but this wave's youngest piece(s): at most 2) -> image 0; its stage then takes window 2
    if (nchunks > 1) { if (wave == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    convert(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunks > 2) issue_A(2);
    if (nchunks > 1) convert(1);
    init_acc(p.first.bias + wave * 64, p.first.w_scale, XTrNo{});
    {
      uint32_t ah, al;
      x_addr(0, 0, 0, ah, al);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        x0.h[i] = *reinterpret_cast<const uint4 *>(lds + ah + i * 32 * XROW);
        x0.l[i] = *reinterpret_cast<const uint4 *>(lds + al + i * 32 * XROW);
      }
    }
    const int G = nchunks * n_taps * 2;
    int c = 0, t = 0;
    auto step = [&](const XF &xc, const WF &wc, XF &xn, WF &wn, int g) {
      const int kg = g & 1;
      int c2 = c, t2 = t, kg2 = kg + 1;
      if (kg2 == 2) { kg2 = 0; t2 = t + 1; if (t2 == n_taps) { t2 = 0; c2 = c + 1; } }
      const bool more = g + 1 < G;
      if (!more) { c2 = c; t2 = t; kg2 = kg; }                     // the last k-group re-fetches itself (valid memory, never used)
      const bool enter = more && c2 != c;
      if (enter) {
        // entering chunk c + 1 (kernels_tdnn_x3.hip, SHARED form): its image is complete, window c + 2 has landed (older than the
        // youngest 4 operations: the fragments of k-group g); window c + 2 becomes image c & 1 now, its stage takes window c + 3 -
        // issued BEHIND this step's weight fetches (below): the vector-memory counter retires in order, a fetch issued behind the
        // DMA could not be consumed before the window has landed
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nchunks) convert(c + 2);
      }
      const size_t woff = ((size_t)t2 * nkg + (size_t)c2 * 2 + kg2) * 1024;
      uint32_t ah, al;
      x_addr(c2, t2, kg2, ah, al);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        // 8 fetches in front of the first 4 pairs: the weight fragments first (L2 latency), then the image fragments
        if (pr == 0) { wn.h[0] = *reinterpret_cast<const uint4 *>(wh + woff); wn.h[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + woff); }
        if (pr == 1) { wn.l[0] = *reinterpret_cast<const uint4 *>(wl + woff); wn.l[1] = *reinterpret_cast<const uint4 *>(wl + frag_stride + woff); }
        if (pr == 2 && enter && c + 3 < nchunks) issue_A(c + 3);
        if (pr == 2) { xn.h[0] = *reinterpret_cast<const uint4 *>(lds + ah); xn.h[1] = *reinterpret_cast<const uint4 *>(lds + ah + 32 * XROW); }
        if (pr == 3) { xn.l[0] = *reinterpret_cast<const uint4 *>(lds + al); xn.l[1] = *reinterpret_cast<const uint4 *>(lds + al + 32 * XROW); }
        mma1(xc, wc, 2 * pr, XTrNo{});
        mma1(xc, wc, 2 * pr + 1, XTrNo{});
        __builtin_amdgcn_sched_barrier(0);
      }
      c = c2; t = t2;
    };
    for (int g = 0; g < G; g += 2) {
      step(x0, w0, x1, w1, g);
      if (g + 1 < G) step(x1, w1, x0, w0, g + 1);
    }
  }

  // epilogue of a 512-wide layer: acc / w_scale -> [ReLU] -> [folded BN unless it sits in the next layer's weights] -> hi / lo halves
  // -> Yh / Yl (row-major, 16-byte slots XOR-swizzled by row & 15)
  auto store_Y = [&](int relu, bool affine, float unscale) {
    const float act_lo = relu ? 0.0f : -INFINITY;
    const int rx = lr & 15;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chl = wave * 64 + j * 32 + 8 * q + 4 * lh;
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (affine) {
          const float4 sc4 = *reinterpret_cast<const float4 *>(par + XN + chl);
          const float4 sh4 = *reinterpret_cast<const float4 *>(par + 2 * XN + chl);
          sc[0] = sc4.x; sc[1] = sc4.y; sc[2] = sc4.z; sc[3] = sc4.w;
          sh[0] = sh4.x; sh[1] = sh4.y; sh[2] = sh4.z; sh[3] = sh4.w;
        }
        // 4 consecutive channels = 8 bytes inside the 16-byte slot (wave * 8 + j * 4 + q), half lh
        const uint32_t slot_off = (uint32_t)((((wave * 8 + j * 4 + q) ^ rx) << 4) + lh * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = max_lo(acc[i][j][q * 4 + e] * unscale, act_lo);
            y[e] = affine ? fmaf(v, sc[e], sh[e]) : v;
          }
          uint2 hi, lo;
          split2<ET>(y[0], y[1], hi.x, lo.x, range);
          split2<ET>(y[2], y[3], hi.y, lo.y, range);
          unsigned char *dst = lds + (i * 32 + lr) * YROW + slot_off;
          *reinterpret_cast<uint2 *>(dst) = hi;
          *reinterpret_cast<uint2 *>(dst + YIMG) = lo;
        }
      }
  };

  // main loop of a layer whose input is Y: K = 512 = 32 k-groups of 16 channels, fetches of the next k-group pinned between the
  // matrix instructions of the current one, no barrier.  wbh / wbl: wave-uniform bases of the hi / lo fragment arrays of this
  // wave's (or unit's) first 32-channel fragment; the second follows at + frag_stride.
  auto yloop = [&](const unsigned char *wbh, const unsigned char *wbl, const float *bias64, float w_scale, auto tr) {
    const size_t frag_stride = (size_t)(XN / 16) * 1024;        // 32 k-groups x 1 KiB
    const uint32_t yb = (uint32_t)(lr * YROW);
    const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
    auto load_g = [&](int g, XF &x, WF &w) {
      const size_t off = (size_t)g * 1024 + lane16;
      w.h[0] = *reinterpret_cast<const uint4 *>(wbh + off); w.h[1] = *reinterpret_cast<const uint4 *>(wbh + frag_stride + off);
      w.l[0] = *reinterpret_cast<const uint4 *>(wbl + off); w.l[1] = *reinterpret_cast<const uint4 *>(wbl + frag_stride + off);
      const uint32_t a = yb + ((((uint32_t)(g * 2)) ^ sx) << 4);            // slot (2 g + lh) ^ (lr & 15)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        x.h[i] = *reinterpret_cast<const uint4 *>(lds + a + i * 32 * YROW);
        x.l[i] = *reinterpret_cast<const uint4 *>(lds + YIMG + a + i * 32 * YROW);
      }
    };
    XF x0, x1;
    WF w0, w1;
    load_g(0, x0, w0);
    init_acc(bias64, w_scale, tr);
    auto step = [&](const XF &xc, const WF &wc, XF &xn, WF &wn, int gn) {
      const size_t off = (size_t)gn * 1024 + lane16;
      const uint32_t a = yb + ((((uint32_t)(gn * 2)) ^ sx) << 4);
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        if (pr == 0) { wn.h[0] = *reinterpret_cast<const uint4 *>(wbh + off); wn.h[1] = *reinterpret_cast<const uint4 *>(wbh + frag_stride + off); }
        if (pr == 1) { wn.l[0] = *reinterpret_cast<const uint4 *>(wbl + off); wn.l[1] = *reinterpret_cast<const uint4 *>(wbl + frag_stride + off); }
        if (pr == 2) { xn.h[0] = *reinterpret_cast<const uint4 *>(lds + a); xn.h[1] = *reinterpret_cast<const uint4 *>(lds + a + 32 * YROW); }
        if (pr == 3) { xn.l[0] = *reinterpret_cast<const uint4 *>(lds + YIMG + a); xn.l[1] = *reinterpret_cast<const uint4 *>(lds + YIMG + a + 32 * YROW); }
        mma1(xc, wc, 2 * pr, tr);
        mma1(xc, wc, 2 * pr + 1, tr);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int g = 0; g < XN / 16; g += 2) {
      step(x0, w0, x1, w1, g + 1);
      step(x1, w1, x0, w0, min(g + 2, XN / 16 - 1));            // the last k-group re-fetches itself (never used)
    }
  };

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done with layer A's stages and images: Y may be written
  asm volatile("" ::: "memory");
  store_Y(p.first.relu, p.first.scale != nullptr, 1.0f / p.first.w_scale);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // Y complete; the constants of layer A are dead
  asm volatile("" ::: "memory");

  // ================================ middle layers: Y -> Y ================================
#pragma unroll 1
  for (int m = 0; m < p.n_mid; ++m) {
    const TdnnChainLayer &L = p.mid[m];
    stage_params(L);
    const size_t frag_stride = (size_t)(XN / 16) * 1024;
    const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(wave * 2) * frag_stride;
    const unsigned char *wbl = reinterpret_cast<const unsigned char *>(L.wlo) + (size_t)(wave * 2) * frag_stride;
    yloop(wbh, wbl, L.bias + wave * 64, L.w_scale, XTrNo{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody reads the old Y any more (and the staged constants are visible)
    asm volatile("" ::: "memory");
    store_Y(L.relu, L.scale != nullptr, 1.0f / L.w_scale);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ================================ last layer + fused statistics pooling ================================
  {
    const TdnnChainLayer &L = p.last;
    const float act_lo = L.relu ? 0.0f : -INFINITY;
    const float unscale = 1.0f / L.w_scale;
    const int tile = m0 >> 6;
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && m0 + k < p.rows) first_seg = p.row_seg[m0 + k];
    const int rowseg = p.row_seg[m0 + lane];                  // the tile's 64 rows: lane l = row l
    const size_t frag_stride = (size_t)(XN / 16) * 1024;
#pragma unroll 1
    for (int cb = wave * 64; cb < L.cout_pad; cb += 512) {
      const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb / 32) * frag_stride;
      const unsigned char *wbl = reinterpret_cast<const unsigned char *>(L.wlo) + (size_t)(cb / 32) * frag_stride;
      yloop(wbh, wbl, L.bias + cb, L.w_scale, XTrYes{});
      // Pooling epilogue, registers only (kernels_tdnn_chain.hip): acc[i][j][r] = channel cb + j*32 + lr, frame i*32 + 8 (r >> 2) +
      // 4 lh + (r & 3); a lane sums its own frames per utterance about the pivot of its first frame, the two lane halves publish
      //   P[tile of 64 rows][segment slot][lh][3 = sum (u - pv), sum (u - pv)^2, pv][channel]
      // with the BN scale applied at publication; pool_finish_kernel merges the parts and adds the BN shift.
      const float sc[2] = {L.scale != nullptr ? L.scale[cb + lr] : 1.0f, L.scale != nullptr ? L.scale[cb + 32 + lr] : 1.0f};
      float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
      int cur_seg = -1;                      // uniform: all lanes walk the utterances of the tile together
      bool have = false;                     // per lane: pv is a frame of cur_seg (the lane has had a frame of it in this tile)
      auto publish = [&]() {
        const int slot = cur_seg - first_seg;
        if (cur_seg >= 0 && slot >= 0 && slot < p.pool_slots) {
          float *dst = p.pool_partial + ((size_t)((tile * p.pool_slots + slot) * 2 + lh) * 3) * p.ld_partial + cb + lr;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (cb + j * 32 + lr < p.ld_partial) {
              dst[j * 32] = ps[j] * sc[j];
              dst[j * 32 + p.ld_partial] = pq[j] * sc[j] * sc[j];
              dst[j * 32 + 2 * p.ld_partial] = pv[j] * sc[j];
            }
        }
      };
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int shift = i * 32;
        uint32_t rem = (uint32_t)(__builtin_amdgcn_ballot_w64(rowseg >= 0) >> shift);       // rows of the fragment that belong to an utterance
        if (rem == 0) continue;                                                              // gap rows only
        float u[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) u[j][r] = max_lo(acc[i][j][r] * unscale, act_lo);
        while (rem != 0) {                                       // one run per utterance present, in row order
          const int sg = __builtin_amdgcn_readlane(rowseg, shift + __builtin_ctz(rem));
          const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(rowseg == sg) >> shift) & rem;
          rem &= ~bits;
          const bool fresh = sg != cur_seg;
          if (fresh) {
            publish();
            cur_seg = sg;
            have = false;
#pragma unroll
            for (int j = 0; j < 2; ++j) { ps[j] = 0.0f; pq[j] = 0.0f; }
          }
          // register r of this lane holds frame 8 (r >> 2) + 4 lh + (r & 3) -> bit r of the lane's mask
          const uint32_t x = bits >> (4 * lh);
          const uint32_t lm = (x & 0xfu) | ((x >> 4) & 0xf0u) | ((x >> 8) & 0xf00u) | ((x >> 12) & 0xf000u);
          // pivot = the lane's FIRST frame of the utterance, in whichever fragment of the tile that frame lies.  (Until round 5 the
          // pivot was only taken in the utterance's first fragment: a lane half without a frame there - an utterance starting in
          // the last rows of a fragment - kept the previous utterance's pivot for the rest of the tile.  Harmless between
          // utterances of like scale, but next to one whose activations are 1e5 x larger the sums about that pivot cancelled:
          // an embedding that depended on its batch neighbour, tests/test_gpu_xvector.py::test_pooled_moments_ignore_the_neighbour.)
          const bool need = !have && lm != 0;
          if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const int rsel = need ? __builtin_ctz(lm) : 16;
#pragma unroll
            for (int r = 15; r >= 0; --r) {
              const bool hit = rsel == r;
              pv[0] = hit ? u[0][r] : pv[0];
              pv[1] = hit ? u[1][r] : pv[1];
            }
            have = have || need;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tm = (int)(lm << (31 - r)) >> 31;          // all ones where the frame is in the run
            const float da = __int_as_float(__float_as_int(u[0][r] - pv[0]) & tm), db = __int_as_float(__float_as_int(u[1][r] - pv[1]) & tm);
            ps[0] += da; pq[0] = fmaf(da, da, pq[0]);
            ps[1] += db; pq[1] = fmaf(db, db, pq[1]);
          }
        }
      }
      publish();
    }
  }
  x3_publish_range(range, p.status);
}

}  // namespace

int launch_tdnn_chainx(const TdnnChainParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows % XM == 0 && p.rows >= XM, "tdnn(chainx): rows %d not a multiple of %d", p.rows, XM);
  ASV_REQUIRE(p.cin_pad % 64 == 0 && p.cin_pad >= 64 && p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS, "tdnn(chainx): first layer with %d channels / %d taps", p.cin_pad, p.n_taps);
  ASV_REQUIRE(p.first.wfrag && p.first.wlo && p.last.wfrag && p.last.wlo && p.last.bias && p.n_mid >= 0 && p.n_mid <= 2 && p.last.cout_pad % 64 == 0,
              "tdnn(chainx): incomplete layer description");
  for (int m = 0; m < p.n_mid; ++m) ASV_REQUIRE(p.mid[m].wfrag && p.mid[m].wlo, "tdnn(chainx): middle layer %d without split weights", m);
  ASV_REQUIRE(p.first.w_scale > 0.0f && p.last.w_scale > 0.0f, "tdnn(chainx): weight scales missing");
  ASV_REQUIRE(p.pool_partial && p.row_seg && p.pool_slots >= 1, "tdnn(chainx): the last layer feeds the fused pooling (partials / row map missing)");
  for (int t = 0; t < p.n_taps; ++t) ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(chainx): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const dim3 grid(p.rows / XM), block(512);
  if (p.et == ET_F16) hipLaunchKernelGGL(tdnn_chainx_kernel<ET_F16>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(tdnn_chainx_kernel<ET_BF16>, grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
