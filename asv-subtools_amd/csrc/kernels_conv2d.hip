// 3x3 convolutions of the ResNet34 trunk with few channels (C = 32 / 64: 13 of its 32 convolutions and 60 % of its
// time on the generic 128 x 128 GEMM tile, which wastes 3/4 of its columns at N = 32 and half of its K chunk at
// C = 32) - reference libs/nnet/resnet.py:12-20 (conv3x3) inside BasicBlock (resnet.py:30-76), eval BatchNorm folded.
//
// The grid domain stores [B, C, F, T] as rows = (time, frequency) positions, frequency fastest, pitch F + 1, so a
// stride-1 3x3 convolution is 9 row-offset taps dt * pitch + df of a [rows][C] matrix (runtime.hip).
// With C <= 64 the whole K extent of a tile is ONE window:
//   workgroup = 256 output rows x all output channels; window = 256 + 2 * halo rows x C channels (27 | 54 KiB) brought
//   in once by LDS-DMA; the nine taps read it shifted - no K-chunk loop, no barrier after the first.
//   Weights: C = 32: all 9 x 2 MFMA fragments of the layer live in 72 VGPRs for the whole kernel;
//            C = 64: 8 fragments per tap stream from L2 one tap ahead (fragment order, 1 KiB wave loads).
//   wave = 64 rows (2 accumulator fragments) x all channels; per tap and 16-channel k-group one ds_read_b128 feeds
//   one (C = 32) or two (C = 64) MFMAs - LDS-read bound by construction, HBM bound at the kernel level
//   (64 B in + 64 B out per row against 18.4 kFLOP).
// LDS rows are C * 2 bytes; 16-byte slot s of window row w sits at s ^ ((w >> 2) & 3) (64-byte rows) or
// s ^ ((w >> 1) & 7) (128-byte rows): conflict-free ds_read_b128 for any tap shift; the DMA applies the same
// permutation on the source side.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int CBM = 256;              // output rows per workgroup
constexpr int CHALO = 88;             // window halo: >= pitch + 1 (<= 84), a multiple of 8
constexpr int CWIN = CBM + 2 * CHALO;  // 432 window rows

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

template <int CIN> __device__ __forceinline__ int cswz(int row, int slot) {
  return CIN == 32 ? (slot ^ ((row >> 2) & 3)) : (slot ^ ((row >> 1) & 7));
}

__device__ __forceinline__ void conv_glds16(const void *gsrc, uint32_t lds_dst) {
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

// weights: [tap][k-group][n-fragment][lane = (k half lh, channel lr)][8] bf16, k = kg * 16 + lh * 8 + e
template <int CIN, int NF, bool GENERIC, int ET = ET_BF16>
__global__ __launch_bounds__(256, CIN == 32 ? 3 : 2) void grid_conv_narrow_kernel(const TdnnKernelParams p) {
  constexpr int ROWB = CIN * 2;                       // bytes per window row
  constexpr int SLOTS = ROWB / 16;                    // 4 | 8
  constexpr int RPP = 1024 / ROWB;                    // window rows per 1 KiB DMA piece: 16 | 8
  constexpr int PIECES = CWIN / RPP;                  // 27 | 54
  constexpr int KG = CIN / 16;                        // k-groups per tap: 2 | 4
  __shared__ __attribute__((aligned(16))) unsigned char win[CWIN * ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = xcd_swizzle(blockIdx.x, gridDim.x) * CBM;

  // ---- window: one LDS-DMA instruction per 1 KiB piece, rows clamped onto the matrix (its first / last rows are gaps)
  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte_t *)win);
  for (int piece = wave; piece < PIECES; piece += 4) {
    const int w = piece * RPP + lane / SLOTS;
    const int row = min(max(m0 - CHALO + w, 0), p.rows - 1);
    const int src_slot = cswz<CIN>(w, lane % SLOTS);
    conv_glds16(xg + (size_t)row * ((size_t)p.ldx * 2) + src_slot * 16, __builtin_amdgcn_readfirstlane(lds_base + piece * 1024));
  }
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  const uint4 *wf = reinterpret_cast<const uint4 *>(p.wconv) + lane;            // fragment f at wf[f * 64]
  // developer aid (ASV_AMD_CONV_ABL with ASV_AMD_LIVE_TUNE=1; results are garbage): bit 0 = no output stores, bit 1 = one tap
  // instead of nine (the window is still fetched whole)
  const int abl = p.tune;

  f32x16_t acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.0f;

  auto read_x = [&](int d, int kg, int i) {
    const int w = CHALO + wave * 64 + i * 32 + lr + d;
    return *reinterpret_cast<const uint4 *>(win + w * ROWB + cswz<CIN>(w, kg * 2 + lh) * 16);
  };

  if constexpr (CIN == 32) {
    uint4 wr[9][KG][NF];                              // the whole layer: 18 fragments
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int n = 0; n < NF; ++n) wr[t][kg][n] = wf[((t * KG + kg) * NF + n) * 64];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if ((abl & 2) && t > 0) break;
      const int d = __builtin_amdgcn_readlane(v_taps, t);
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 x = read_x(d, kg, i);
#pragma unroll
          for (int n = 0; n < NF; ++n)
            acc[i][n] = mfma16<ET>(wr[t][kg][n], x, acc[i][n]);
        }
    }
  } else {
    uint4 wa[KG][NF], wb[KG][NF];
    auto fetch = [&](int t, uint4 (&w)[KG][NF]) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int n = 0; n < NF; ++n) w[kg][n] = wf[((t * KG + kg) * NF + n) * 64];
    };
    auto tap = [&](int t, const uint4 (&w)[KG][NF]) {
      const int d = __builtin_amdgcn_readlane(v_taps, t);
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 x = read_x(d, kg, i);
#pragma unroll
          for (int n = 0; n < NF; ++n)
            acc[i][n] = mfma16<ET>(w[kg][n], x, acc[i][n]);
        }
    };
    fetch(0, wa);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 8; t += 2) {                  // taps 0..7 in pairs, the other register set one tap ahead
      fetch(t + 1, wb);
      tap(t, wa);
      fetch(t + 2, wa);
      tap(t + 1, wb);
    }
    tap(8, wa);
  }

  // ---- epilogue: acc[i][n][r] = row m0 + wave*64 + i*32 + lr, channel n*32 + 8*(r>>2) + 4*lh + (r&3): 8-byte stores
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wave * 64 + i * 32 + lr;
    const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = n * 32 + 8 * q + 4 * lh;
        if (ch >= p.cout_store) continue;
        const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
        const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<true>(p, acc[i][n][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][n][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        if (!(abl & 1) || pk.x == 0x12345678u) *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)row * p.ldy + ch) = pk;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The C = 128 / 256 stages (17 of the trunk's 32 convolutions, 2.0 of the 7.5 ms of a ResNet34-SE step on the generic
// 128 x 128 register-staged tile at ~600 TFLOP/s).  Same idea as above - the whole K extent of a tile is ONE LDS window, no
// K-chunk ring, no barrier after the first - with the wave tiling of the wide frame-layer kernel (kernels_tdnn_v3.hip):
//   wave = 128 rows x 64 channels (4 x 2 accumulators): one 1 KiB weight fragment from L2 feeds 4 MFMAs, one ds_read_b128
//   feeds 2 - half the LDS-read and L2 traffic per MFMA of 64-row waves;
//   C = 128: workgroup = 256 rows x 128 channels (2 x 2 waves), window (256 + 2 x 24) rows x 256 B = 76 KiB;
//   C = 256: workgroup = 128 rows x 256 channels (1 x 4 waves), window (128 + 2 x 16) rows x 512 B = 80 KiB;
//   two workgroups per CU either way.  The halo (>= pitch + 1 of the stage's grid: 21-bin grids at C = 128, 11-bin grids at
//   C = 256 for the 82 frequency bins the trunk accepts at most) bounds the layers this kernel takes; others stay on the
//   generic tile.
//   K walks 64-channel chunks x taps in steps of 4 k-groups; the fragments of a step's k-group are re-fetched for the next step as soon
//   as their MFMAs have issued (8 fragments = 32 VGPRs in flight), rows are read one k-group ahead.  Same (tap, k-group)
//   accumulation order as the generic tile: bit-identical outputs (tests/test_gpu_resnet.py).
//   16-byte slot s of window row w sits at s ^ (w & 15): conflict-free ds_read_b128 for any tap shift.
template <int CIN> struct WideGeom {
  static constexpr int HALO = CIN == 128 ? 24 : 16;
  static constexpr int BM = CIN == 128 ? 256 : 128;
  static constexpr int WMS = BM / 128, WNS = 4 / WMS;             // waves along rows / along channels
  static constexpr int WIN = BM + 2 * HALO;                        // 304 | 160 rows
  static constexpr int ROWB = CIN * 2;                             // 256 | 512 bytes
  static constexpr int SLOTS = ROWB / 16;                          // 16 | 32
  static constexpr int RPP = 1024 / ROWB;                          // window rows per 1 KiB DMA piece: 4 | 2
  static constexpr int PIECES = WIN / RPP;                         // 76 | 80
  static constexpr int KG = CIN / 16;                              // k-groups per tap: 8 | 16
  static constexpr int NFR = CIN / 32;                             // 32-channel output fragments of the layer: 4 | 8
  static constexpr int STEPS = 9 * KG / 4;                         // 18 | 36 steps of 4 k-groups
  static_assert(WNS * 64 == CIN && WIN % RPP == 0 && WIN * ROWB <= 81920, "wide grid conv geometry");
};

template <int CIN, bool GENERIC, int ET = ET_BF16>
__global__ __launch_bounds__(256, 2) void grid_conv_wide_kernel(const TdnnKernelParams p) {
  using G = WideGeom<CIN>;
  __shared__ __attribute__((aligned(16))) unsigned char win[G::WIN * G::ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WNS, wn = wave % G::WNS;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = xcd_swizzle(blockIdx.x, gridDim.x) * G::BM;

  // ---- window: one LDS-DMA instruction per 1 KiB piece, rows clamped onto the matrix (its first / last rows are gaps)
  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte_t *)win);
  for (int piece = wave; piece < G::PIECES; piece += 4) {
    const int w = piece * G::RPP + lane / G::SLOTS;
    const int row = min(max(m0 - G::HALO + w, 0), p.rows - 1);
    const int src_slot = (lane % G::SLOTS) ^ (w & 15);
    conv_glds16(xg + (size_t)row * ((size_t)p.ldx * 2) + src_slot * 16, __builtin_amdgcn_readfirstlane(lds_base + piece * 1024));
  }
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  // weights: [tap][k-group][n-fragment][lane][8]; this wave's two fragments are n-fragments wn * 2 and wn * 2 + 1
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wconv) + (size_t)(wn * 2) * 1024 + (size_t)lane * 16;
  auto frag_ptr = [&](int t, int kg, int j) { return wbase + ((size_t)(t * G::KG + kg) * G::NFR + j) * 1024; };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  struct XF { uint4 x[4]; };
  auto read_x1 = [&](int d, int kg_abs, int i, XF &f) {           // kg_abs: k-group within the tap (0 .. KG-1)
    const int w = G::HALO + wm * 128 + i * 32 + lr + d;
    f.x[i] = *reinterpret_cast<const uint4 *>(win + w * G::ROWB + (((kg_abs * 2 + lh) ^ (w & 15)) << 4));
  };
  uint4 wf[4][2];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    wf[kg][0] = *reinterpret_cast<const uint4 *>(frag_ptr(0, kg, 0));
    wf[kg][1] = *reinterpret_cast<const uint4 *>(frag_ptr(0, kg, 1));
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // the window pieces are older than the 8 fragment loads
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  XF x0, x1;
  {
    const int d0 = __builtin_amdgcn_readlane(v_taps, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) read_x1(d0, 0, i, x0);
  }
#pragma unroll 1
  for (int st = 0; st < G::STEPS; ++st) {
    // 64-channel chunk outermost, taps inside it: the K order of the generic tile (kernels_tdnn.hip), hence the same bits
    const int c4 = st / 9, t = st % 9;
    const int stn = st + 1 < G::STEPS ? st + 1 : st;               // the last step re-fetches its own fragments (never used)
    const int c4n = stn / 9, tn = stn % 9;
    const int d = __builtin_amdgcn_readlane(v_taps, t), dn = __builtin_amdgcn_readlane(v_taps, tn);
    auto group = [&](const XF &xc, int kg, XF &xn, int d_next, int kg_abs_next) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // pair q: channel fragment j = q / 2 against row fragments 2 (q % 2) and + 1; one row read of the next k-group in
        // front of every pair, the fragment of the next step behind its last pair (the order of kernels_tdnn_v3.hip)
        read_x1(d_next, kg_abs_next, q, xn);
#pragma unroll
        for (int i = (q % 2) * 2; i < (q % 2) * 2 + 2; ++i)
          acc[i][q / 2] = mfma16<ET>(wf[kg][q / 2], xc.x[i], acc[i][q / 2]);
        if (q == 1) wf[kg][0] = *reinterpret_cast<const uint4 *>(frag_ptr(tn, c4n * 4 + kg, 0));
        if (q == 3) wf[kg][1] = *reinterpret_cast<const uint4 *>(frag_ptr(tn, c4n * 4 + kg, 1));
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    group(x0, 0, x1, d, c4 * 4 + 1);
    group(x1, 1, x0, d, c4 * 4 + 2);
    group(x0, 2, x1, d, c4 * 4 + 3);
    group(x1, 3, x0, dn, c4n * 4);
  }

  // ---- epilogue: acc[i][j][r] = row m0 + wm*128 + i*32 + lr, channel wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3): 8-byte stores
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = wn * 64 + j * 32 + 8 * q + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
      const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 128 + i * 32 + lr;
        const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<true>(p, acc[i][j][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)row * p.ldy + ch) = pk;
      }
    }
}

// The stride-2 3x3 convolution that opens the C = 64 stage (32 -> 64 channels, resnet.py:352-368 through BasicBlock's conv1) in its
// "space to depth" form (pytorch/libs/nnet/resnet.py emit_conv_bn): the four phases of the input grid side by side = 128 input
// channels on the OUTPUT grid, 4 taps {-(pitch' + 1), -pitch', -1, 0}, 64 output channels.  (As im2col + 1-tap GEMM this layer cost
// 342 + 231 us of a 6.6 ms step; on the generic 128 x 128 tile its 4-tap form ran at 87 TFLOP/s.)  The wide kernel's scheme with
// the geometry this layer needs: all taps look BACK, so the 304-row window is 48 rows of halo + 256 output rows; a wave owns 64
// rows x all 64 channels (2 x 2 accumulators).  Weights [tap][k-group][2 n-fragments][lane][8]; K order = chunk outermost, taps
// inside, as the generic tile: bit-identical to it.
constexpr int S2D_HLO = 48, S2D_BM = 256, S2D_WIN = S2D_BM + S2D_HLO, S2D_ROWB = 256, S2D_KG = 8, S2D_NT = 4;
template <bool GENERIC, int ET = ET_BF16>
__global__ __launch_bounds__(256, 2) void grid_conv_s2d_kernel(const TdnnKernelParams p) {
  constexpr int SLOTS = S2D_ROWB / 16, RPP = 1024 / S2D_ROWB, PIECES = S2D_WIN / RPP, STEPS = S2D_NT * S2D_KG / 4;
  static_assert(S2D_WIN % RPP == 0 && S2D_WIN * S2D_ROWB <= 81920, "s2d grid conv geometry");
  __shared__ __attribute__((aligned(16))) unsigned char win[S2D_WIN * S2D_ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = xcd_swizzle(blockIdx.x, gridDim.x) * S2D_BM;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte_t *)win);
  for (int piece = wave; piece < PIECES; piece += 4) {
    const int w = piece * RPP + lane / SLOTS;
    const int row = min(max(m0 - S2D_HLO + w, 0), p.rows - 1);
    const int src_slot = (lane % SLOTS) ^ (w & 15);
    conv_glds16(xg + (size_t)row * ((size_t)p.ldx * 2) + src_slot * 16, __builtin_amdgcn_readfirstlane(lds_base + piece * 1024));
  }
  const int v_taps = p.taps[lane < S2D_NT ? lane : 0];
  const unsigned char *wbase = reinterpret_cast<const unsigned char *>(p.wconv) + (size_t)lane * 16;
  auto frag_ptr = [&](int t, int kg, int j) { return wbase + ((size_t)(t * S2D_KG + kg) * 2 + j) * 1024; };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  struct XF { uint4 x[2]; };
  auto read_x1 = [&](int d, int kg_abs, int i, XF &f) {
    const int w = S2D_HLO + wave * 64 + i * 32 + lr + d;
    f.x[i] = *reinterpret_cast<const uint4 *>(win + w * S2D_ROWB + (((kg_abs * 2 + lh) ^ (w & 15)) << 4));
  };
  uint4 wf[4][2];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    wf[kg][0] = *reinterpret_cast<const uint4 *>(frag_ptr(0, kg, 0));
    wf[kg][1] = *reinterpret_cast<const uint4 *>(frag_ptr(0, kg, 1));
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // the window pieces are older than the 8 fragment loads
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  XF x0, x1;
  {
    const int d0 = __builtin_amdgcn_readlane(v_taps, 0);
    read_x1(d0, 0, 0, x0);
    read_x1(d0, 0, 1, x0);
  }
#pragma unroll 1
  for (int st = 0; st < STEPS; ++st) {
    const int c4 = st / S2D_NT, t = st % S2D_NT;
    const int stn = st + 1 < STEPS ? st + 1 : st;                  // the last step re-fetches its own fragments (never used)
    const int c4n = stn / S2D_NT, tn = stn % S2D_NT;
    const int d = __builtin_amdgcn_readlane(v_taps, t), dn = __builtin_amdgcn_readlane(v_taps, tn);
    auto group = [&](const XF &xc, int kg, XF &xn, int d_next, int kg_abs_next) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        read_x1(d_next, kg_abs_next, j, xn);
        acc[0][j] = mfma16<ET>(wf[kg][j], xc.x[0], acc[0][j]);
        acc[1][j] = mfma16<ET>(wf[kg][j], xc.x[1], acc[1][j]);
        wf[kg][j] = *reinterpret_cast<const uint4 *>(frag_ptr(tn, c4n * 4 + kg, j));
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    group(x0, 0, x1, d, c4 * 4 + 1);
    group(x1, 1, x0, d, c4 * 4 + 2);
    group(x0, 2, x1, d, c4 * 4 + 3);
    group(x1, 3, x0, dn, c4n * 4);
  }

  // ---- epilogue: acc[i][j][r] = row m0 + wave*64 + i*32 + lr, channel j*32 + 8*(r>>2) + 4*lh + (r&3): 8-byte stores
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = j * 32 + 8 * q + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
      const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = m0 + wave * 64 + i * 32 + lr;
        const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<true>(p, acc[i][j][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)row * p.ldy + ch) = pk;
      }
    }
}

// The first convolution of the trunk: ONE input channel (the fbank map itself), 3x3, 32 output channels
// (resnet.py:96-99).  9 multiply-adds per output - nothing for a matrix core: f32 fma in tap order (bit-identical to the
// MFMA path, whose other 15 k-lanes are zeros).  A workgroup owns C1_SPAN consecutive rows: their inputs (the span plus
// C1_HALO rows on each side) go to LDS once as f32; a thread owns 8 output channels - their 9 x 8 weights and per-channel
// constants sit in registers - and walks C1_ROWS rows, a wave covering 16 consecutive rows x 4 channel chunks per step
// (1 KiB of contiguous 16-byte stores).  HBM bound on the 64 B it writes per row.  History (4.3 M rows): weights fetched per
// thread from global memory 600 us; weights from LDS per row 220 us (18 ds_read_b128 per 72 fma: LDS bound); weights in
// registers but the nine inputs of a row gathered from global memory inside the row loop 319 us (a dependent ~2 us load per
// step); inputs staged in LDS: see profiles/.
constexpr int C1_ROWS = 8;
constexpr int C1_HALO = 84;                      // >= pitch + 1 of the widest grid (ir.py: 82 frequency bins + 2)
template <bool GENERIC, int ET = ET_BF16>
__global__ __launch_bounds__(256) void grid_conv_c1_kernel(const TdnnKernelParams p) {
  __shared__ __attribute__((aligned(16))) float w_s[9 * 64];
  __shared__ __attribute__((aligned(16))) float c_s[3 * 64];                   // bias | scale | shift
  __shared__ float x_s[64 * C1_ROWS + 2 * C1_HALO];                             // span of the widest case (4 chunks: 64 rows per step)
  const int chunks = p.cout_store / 8;                                         // 4 (32 channels) .. 8
  const int rows_per_step = 256 / chunks, span = rows_per_step * C1_ROWS;
  const long long base = (long long)blockIdx.x * span;
  {
    const uint16_t *w = reinterpret_cast<const uint16_t *>(p.w);               // [cout_pad][n_taps][cin_pad] bf16
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
      const int t = i / 64, c = i % 64;
      w_s[i] = (t < p.n_taps && c < p.cout_store) ? h16_bits_to_f32<ET>(w[((size_t)c * p.n_taps + t) * p.cin_pad]) : 0.0f;
    }
    for (int c = threadIdx.x; c < 64; c += 256) {
      const bool ok = c < p.cout_store;
      c_s[c] = ok ? p.bias[c] : 0.0f;
      c_s[64 + c] = (ok && p.scale) ? p.scale[c] : 1.0f;
      c_s[128 + c] = (ok && p.shift) ? p.shift[c] : 0.0f;
    }
    const uint16_t *x = reinterpret_cast<const uint16_t *>(p.x);
    for (int i = threadIdx.x; i < span + 2 * C1_HALO; i += 256) {
      const long long r = base - C1_HALO + i;
      x_s[i] = (r >= 0 && r < p.rows) ? h16_bits_to_f32<ET>(x[(size_t)r * p.ldx]) : 0.0f;
    }
  }
  __syncthreads();
  const int slot = threadIdx.x / chunks, ch = (threadIdx.x % chunks) * 8;
  if (slot >= rows_per_step) return;                                           // chunks that do not divide 256 leave idle threads
  float wv[9][8], cb[8], cs[8], ct[8];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float4 w0 = *reinterpret_cast<const float4 *>(w_s + t * 64 + ch), w1 = *reinterpret_cast<const float4 *>(w_s + t * 64 + ch + 4);
    wv[t][0] = w0.x; wv[t][1] = w0.y; wv[t][2] = w0.z; wv[t][3] = w0.w; wv[t][4] = w1.x; wv[t][5] = w1.y; wv[t][6] = w1.z; wv[t][7] = w1.w;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { cb[e] = c_s[ch + e]; cs[e] = c_s[64 + ch + e]; ct[e] = c_s[128 + ch + e]; }
  int tap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) tap[t] = t < p.n_taps ? p.taps[t] : 0;          // unused taps carry zero weights
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll 2
  for (int k = 0; k < C1_ROWS; ++k) {
    const int local = k * rows_per_step + slot;
    const long long rl = base + local;
    if (rl >= p.rows) break;
    const int row = (int)rl;
    const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    if (valid) {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float xv = x_s[local + C1_HALO + tap[t]];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(wv[t][e], xv, acc[e]);
      }
    }
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (GENERIC) y[e] = tdnn_epilogue<true>(p, acc[e], row, ch + e, cb[e], cs[e], ct[e], valid);
      else y[e] = tdnn_epilogue_fast(acc[e], cb[e], act_lo, cs[e], ct[e], valid);
    }
    uint4 o;
    o.x = pack_h16x2<ET>(y[0], y[1]); o.y = pack_h16x2<ET>(y[2], y[3]); o.z = pack_h16x2<ET>(y[4], y[5]); o.w = pack_h16x2<ET>(y[6], y[7]);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)row * p.ldy + ch) = o;
  }
}

}  // namespace

// one launch, instantiated for the rows' element type (p.et: bf16, or IEEE half in the f16 precision mode)
#define ASV_CONV_ET(...) do { if (p.et == ET_F16) hipLaunchKernelGGL((__VA_ARGS__, ET_F16>), grid, block, 0, s, p); \
                              else hipLaunchKernelGGL((__VA_ARGS__, ET_BF16>), grid, block, 0, s, p); } while (0)
bool grid_conv_narrow_supported(const TdnnKernelParams &p, int et) {
  if (et == ET_F32 || p.n_taps != 9 || p.x2 != nullptr || p.wconv == nullptr) return false;
  if (p.cin_pad != 32 && p.cin_pad != 64) return false;
  if (p.cout_store != p.cin_pad) return false;                 // the trunk's 3x3 convolutions keep the channel count
  if (p.halo > CHALO || p.rows % CBM != 0 || p.ldx % 8 != 0 || p.ldy % 4 != 0) return false;
  return true;
}

// elements of the fragment-ordered weight copy for this kernel
size_t grid_conv_frag_elems(int cin_pad, int cout_pad32, int n_taps) { return (size_t)n_taps * (cin_pad / 16) * (cout_pad32 / 32) * 512; }

int launch_grid_conv_narrow(const TdnnKernelParams &p0, hipStream_t s) {
  TdnnKernelParams p = p0;
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  p.tune = live && getenv("ASV_AMD_CONV_ABL") != nullptr ? atoi(getenv("ASV_AMD_CONV_ABL")) : 0;
  ASV_REQUIRE(grid_conv_narrow_supported(p, true), "grid conv (narrow): unsupported layer");
  const dim3 grid(p.rows / CBM), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  if (p.cin_pad == 32) {
    if (fast) ASV_CONV_ET(grid_conv_narrow_kernel<32, 1, false);
    else ASV_CONV_ET(grid_conv_narrow_kernel<32, 1, true);
  } else {
    if (fast) ASV_CONV_ET(grid_conv_narrow_kernel<64, 2, false);
    else ASV_CONV_ET(grid_conv_narrow_kernel<64, 2, true);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

bool grid_conv_wide_supported(const TdnnKernelParams &p, int et) {
  if (et == ET_F32 || p.n_taps != 9 || p.x2 != nullptr || p.wconv == nullptr) return false;
  if (p.cin_pad != 128 && p.cin_pad != 256) return false;
  if (p.cout_store != p.cin_pad) return false;                 // the trunk's 3x3 convolutions keep the channel count
  const int halo_max = p.cin_pad == 128 ? WideGeom<128>::HALO : WideGeom<256>::HALO;
  const int bm = p.cin_pad == 128 ? WideGeom<128>::BM : WideGeom<256>::BM;
  if (p.halo > halo_max || p.rows % bm != 0 || p.ldx % 8 != 0 || p.ldy % 4 != 0) return false;
  return true;
}

int launch_grid_conv_wide(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(grid_conv_wide_supported(p, true), "grid conv (wide): unsupported layer");
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  if (p.cin_pad == 128) {
    const dim3 grid(p.rows / WideGeom<128>::BM), block(256);
    if (fast) ASV_CONV_ET(grid_conv_wide_kernel<128, false);
    else ASV_CONV_ET(grid_conv_wide_kernel<128, true);
  } else {
    const dim3 grid(p.rows / WideGeom<256>::BM), block(256);
    if (fast) ASV_CONV_ET(grid_conv_wide_kernel<256, false);
    else ASV_CONV_ET(grid_conv_wide_kernel<256, true);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

// the 32 -> 64 stride-2 convolution in space-to-depth form: 128 input channels, 4 backward taps, 64 output channels
bool grid_conv_s2d_supported(const TdnnKernelParams &p, int et) {
  if (et == ET_F32 || p.n_taps != S2D_NT || p.x2 != nullptr || p.wconv == nullptr) return false;
  if (p.cin_pad != 128 || p.cout_store != 64) return false;
  for (int t = 0; t < p.n_taps; ++t) if (p.taps[t] > 0 || p.taps[t] < -S2D_HLO) return false;
  if (p.rows % S2D_BM != 0 || p.ldx % 8 != 0 || p.ldy % 4 != 0) return false;
  return true;
}

int launch_grid_conv_s2d(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(grid_conv_s2d_supported(p, true), "grid conv (space to depth): unsupported layer");
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  const dim3 grid(p.rows / S2D_BM), block(256);
  if (fast) ASV_CONV_ET(grid_conv_s2d_kernel<false);
  else ASV_CONV_ET(grid_conv_s2d_kernel<true);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

bool grid_conv_c1_supported(const TdnnKernelParams &p, int et, int in_ch) {
  return et != ET_F32 && in_ch == 1 && p.x2 == nullptr && p.w != nullptr && p.cout_store % 8 == 0 && p.cout_store >= 32 && p.cout_store <= 64 && p.ldy % 8 == 0 &&
         p.n_taps <= 9 && p.halo <= C1_HALO;
}

int launch_grid_conv_c1(const TdnnKernelParams &p, hipStream_t s) {
  const int chunks = p.cout_store / 8, rows_per_wg = (256 / chunks) * C1_ROWS;
  const dim3 grid((unsigned)((p.rows + rows_per_wg - 1) / rows_per_wg)), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  if (fast) ASV_CONV_ET(grid_conv_c1_kernel<false);
  else ASV_CONV_ET(grid_conv_c1_kernel<true);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
