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
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int CBM = 256;              // output rows per workgroup
constexpr int CHALO = 88;             // window halo: >= pitch + 1 (<= 84), a multiple of 8
constexpr int CWIN = CBM + 2 * CHALO;  // 432 window rows

// Ablation bits (ASV_AMD_CONV_ABL; results are garbage) exist in the developer build only (libasv_amd_dev.so, `make dev`): in the
// product library this is the constant 0 and the ablated paths are not in the code.
#ifdef ASV_WITH_ABLATION
__device__ __forceinline__ int conv_abl(const TdnnKernelParams &p) { return p.tune; }
#else
__device__ __forceinline__ constexpr int conv_abl(const TdnnKernelParams &) { return 0; }
#endif

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
  const int abl = conv_abl(p);

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
          if constexpr (GENERIC) y[e] = tdnn_epilogue<ET>(p, acc[i][n][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
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
// The same convolution as a PERSISTENT kernel with a sliding window (round 3).  Ablations of the kernel above on the C = 32 layers
// (profiles/r3o_conv_abl.txt: 168 - 184 us as is, 112 - 121 without its stores, 159 - 176 with one tap instead of nine, 102 - 110
// with neither) say the matrix work is ~free and the load and store phases simply add up: a workgroup fetches its whole window,
// waits, computes, stores, ends - with three workgroups per CU on average ONE is fetching, 27 KiB in flight per CU against a
// ~2.5 us round trip, and every tile re-fetches 176 halo rows its neighbour already had (1.69 x read amplification).  Here a
// workgroup walks a contiguous run of tiles: the window lives in an LDS ring addressed by the global row index, the rows tile
// k + 1 adds (BM of them: amplification 1.0) are fetched while tile k is computed and stored, and the stores of tile k stay in
// flight across the barrier (the vector-memory counter retires in order: waiting for the older fetches leaves the younger stores
// outstanding).  Two tiles are fetched ahead; two workgroups per CU: C = 32: 256-row tiles, 960-row ring (60 KiB); C = 64: 128-row
// tiles, 576-row ring (72 KiB).
// Same (tap, k-group) accumulation order: bit-identical to the kernel above and to the generic tile.
template <int CIN> struct NarrowPers {
  static constexpr int WN = CIN / 32;                      // waves along the output channels (32 each): 1 | 2
  static constexpr int WM = 4 / WN;                        // waves along the rows (64 each): 4 | 2
  static constexpr int BM = WM * 64;                       // rows per tile: 256 | 128
  static constexpr int ROWB = CIN * 2, SLOTS = ROWB / 16, RPP = 1024 / ROWB;       // 16 | 8 rows per 1 KiB piece
  static constexpr int HALO = CIN == 32 ? 96 : 88;         // >= pitch + 1 (<= 84), a multiple of RPP
  static constexpr int WIN = BM + 2 * HALO;                // 448 | 304
  static constexpr int PF = 2;                             // tiles fetched ahead: what is in flight per CU is what bounds this kernel
  static constexpr int RING = WIN + PF * BM;               // 960 | 560 -> 576 rows
  static constexpr int RING_ROWS = (RING + RPP * 4 - 1) / (RPP * 4) * (RPP * 4);
  static constexpr int STAGE = 32 * 64;                    // per wave: 32 rows x 32 channels of output on their way to 16-byte stores
  static constexpr int LDS = RING_ROWS * ROWB + 4 * STAGE; // 69632 | 81920: two workgroups per CU
  static_assert(HALO % RPP == 0 && BM % RPP == 0 && RING_ROWS % RPP == 0 && (BM / RPP) % 4 == 0 && 2 * LDS <= 163840, "sliding-window geometry");
};

// A wave = 64 rows x 32 output channels with the layer's weights for those channels in registers for the whole run (C = 32: 18
// fragments, C = 64: 36 - the one-tile kernel streams the C = 64 layer's 72 KiB from L2 per wave and tile).  Epilogue: the 8-byte
// (row, 4 channels) pieces the accumulator layout yields go through a per-wave LDS tile and leave as 16-byte stores, 16 whole
// 64-byte row pieces per instruction (the direct 8-byte stores touch 32 rows with 16 bytes each: four times the address work).
template <int CIN, bool GENERIC, int ET = ET_BF16>
__global__ __launch_bounds__(256, 2) void grid_conv_narrow_pers_kernel(const TdnnKernelParams p, const int tiles_per_wg) {
  using G = NarrowPers<CIN>;
  constexpr int KG = CIN / 16, NF = CIN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  unsigned char *ring = lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  unsigned char *stage = lds + G::RING_ROWS * G::ROWB + wave * G::STAGE;
  const int lr = lane & 31, lh = lane >> 5;
  const int n_tiles = p.rows / G::BM;
  const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(t_begin + tiles_per_wg, n_tiles);
  if (t_begin >= t_end) return;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte_t *)ring);
  // `count` 1 KiB pieces of window rows starting at virtual row v0 (virtual row v = matrix row v - HALO, clamped onto the matrix:
  // its first / last rows are gaps); a piece's ring position is its virtual row modulo the ring
  auto fetch = [&](int v0, int count) {
    for (int j = wave; j < count; j += 4) {
      const int v = v0 + j * G::RPP;                             // first virtual row of the piece (wave-uniform)
      const int rbase = v % G::RING_ROWS;
      const int rrow = rbase + lane / G::SLOTS;                  // ring row of this lane's 16 bytes
      const int row = min(max(v - G::HALO + lane / G::SLOTS, 0), p.rows - 1);
      const int src_slot = cswz<CIN>(rrow, lane % G::SLOTS);
      conv_glds16(xg + (size_t)row * ((size_t)p.ldx * 2) + src_slot * 16, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)rbase * G::ROWB));
    }
  };
  fetch(t_begin * G::BM, (G::WIN + (min(t_end - t_begin, G::PF) - 1) * G::BM) / G::RPP);          // the first window + the rows of the next PF - 1 tiles

  const int v_taps = p.taps[lane < 9 ? lane : 0];
  const uint4 *wf = reinterpret_cast<const uint4 *>(p.wconv) + lane;            // fragment f at wf[f * 64]
  uint4 wr[9][KG];                                                              // this wave's 32 output channels, all taps
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) wr[t][kg] = wf[((t * KG + kg) * NF + wn) * 64];
  // the epilogue's constants for this lane's 16 channels (wn * 32 + 8 q + 4 lh + e)
  float cb[16], cs[16], ct[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ch = wn * 32 + 8 * q + 4 * lh;
    const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
    const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
    cb[q * 4] = b4.x; cb[q * 4 + 1] = b4.y; cb[q * 4 + 2] = b4.z; cb[q * 4 + 3] = b4.w;
    cs[q * 4] = sc4.x; cs[q * 4 + 1] = sc4.y; cs[q * 4 + 2] = sc4.z; cs[q * 4 + 3] = sc4.w;
    ct[q * 4] = sh4.x; ct[q * 4 + 1] = sh4.y; ct[q * 4 + 2] = sh4.z; ct[q * 4 + 3] = sh4.w;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int m0 = tile * G::BM;
    const int wbase = __builtin_amdgcn_readfirstlane(m0 % G::RING_ROWS);       // ring row of the window's first row (virtual row m0)
    const bool ahead = tile + G::PF < t_end;
    if (ahead) fetch(m0 + G::WIN + (G::PF - 1) * G::BM, G::BM / G::RPP);        // the rows tile + PF adds

    f32x16_t acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int d = __builtin_amdgcn_readlane(v_taps, t);
#pragma unroll
      for (int kg = 0; kg < KG; ++kg)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int w = wbase + G::HALO + wm * 64 + i * 32 + lr + d;
          w = w >= G::RING_ROWS ? w - G::RING_ROWS : w;
          const uint4 x = *reinterpret_cast<const uint4 *>(ring + w * G::ROWB + cswz<CIN>(w, kg * 2 + lh) * 16);
          acc[i] = mfma16<ET>(wr[t][kg], x, acc[i]);
        }
    }

    // ---- epilogue: acc[i][r] = row m0 + wm*64 + i*32 + lr, channel wn*32 + 8*(r>>2) + 4*lh + (r&3)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm * 64 + i * 32 + lr;
      const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<ET>(p, acc[i][q * 4 + e], row, wn * 32 + 8 * q + 4 * lh + e, cb[q * 4 + e], cs[q * 4 + e], ct[q * 4 + e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][q * 4 + e], cb[q * 4 + e], act_lo, cs[q * 4 + e], ct[q * 4 + e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        // staging tile: row lr, 16-byte slot q ^ ((lr >> 2) & 3), half lh
        *reinterpret_cast<uint2 *>(stage + lr * 64 + ((q ^ ((lr >> 2) & 3)) << 4) + lh * 8) = pk;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the wave's own writes (no other wave touches this tile)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int srow = h * 16 + (lane >> 2), slot = lane & 3;
        const uint4 v = *reinterpret_cast<const uint4 *>(stage + srow * 64 + ((slot ^ ((srow >> 2) & 3)) << 4));
        const int orow = m0 + wm * 64 + i * 32 + srow;
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)orow * p.ldy + wn * 32 + slot * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // reads done before the next fragment overwrites the tile
    }
    // every wave is through with window `tile` and the NEXT tile's rows have landed.  Younger than that fetch and free to stay in
    // flight (the counter retires in order): per further tile ahead a fetch (BM / RPP / 4 pieces per wave) and a tile's 4 stores,
    // and this tile's 4 stores.  Near the end of the run nothing was fetched in this iteration: only the own stores may remain.
    constexpr int kYounger = (G::PF - 1) * (G::BM / G::RPP / 4 + 4) + 4;
    if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYounger) : "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
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
    if (conv_abl(p) & 4) break;                                            // developer aid (ASV_AMD_CONV_ABL bit 2): no window fetch
    const int w = piece * G::RPP + lane / G::SLOTS;
    const int row = min(max(m0 - G::HALO + w, 0), p.rows - 1);
    const int src_slot = (lane % G::SLOTS) ^ (w & 15);
    conv_glds16(xg + (size_t)row * ((size_t)p.ldx * 2) + src_slot * 16, __builtin_amdgcn_readfirstlane(lds_base + piece * 1024));
  }
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  // The epilogue's per-channel constants, fetched NOW, one to three floats per thread (bias | scale | shift, CIN each): behind the K
  // loop they go through LDS to the lanes that need them.  (Until round 3 every (j, q) step of the epilogue fetched its three
  // float4 and waited: eight L2 round trips in a row per tile - with fetch, K loop and stores ablated the kernel still took 85 of
  // its 103 us, profiles/r3v_wide_abl.txt.)
  constexpr int NCST = 3 * CIN / 256 + (3 * CIN % 256 != 0);
  float cpre[NCST];
#pragma unroll
  for (int k = 0; k < NCST; ++k) {
    const int e = tid + 256 * k, which = e / CIN, c = e % CIN;
    cpre[k] = which == 0 ? 0.0f : (which == 1 ? 1.0f : 0.0f);
    if (e < 3 * CIN) {
      const float *src = which == 0 ? p.bias : (which == 1 ? p.scale : p.shift);
      if (src != nullptr) cpre[k] = src[c];
    }
  }
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
    if ((conv_abl(p) & 2) && st >= 2) break;                               // developer aid (ASV_AMD_CONV_ABL bit 1): two K steps instead of all
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

  // ---- epilogue: acc[i][j][r] = row m0 + wm*128 + i*32 + lr, channel wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3).
  // The window is dead: its first 16 KiB become four per-wave tiles (32 rows x 64 channels) through which the 8-byte pieces of the
  // accumulator layout turn into 16-byte stores of whole 128-byte row pieces, the 3 x CIN constants sit behind them.
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  __builtin_amdgcn_s_barrier();                                       // every wave is through with the window
  asm volatile("" ::: "memory");
  float *cst = reinterpret_cast<float *>(win + 16384);
#pragma unroll
  for (int k = 0; k < NCST; ++k)
    if (tid + 256 * k < 3 * CIN) cst[tid + 256 * k] = cpre[k];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned char *stage = win + wave * 4096;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wm * 128 + i * 32 + lr;
    const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = wn * 64 + j * 32 + 8 * q + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4 *>(cst + ch);
        const float4 sc4 = *reinterpret_cast<const float4 *>(cst + CIN + ch);
        const float4 sh4 = *reinterpret_cast<const float4 *>(cst + 2 * CIN + ch);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<ET>(p, acc[i][j][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        // tile row lr (128 bytes), 16-byte slot (j * 4 + q) ^ (lr & 7), half lh
        *reinterpret_cast<uint2 *>(stage + lr * 128 + (((j * 4 + q) ^ (lr & 7)) << 4) + lh * 8) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the wave's own writes (no other wave touches this tile)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int srow = h * 8 + (lane >> 3), slot = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(stage + srow * 128 + ((slot ^ (srow & 7)) << 4));
      const int orow = m0 + wm * 128 + i * 32 + srow;
      *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)orow * p.ldy + wn * 64 + slot * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // reads done before the next fragment overwrites the tile
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
  // bias | scale | shift of the 64 output channels: fetched now by the first 192 threads, handed out through LDS behind the K loop
  float cpre = tid < 128 ? (tid < 64 ? 0.0f : 1.0f) : 0.0f;
  if (tid < 192) {
    const float *src = tid < 64 ? p.bias : (tid < 128 ? p.scale : p.shift);
    if (src != nullptr) cpre = src[tid & 63];
  }
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

  // ---- epilogue: acc[i][j][r] = row m0 + wave*64 + i*32 + lr, channel j*32 + 8*(r>>2) + 4*lh + (r&3); through per-wave LDS tiles
  // in the dead window to 16-byte stores of whole 128-byte rows (see grid_conv_wide_kernel)
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  __builtin_amdgcn_s_barrier();                                       // every wave is through with the window
  asm volatile("" ::: "memory");
  float *cst = reinterpret_cast<float *>(win + 16384);
  if (tid < 192) cst[tid] = cpre;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  unsigned char *stage = win + wave * 4096;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = m0 + wave * 64 + i * 32 + lr;
    const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = j * 32 + 8 * q + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4 *>(cst + ch);
        const float4 sc4 = *reinterpret_cast<const float4 *>(cst + 64 + ch);
        const float4 sh4 = *reinterpret_cast<const float4 *>(cst + 128 + ch);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) y[e] = tdnn_epilogue<ET>(p, acc[i][j][q * 4 + e], row, ch + e, b[e], sc[e], sh[e], valid);
          else y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        *reinterpret_cast<uint2 *>(stage + lr * 128 + (((j * 4 + q) ^ (lr & 7)) << 4) + lh * 8) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int srow = h * 8 + (lane >> 3), slot = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4 *>(stage + srow * 128 + ((slot ^ (srow & 7)) << 4));
      const int orow = m0 + wave * 64 + i * 32 + srow;
      *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)orow * p.ldy + slot * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    // weights [cout_pad][n_taps][cin_pad] in the rows' element type (f32 in the parity modes: the same exact fma chain)
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
      const int t = i / 64, c = i % 64;
      w_s[i] = (t < p.n_taps && c < p.cout_store) ? load_elem<ET>(p.w, ((size_t)c * p.n_taps + t) * p.cin_pad) : 0.0f;
    }
    for (int c = threadIdx.x; c < 64; c += 256) {
      const bool ok = c < p.cout_store;
      c_s[c] = ok ? p.bias[c] : 0.0f;
      c_s[64 + c] = (ok && p.scale) ? p.scale[c] : 1.0f;
      c_s[128 + c] = (ok && p.shift) ? p.shift[c] : 0.0f;
    }
    for (int i = threadIdx.x; i < span + 2 * C1_HALO; i += 256) {
      const long long r = base - C1_HALO + i;
      x_s[i] = (r >= 0 && r < p.rows) ? load_elem<ET>(p.x, (size_t)r * p.ldx) : 0.0f;
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
      if constexpr (GENERIC) y[e] = tdnn_epilogue<ET>(p, acc[e], row, ch + e, cb[e], cs[e], ct[e], valid);
      else y[e] = tdnn_epilogue_fast(acc[e], cb[e], act_lo, cs[e], ct[e], valid);
    }
    if constexpr (ET == ET_F32) {
      float *yo = reinterpret_cast<float *>(p.y) + (size_t)row * p.ldy + ch;
      *reinterpret_cast<float4 *>(yo) = make_float4(y[0], y[1], y[2], y[3]);
      *reinterpret_cast<float4 *>(yo + 4) = make_float4(y[4], y[5], y[6], y[7]);
    } else {
      uint4 o;
      o.x = pack_h16x2<ET>(y[0], y[1]); o.y = pack_h16x2<ET>(y[2], y[3]); o.z = pack_h16x2<ET>(y[4], y[5]); o.w = pack_h16x2<ET>(y[6], y[7]);
      *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.y) + (size_t)row * p.ldy + ch) = o;
    }
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
  p.tune = 0;
#ifdef ASV_WITH_ABLATION
  p.tune = live && getenv("ASV_AMD_CONV_ABL") != nullptr ? atoi(getenv("ASV_AMD_CONV_ABL")) : 0;
#endif
  ASV_REQUIRE(grid_conv_narrow_supported(p, true), "grid conv (narrow): unsupported layer");
  const dim3 grid(p.rows / CBM), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  static const int pers0 = getenv("ASV_AMD_CONV_PERS") != nullptr ? atoi(getenv("ASV_AMD_CONV_PERS")) : 1;
  const int pers = live && getenv("ASV_AMD_CONV_PERS") != nullptr ? atoi(getenv("ASV_AMD_CONV_PERS")) : pers0;
  if (pers && p.tune == 0 && p.ldy % 8 == 0) {                 // (its 16-byte stores)
    // persistent sliding-window form: every workgroup a contiguous run of tiles; as many workgroups as the chip holds at once
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int bm = p.cin_pad == 32 ? NarrowPers<32>::BM : NarrowPers<64>::BM, per_cu = 2;
    const int n_tiles = p.rows / bm, wgs = std::min(n_tiles, cus * per_cu), per_wg = (n_tiles + wgs - 1) / wgs;
    const dim3 pgrid((n_tiles + per_wg - 1) / per_wg);
#define ASV_CONV_PERS(...) do { if (p.et == ET_F16) hipLaunchKernelGGL((__VA_ARGS__, ET_F16>), pgrid, block, 0, s, p, per_wg); \
                                else hipLaunchKernelGGL((__VA_ARGS__, ET_BF16>), pgrid, block, 0, s, p, per_wg); } while (0)
    if (p.cin_pad == 32) { if (fast) ASV_CONV_PERS(grid_conv_narrow_pers_kernel<32, false); else ASV_CONV_PERS(grid_conv_narrow_pers_kernel<32, true); }
    else { if (fast) ASV_CONV_PERS(grid_conv_narrow_pers_kernel<64, false); else ASV_CONV_PERS(grid_conv_narrow_pers_kernel<64, true); }
#undef ASV_CONV_PERS
    ASV_HIP_CHECK(hipGetLastError());
    return ASV_OK;
  }
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
  if (p.halo > halo_max || p.rows % bm != 0 || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;      // (16-byte stores)
  return true;
}

int launch_grid_conv_wide(const TdnnKernelParams &p0, hipStream_t s) {
  TdnnKernelParams p = p0;
  p.tune = 0;
#ifdef ASV_WITH_ABLATION
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  p.tune = live && getenv("ASV_AMD_CONV_ABL") != nullptr ? atoi(getenv("ASV_AMD_CONV_ABL")) : 0;
#endif
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
  if (p.rows % S2D_BM != 0 || p.ldx % 8 != 0 || p.ldy % 8 != 0) return false;
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
  return in_ch == 1 && p.x2 == nullptr && p.w != nullptr && p.cout_store % 8 == 0 && p.cout_store >= 32 && p.cout_store <= 64 && p.ldy % 8 == 0 &&
         p.n_taps <= 9 && p.halo <= C1_HALO;
}

int launch_grid_conv_c1(const TdnnKernelParams &p, hipStream_t s) {
  const int chunks = p.cout_store / 8, rows_per_wg = (256 / chunks) * C1_ROWS;
  const dim3 grid((unsigned)((p.rows + rows_per_wg - 1) / rows_per_wg)), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  if (p.et == ET_F32) {                              // the parity modes (f32 / f32x): f32 rows and weights, the same fma chain
    if (fast) hipLaunchKernelGGL((grid_conv_c1_kernel<false, ET_F32>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((grid_conv_c1_kernel<true, ET_F32>), grid, block, 0, s, p);
  } else if (fast) ASV_CONV_ET(grid_conv_c1_kernel<false);
  else ASV_CONV_ET(grid_conv_c1_kernel<true);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
