// Back-to-back TDNN layers of one 128-frame tile in ONE kernel: layer A (any taps, -> 512 channels), up to two
// 1-tap 512 -> 512 layers, and a last 1-tap layer (-> any width) whose output goes straight into the fused
// statistics pooling.  For the standard x-vector this is tdnn3 -> tdnn4 -> tdnn5 -> StatisticsPooling
// (model/xvector.py:77-98; components.py:107-149, 410-431; pooling.py:58-67): 65 % of the network's FLOPs, of which
// nothing but the per-tile pooling moments reaches HBM.
//
// Why: a 1-tap 512 -> 512 layer is a K = 512 GEMM - per 128 x 256 tile a 14 us main loop under 5.6 us of prologue +
// epilogue, with a workgroup barrier every 64 channels for the LDS window ring (profiles/r1_*: 0.32 of the bf16 MFMA peak
// against 0.43 for the 3-tap layers).  Here the 128 x 512 bf16 output tile of a layer stays in LDS (128 KiB, "Y") and is
// the B operand of the next layer's MFMAs directly:
//   * phase 1 (layer A): kernels_tdnn_v3.hip's loop - feature window through a 4-stage LDS-DMA ring (which lives inside
//     the not yet used Y region), weight fragments from L2 - with 8 waves, each 128 frames x 64 channels; epilogue
//     (bias, ReLU, folded BN, bf16) writes Y;
//   * middle layers: K = 512 comes from Y - no LDS-DMA, no ring, NO barrier in the main loop; one barrier before the
//     epilogue overwrites Y in place;
//   * last layer: the output channels are cut into 64-channel units, unit u = pass * 8 + wave; a wave runs its units
//     back to back without ever meeting the others, so the matrix work of one wave of a SIMD overlaps the pooling
//     epilogue of its partner.  Its MFMAs take the operands in the other order (D = X W^T): the accumulator then has
//     lane = channel, registers = frames, and the pooling epilogue sums pivoted moments per utterance inside a lane -
//     no LDS transposition (the first version went through a scratch tile like the POOL epilogue of kernels_tdnn_v3.hip:
//     15.6k instead of 11.5k cycles per unit); pool_finish_kernel merges the tiles.
// Measured (640 x 200 frames): 385 us (profiles/r2_*) - 405 us (profiles/r2final_*, another box) against 190 + 84 + 253 us for
// the three launches it replaces (433 us before the run-based pooling epilogue, profiles/r2a_*).  Per tile
// (s_memtime stamps, ASV_AMD_CHAIN_DBG=1): layer A 57.7k cycles for 49.2k of MFMA issue, middle layer 18.3k (16.4k), last
// layer 3 x (11.6k loop + 12k epilogue next to the partner's loop); the shader clock inside the kernel is 1.76 GHz.
// Tried without effect (in-process A/B, tools/chain_ab.py): s_setprio around either phase, four partial sums instead of
// the 16-deep chains, 22 instead of 48 instructions per k-group (one address register per step): the loops sit at the
// matrix pipe's rate, the epilogue at ~14 cycles per VALU operation beside a streaming partner.
// Where the last layer's time goes (ASV_AMD_CHAIN_DBG=2, profiles/r2t_chain_timeline.txt; tools/coissue_probe2.hip,
// profiles/r2r_coissue2.txt): the two waves of a SIMD do alternate loop / epilogue, a loop takes 10.3k cycles with or without a
// computing partner, an epilogue 12 - 15k next to a partner's loop and 4.0k alone - the epilogue sets the period.  Its packed f32
// operations only issue in the gaps of the partner's MFMA stream, plain ones would at ~8 cycles each (twice as many: no gain,
// profiles/r2s_lib_ab_packed_fp32.txt); behind a wave's OWN MFMAs a plain VALU operation costs ~0.5 cycle.  Next step (LABLOG.md
// section 8): the pooling arithmetic of unit u interleaved into the MFMA loop of unit u + 1 of the same wave.
// Tried and dropped at the end of round 2 (profiles/r2v_ab_chain_prefetch.txt): bias / BN scale of the NEXT unit loaded before a
// unit's K loop and the next unit's first weight fragments fetched by the last K step (no memory round trip at the start of a
// unit or of its epilogue): bit-identical, 703.6 -> 702.5 us per step - those waits are not what stretches the phase.
// Likewise (profiles/r2w_ab_chain_pin.txt): hipcc orders the eight fragment loads ahead of a Y-fed K loop by base address
// (a0 a1 a2 a3 b1 b2 b0 b3), and because the loop's waits must also hold on the path from its entry, the wait in front of b0's
// first use is vmcnt(2) where the steady state needs vmcnt(7) - every K step waits for fragments fetched 8 MFMAs earlier.  With the
// entry loads pinned to the loop's order all waits become vmcnt(7): bit-identical, 705.8 -> 708.4 us per step (L2 hits return
// within those 256 cycles) - dropped.
// What bounds the K loops (round 6, experiment builds -DCHAIN_EXP below, profiles/r6z_chain_fetch_experiments.txt; csrc/tools/vmem_probe.hip,
// profiles/r6z_vmem_probe.txt): every layer's loop runs at 79 % (20.6 k ticks per 16 steps against 16.4 k at 32 per matrix instruction).  With
// one of the two weight-fragment loads per k-group taken away the same loop runs at 16.6 - 16.8 k; with all loads but from one address (L1
// hits) at 20.1 k.  It is what the operand fetches cost the SIMD that issues them, whatever level serves them: a bare probe with this
// kernel's exact mix - per 16 matrix instructions 4 fragment loads of 1 KiB, spread and two chunks ahead, + 16 ds_read_b128 - reaches 71 % of
// the datasheet rate on trivial operands (the loads alone 87 %, the LDS reads alone 76 %, no fetches 98 %); these loops run at ~67 % in wall-clock
// terms.  0.75 fetches per matrix instruction is what 128 accumulator registers per wave allow whatever the wave's tile shape; more frames
// per resident tile would cut the loads (128 x 512 x 16 bit IS the LDS).
// One workgroup (512 threads, 160 KiB LDS) per CU.
#include <cstdlib>

#include "device_utils.h"

#ifndef CHAIN_EXP
#define CHAIN_EXP 0      // experiment builds: bit 0 / 1 = the Y loops fetch only one / none of the two weight fragments per k-group, bit 2 = every chunk fetches the SAME fragments (results garbage)
#endif

namespace asv {
namespace {

constexpr int CN = kChainWidth;           // channels of the resident tile (512)
constexpr int CBK = 64;
constexpr int CROWB = 128;                // window row: 64 bf16
constexpr int CSTAGES = 4;
constexpr int YROWB = CN * 2;             // 1024 B per Y row
// MF = 32-frame fragments per tile: 4 (128 frames, the default), 3 or 2 (96 / 64 frames: batches that do not fill one round of the
// chip's CUs run in the smallest tile that still fits one round - more CUs, each for 3/4 or 1/2 of the time; chain_tile_plan)
template <int MF> struct ChainGeom {
  static constexpr int CM = MF * 32;                  // frames per workgroup
  static constexpr int CWIN = CM + 2 * kHalo;         // 136 | 104 | 72
  static constexpr int CSTAGE = CWIN * CROWB;         // 17408 | 13312 B
  static constexpr int CGROUPS = CWIN / 8;            // 17 | 13 eight-row DMA pieces
  static constexpr int CPIECES = (CGROUPS + 7) / 8;   // 3 | 2 per wave
  static constexpr int Y_BYTES = CM * YROWB;          // 131072 | 98304
  static constexpr int SCR_OFF = Y_BYTES;             // 32 KiB: epilogue constants (phases 1, 2) | 8 x 4 KiB pooling scratch (last phase)
  static constexpr int LDS = Y_BYTES + 32768;
  static_assert(CSTAGES * CSTAGE <= Y_BYTES, "the window ring lives inside the Y region");
  static_assert(LDS <= 163840, "160 KiB of LDS per CU");
};

typedef __attribute__((address_space(3))) unsigned char chain_lds_byte;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct TrNo { static constexpr bool value = false; };
struct TrYes { static constexpr bool value = true; };

__device__ __forceinline__ int cswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void chain_glds16_s(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

// POOLV: pooling epilogue of the last layer.  0 = first version (per-lane segment tracking, register-by-register seams);
//        1 = run-based (default): every utterance inside a 32-frame fragment is one masked run over all 16 registers,
//            packed f32 arithmetic (v_pk_add_f32 / v_pk_fma_f32).  ASV_AMD_CHAIN_POOLV selects at launch (A/B aid).
// ABL (developer aid: instantiated in the developer build libasv_amd_dev.so only, ASV_AMD_CHAIN_ABL; results are garbage): 1 = the last layer without its pooling
//        epilogue (what do the K loops cost on their own), 2 = the epilogue's arithmetic without its global loads / stores;
//        3 (ASV_AMD_CHAIN_DBG >= 3; results valid) = the production kernel + stamps inside every wave's first pooling epilogue;
//        4 (results valid) = the last layer in lockstep: a workgroup barrier behind every unit's K loop and behind every pooling
//        epilogue, so that the two waves of a SIMD run their loops together (sharing the matrix pipe at its full rate) and their
//        epilogues together (no matrix stream beside the packed f32 arithmetic, which otherwise only issues in its gaps).
template <int POOLV, int ET = ET_BF16, int ABL = 0, int MF = 4>
__global__ __launch_bounds__(512, 2) void tdnn_chain_kernel(const TdnnChainParams p) {
  using Geo = ChainGeom<MF>;
  constexpr int CM = Geo::CM, CSTAGE = Geo::CSTAGE, CGROUPS = Geo::CGROUPS, CPIECES = Geo::CPIECES, SCR_OFF = Geo::SCR_OFF;
  __shared__ __attribute__((aligned(16))) unsigned char lds[Geo::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..7: channel slice of phases 1-2, unit index of the last
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = p.row_base + blockIdx.x * CM;                      // (row_base / tile_base: the launch's first row and tile, launch_tdnn_chain)
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(chain_lds_byte *)lds);
  float *par = reinterpret_cast<float *>(lds + SCR_OFF);            // bias[512] | scale[512] | shift[512] of the layer in flight

  // developer aid: [workgroup][wave][32] s_memtime stamps; 0..12 phase boundaries, 14 / 15 s_memrealtime at start / end,
  // 16..23 (dbg_fine) inside the pooling epilogue of the wave's first unit: before the fragments, after each of the four, after
  // the final publish
  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.dbg != nullptr && lane == 0 && n_stamp < 14) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + n_stamp] = __builtin_amdgcn_s_memtime();
    ++n_stamp;
  };
  auto fine = [&](int k) {
    if constexpr (ABL == 3) if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 16 + k] = __builtin_amdgcn_s_memtime();
  };
  stamp();                                                       // 0: start
  if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 14] = __builtin_amdgcn_s_memrealtime();
  auto stage_params = [&](const TdnnChainLayer &L) {
    if (tid < 384) {
      const int which = tid >> 7, idx = (tid & 127) * 4;
      float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = (which == 0) ? L.bias : (which == 1 ? L.scale : L.shift);
      if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + idx);
      *reinterpret_cast<float4 *>(par + which * CN + idx) = v;
    }
  };

  uint4 wf[4][2];
  f32x16_t acc[MF][2];
  struct XFrags { uint4 x[MF]; };
  // the accumulators start from the bias (one v_mov per register either way; saves the add in every epilogue):
  // acc[i][j][4 q + e] belongs to channel j * 32 + 8 q + 4 lh + e of the wave's 64-channel slice, for every frame fragment i
  auto init_acc = [&](const float *bias64) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4 *>(bias64 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          acc[i][j][q * 4 + 0] = b4.x; acc[i][j][q * 4 + 1] = b4.y; acc[i][j][q * 4 + 2] = b4.z; acc[i][j][q * 4 + 3] = b4.w;
        }
      }
  };
  // TR = false: D = W X^T, accumulator lane = frame, registers = channels (4 consecutive channels per lane and row: what
  //              the row-major Y store wants);
  // TR = true:  D = X W^T, accumulator lane = CHANNEL (column j*32 + lr), registers = frames 8 (r >> 2) + 4 lh + (r & 3) of
  //              fragment i - the pooling epilogue then sums over frames inside a lane, without any transposition.
  auto mma2 = [&](const XFrags &f, int kg, int j, int i0, auto tr) {
#pragma unroll
    for (int i = i0; i < i0 + 2 && i < MF; ++i) {
      if constexpr (decltype(tr)::value) acc[i][j] = mfma16<ET>(f.x[i], wf[kg][j], acc[i][j]);
      else acc[i][j] = mfma16<ET>(wf[kg][j], f.x[i], acc[i][j]);
    }
  };


  // ================================ phase 1: layer A through the window ring ================================
  stage_params(p.first);
  {
    const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
    const size_t x_pitch = (size_t)p.ldx * 2;
    const int g_row = lane >> 3, g_slot = lane & 7;
    const int nchunks = p.cin_pad / CBK;
    const int n_taps = p.n_taps;
    uint32_t a_off[CPIECES];
#pragma unroll
    for (int i = 0; i < CPIECES; ++i) {
      const int grp = min(wave + i * 8, CGROUPS - 1);
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      a_off[i] = (uint32_t)row * (uint32_t)x_pitch + (uint32_t)cswz(w, g_slot) * 16u;
    }
    auto issue_A = [&](int c, int st) {
      const unsigned char *base = xg + (size_t)c * (CBK * 2);
#pragma unroll
      for (int i = 0; i < CPIECES; ++i) {
        const int grp = min(wave + i * 8, CGROUPS - 1);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + st * CSTAGE + grp * 1024);
        chain_glds16_s(base, a_off[i], dst);
      }
    };
    // Instruction diet (the SIMD's issue slots are what two co-resident waves really share - a wave in an epilogue gets
    // what the partner's main loop leaves): the four frame fragments of a k-group sit 32 window rows = 4096 B apart and
    // share their swizzle term ((row >> 1) & 7 is blind to +32), so ONE address register per (chunk, tap) step serves all
    // of them through immediate offsets, and the k-group enters as an XOR of bits 5-6; weight fragments use a scalar
    // base + the constant lane offset.
    const size_t frag_stride = (size_t)n_taps * nchunks * 4096;
    const unsigned char *wA0 = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 2) * frag_stride;   // wave-uniform
    const unsigned char *wA1 = wA0 + frag_stride;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto x_base = [&](int c, int d) -> uint32_t {
      const int wrow = lr + kHalo + d;
      return (uint32_t)((c % CSTAGES) * CSTAGE + wrow * CROWB + ((lh ^ ((wrow >> 1) & 7)) << 4));
    };
    auto load_x4 = [&](uint32_t xb, int kg, int i, XFrags &f) {
      f.x[i] = *reinterpret_cast<const uint4 *>(lds + (xb ^ (uint32_t)(kg << 5)) + i * 4096);
    };
    issue_A(0, 0);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      wf[kg][0] = *reinterpret_cast<const uint4 *>(wA0 + (size_t)kg * 1024 + lane16);
      wf[kg][1] = *reinterpret_cast<const uint4 *>(wA1 + (size_t)kg * 1024 + lane16);
    }
    if (nchunks > 2) {
      issue_A(1, 1);
      issue_A(2, 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CPIECES) : "memory");
    } else {
      if (nchunks > 1) issue_A(1, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();                                                     // 1: first window + fragments in
    init_acc(p.first.bias + wave * 64);
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    const int d_first = __builtin_amdgcn_readlane(v_taps, 0);
    XFrags x0, x1;
    uint32_t xb = x_base(0, d_first);
#pragma unroll
    for (int i = 0; i < MF; ++i) load_x4(xb, 0, i, x0);
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      for (int t = 0; t < n_taps; ++t) {
        const bool last_tap = (t + 1 == n_taps);
        int cn = c, tn = t + 1;
        if (last_tap) { tn = 0; cn = c + 1; }
        const bool more = cn < nchunks;
        if (!more) { cn = c; tn = t; }                       // the last step re-fetches its own fragments (never used)
        const size_t wnext = ((size_t)tn * nchunks + cn) * 4096;
        auto group = [&](const XFrags &xc, int kg, XFrags &xn, uint32_t xbn, int kgn) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q < MF) load_x4(xbn, kgn, q, xn);
            mma2(xc, kg, q / 2, (q % 2) * 2, TrNo{});
            if (q == 1) wf[kg][0] = *reinterpret_cast<const uint4 *>(wA0 + wnext + (size_t)kg * 1024 + lane16);
            if (q == 3) wf[kg][1] = *reinterpret_cast<const uint4 *>(wA1 + wnext + (size_t)kg * 1024 + lane16);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        group(x0, 0, x1, xb, 1);
        group(x1, 1, x0, xb, 2);
        group(x0, 2, x1, xb, 3);
        if (last_tap && c + 1 < nchunks) {
          // see kernels_tdnn_v3.hip: the youngest 8 VMEM operations are fragment fetches; window c+1 is older
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          if (c + 3 < nchunks) issue_A(c + 3, (c + 3) % CSTAGES);
        }
        const uint32_t xbn = x_base(cn, __builtin_amdgcn_readlane(v_taps, tn));   // next step (the last one: itself, harmless)
        group(x1, 3, x0, xbn, 0);
        xb = xbn;
      }
    }
  }

  // epilogue of a 512-wide layer: [ReLU], folded BN, 16-bit -> Y (row-major, 16-byte slots XOR-swizzled by row & 15); the bias
  // is in the accumulators already.  `affine` = false: the layer's eval BatchNorm was folded into the NEXT layer's weights and
  // bias on the host (exact for a 1-tap consumer: W (s u + t) + b = (W diag s) u + (W t + b); runtime.hip asv_net_finalize) -
  // what is stored is ReLU(acc) alone: one conversion per pair and the ReLU as ONE integer maximum on the packed pair
  // (relu_h16x2) instead of a maximum and a multiply-add per value: 6 instead of 10 VALU operations per four values, in a
  // phase in which no wave of the workgroup has matrix work to hide them behind.
  auto store_Y = [&](int relu, bool affine) {
    const float act_lo = relu ? 0.0f : -INFINITY;
    unsigned char *yrow = lds + lr * YROWB + lh * 8;
    const int rx = lr & 15;
    if (!affine) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned char *dst = yrow + (((wave * 8 + j * 4 + q) ^ rx) << 4);
#pragma unroll
          for (int i = 0; i < MF; ++i) {
            uint2 pk;
            pk.x = pack_h16x2<ET>(acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1]);
            pk.y = pack_h16x2<ET>(acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]);
            if (relu) { pk.x = relu_h16x2(pk.x); pk.y = relu_h16x2(pk.y); }
            *reinterpret_cast<uint2 *>(dst + i * 32 * YROWB) = pk;
          }
        }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chl = wave * 64 + j * 32 + 8 * q + 4 * lh;
        const float4 sc4 = *reinterpret_cast<const float4 *>(par + CN + chl);
        const float4 sh4 = *reinterpret_cast<const float4 *>(par + 2 * CN + chl);
        const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        unsigned char *dst = yrow + (((wave * 8 + j * 4 + q) ^ rx) << 4);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = fmaf(max_lo(acc[i][j][q * 4 + e], act_lo), sc[e], sh[e]);
          uint2 pk;
          pk.x = pack_h16x2<ET>(y[0], y[1]);
          pk.y = pack_h16x2<ET>(y[2], y[3]);
          *reinterpret_cast<uint2 *>(dst + i * 32 * YROWB) = pk;
        }
      }
  };

  // main loop of a layer whose input is Y: K = 512 = 8 steps of 4 k-groups, fragments of the next step prefetched, no barrier.
  // wb0 / wb1: wave-uniform fragment bases; bias64: the 64 biases of this wave's output channels (global)
  auto yloop = [&](const unsigned char *wb0, const unsigned char *wb1, const float *bias64, auto tr) {
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t yb = (uint32_t)(lr * YROWB);                  // fragment i: + i * 32 KiB (immediate offsets)
    const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
    auto load_y4 = [&](int c, int kg, int i, XFrags &f) {       // slot (c*8 + kg*2 + lh) ^ (lr & 15) = (c*8 + kg*2) ^ sx
      f.x[i] = *reinterpret_cast<const uint4 *>(lds + yb + ((((uint32_t)(c * 8 + kg * 2)) ^ sx) << 4) + i * 32 * YROWB);
    };
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      wf[kg][0] = *reinterpret_cast<const uint4 *>(wb0 + (size_t)kg * 1024 + lane16);
      wf[kg][1] = *reinterpret_cast<const uint4 *>(wb1 + (size_t)kg * 1024 + lane16);
    }
    if constexpr (decltype(tr)::value) {                        // lane = channel: one bias per lane and channel fragment
      const float b0 = bias64[lr], b1 = bias64[32 + lr];
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    } else {
      init_acc(bias64);
    }
    XFrags x0, x1;
#pragma unroll
    for (int i = 0; i < MF; ++i) load_y4(0, 0, i, x0);
#pragma unroll 1
    for (int c = 0; c < CN / CBK; ++c) {
      const int cn = min(c + 1, CN / CBK - 1);
      const size_t wnext = (CHAIN_EXP & 4) ? 0 : (size_t)cn * 4096;
      auto group = [&](const XFrags &xc, int kg, XFrags &xn, int c2, int kgn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < MF) load_y4(c2, kgn, q, xn);
          mma2(xc, kg, q / 2, (q % 2) * 2, tr);
          if (q == 1 && !(CHAIN_EXP & 2)) wf[kg][0] = *reinterpret_cast<const uint4 *>(wb0 + wnext + (size_t)kg * 1024 + lane16);
          if (q == 3 && !(CHAIN_EXP & 1)) wf[kg][1] = *reinterpret_cast<const uint4 *>(wb1 + wnext + (size_t)kg * 1024 + lane16);      // (CHAIN_EXP: experiment builds only, never the product - profiles/r6z_*)
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      group(x0, 0, x1, c, 1);
      group(x1, 1, x0, c, 2);
      group(x0, 2, x1, c, 3);
      group(x1, 3, x0, cn, 0);
    }
  };

  stamp();                                 // 2: main loop of layer A done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done with the ring: Y may be written
  asm volatile("" ::: "memory");
  stamp();                                 // 3
  store_Y(p.first.relu, p.first.scale != nullptr);
  __builtin_amdgcn_s_barrier();            // Y complete; the constants of layer A are dead
  asm volatile("" ::: "memory");
  stamp();                                 // 4: Y of layer A complete

  // ================================ middle layers: Y -> Y ================================
#pragma unroll 1
  for (int m = 0; m < p.n_mid; ++m) {
    const TdnnChainLayer &L = p.mid[m];
    stage_params(L);
    const size_t frag_stride = (size_t)(CN / CBK) * 4096;
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(wave * 2) * frag_stride;
    yloop(wb, wb + frag_stride, L.bias + wave * 64, TrNo{});
    stamp();                               // 5: main loop of the middle layer done
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody reads the old Y any more (and the staged constants are visible)
    asm volatile("" ::: "memory");
    store_Y(L.relu, L.scale != nullptr);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();                               // 6: Y of the middle layer complete
  }

  // ================================ last layer + fused statistics pooling ================================
  {
    const TdnnChainLayer &L = p.last;
    const float act_lo = L.relu ? 0.0f : -INFINITY;
    const int half = p.tile_base + (int)blockIdx.x;      // index of this tile's partial-moment block
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && m0 + k < p.rows) first_seg = p.row_seg[m0 + k];
    // row -> utterance of the tile's rows (-1: gap row; rows beyond the tile - the 96-frame form - or beyond the matrix count as gaps)
    const int rowseg_lo = (m0 + lane < p.rows) ? p.row_seg[m0 + lane] : -1;
    const int rowseg_hi = (64 + lane < CM && m0 + 64 + lane < p.rows) ? p.row_seg[m0 + 64 + lane] : -1;
    const size_t frag_stride = (size_t)(CN / CBK) * 4096;
#pragma unroll 1
    for (int cb = wave * 64; cb < L.cout_pad; cb += 512) {
      const unsigned char *wb = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb / 32) * frag_stride;
      yloop(wb, wb + frag_stride, L.bias + cb, TrYes{});
      if constexpr (ABL == 4) __builtin_amdgcn_s_barrier();
      stamp();                             // 7, 9, 11: main loop of a unit done
      // Pooling epilogue, registers only.  acc[i][j][r] = channel cb + j*32 + lr, frame i*32 + 8 (r >> 2) + 4 lh + (r & 3): a
      // lane sums its own frames (those with bit 2 of the row index == lh) per utterance, about the pivot of its first
      // frame; the two lane halves publish separate partials
      //   P[tile][segment slot][lh][3 = sum (u - pv), sum (u - pv)^2, pv][channel]
      // which pool_finish_kernel merges (Chan et al.), knowing how many frames each half holds.  The BN scale multiplies the
      // three moments at publication (u = scale * act(acc): moments about a pivot are linear / quadratic in it); the BN
      // shift is added to the mean by pool_finish.
      if constexpr (ABL == 1) {                // no pooling epilogue at all: the accumulators only have to stay alive up to here
#if defined(__HIP_DEVICE_COMPILE__)      // (a "v" constraint means nothing to the host pass: with a template-dependent bound it drops the kernel's host stub)
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("" ::"v"(acc[i][0]), "v"(acc[i][1]));
#endif
        stamp();
        continue;
      }
      const float sc[2] = {(ABL != 2 && L.scale != nullptr) ? L.scale[cb + lr] : 1.0f, (ABL != 2 && L.scale != nullptr) ? L.scale[cb + 32 + lr] : 1.0f};
      if constexpr (POOLV == 0) {
        float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
        int cur_seg = -1;                    // per lane: the halves cross an utterance seam at different registers
        auto publish = [&](bool mine) {      // lanes with `mine` write the moments of their current segment
          const int slot = cur_seg - first_seg;
          if (mine && cur_seg >= 0 && slot >= 0 && slot < p.pool_slots) {
            float *dst = p.pool_partial + ((size_t)((half * p.pool_slots + slot) * 2 + lh) * 3) * p.ld_partial + cb + lr;
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
        for (int i = 0; i < MF; ++i) {
          const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
          const unsigned long long in_frag = 0xffffffffull << ((i & 1) * 32);
          const unsigned long long m_valid = __builtin_amdgcn_ballot_w64(rs_vec >= 0) & in_frag;
          if (m_valid == 0) continue;                                                  // gap rows only
          const int sg0 = __builtin_amdgcn_readlane(rs_vec, __builtin_ctzll(m_valid));
          const unsigned long long m_same = __builtin_amdgcn_ballot_w64(rs_vec == sg0) & in_frag;
          if (m_same == in_frag) {
            // the 32 frames of the fragment belong to one utterance: 4 VALU operations per accumulator register
            const bool chg = cur_seg != sg0;
            if (__builtin_amdgcn_ballot_w64(chg) != 0) {
              publish(chg);
  #pragma unroll
              for (int j = 0; j < 2; ++j) {
                const float v0 = max_lo(acc[i][j][0], act_lo);
                ps[j] = chg ? 0.0f : ps[j]; pq[j] = chg ? 0.0f : pq[j]; pv[j] = chg ? v0 : pv[j];
              }
              cur_seg = chg ? sg0 : cur_seg;
            }
  #pragma unroll
            for (int j = 0; j < 2; ++j) {
              float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float dlt = max_lo(acc[i][j][r], act_lo) - pv[j];
                s4[r & 3] += dlt;
                q4[r & 3] = fmaf(dlt, dlt, q4[r & 3]);
              }
              ps[j] += (s4[0] + s4[1]) + (s4[2] + s4[3]);
              pq[j] += (q4[0] + q4[1]) + (q4[2] + q4[3]);
            }
          } else {
            // an utterance seam or gap rows inside the fragment: register by register (frames ascend with r inside a half)
  #pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int f = (i & 1) * 32 + 8 * (r >> 2) + (r & 3);                    // lane index of the frame of half 0
              const int sg_a = __builtin_amdgcn_readlane(rs_vec, f), sg_b = __builtin_amdgcn_readlane(rs_vec, f + 4);
              const int sg = lh ? sg_b : sg_a;
              const bool ok = sg >= 0;
              const bool chg = ok && sg != cur_seg;
              if (__builtin_amdgcn_ballot_w64(chg) != 0) {
                publish(chg);
  #pragma unroll
                for (int j = 0; j < 2; ++j) {
                  const float v0 = max_lo(acc[i][j][r], act_lo);
                  ps[j] = chg ? 0.0f : ps[j]; pq[j] = chg ? 0.0f : pq[j]; pv[j] = chg ? v0 : pv[j];
                }
                cur_seg = chg ? sg : cur_seg;
              }
  #pragma unroll
              for (int j = 0; j < 2; ++j) {
                const float dlt = ok ? max_lo(acc[i][j][r], act_lo) - pv[j] : 0.0f;
                ps[j] += dlt;
                pq[j] = fmaf(dlt, dlt, pq[j]);
              }
            }
          }
        }
        publish(true);
      } else {
        float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
        const bool first_unit = ABL == 3 && cb < 512;
        if (first_unit) fine(0);
        int cur_seg = -1;                    // uniform: all lanes walk the utterances of the tile together
        bool have = false;                   // per lane: pv is a frame of cur_seg (the lane has had a frame of it in this tile)
        auto publish = [&]() {
          const int slot = cur_seg - first_seg;
          if constexpr (ABL == 2) { asm volatile("" ::"v"(ps[0]), "v"(ps[1]), "v"(pq[0]), "v"(pq[1]), "v"(pv[0]), "v"(pv[1])); return; }
          if (cur_seg >= 0 && slot >= 0 && slot < p.pool_slots) {
            float *dst = p.pool_partial + ((size_t)((half * p.pool_slots + slot) * 2 + lh) * 3) * p.ld_partial + cb + lr;
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
        for (int i = 0; i < MF; ++i) {
          const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
          const int shift = (i & 1) * 32;
          uint32_t rem = (uint32_t)(__builtin_amdgcn_ballot_w64(rs_vec >= 0) >> shift);       // rows of the fragment that belong to an utterance
          if (rem == 0) continue;                                                              // gap rows only
          // u = act(acc) once per fragment (one v_max_f32 each; the accumulators of the fragment die here)
          float u[2][16];
  #pragma unroll
          for (int j = 0; j < 2; ++j)
  #pragma unroll
            for (int r = 0; r < 16; ++r) u[j][r] = max_lo(acc[i][j][r], act_lo);
          // one run per utterance present, in row order (ctz).  bits = its rows inside the fragment.
          while (rem != 0) {
            const int sg = __builtin_amdgcn_readlane(rs_vec, shift + __builtin_ctz(rem));
            const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(rs_vec == sg) >> shift) & rem;
            rem &= ~bits;
            const bool fresh = sg != cur_seg;
            if (fresh) {
              publish();
              cur_seg = sg;
              have = false;
  #pragma unroll
              for (int j = 0; j < 2; ++j) { ps[j] = 0.0f; pq[j] = 0.0f; }
            }
            if (bits == 0xffffffffu) {
              // the whole fragment is one utterance (84 % of the fragments at 200 frames): 2.5 VALU operations per value
              // (every lane has frames here: one without a pivot of this utterance yet - see the seam path - takes its first)
              const bool need = !have;
              have = true;
  #pragma unroll
              for (int j = 0; j < 2; ++j) {
                pv[j] = need ? u[j][0] : pv[j];
                const f32x2_t pv2 = {pv[j], pv[j]};
                f32x2_t s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, q2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  #pragma unroll
                for (int r = 0; r < 16; r += 2) {
                  const f32x2_t uu = {u[j][r], u[j][r + 1]};
                  const f32x2_t d = uu - pv2;
                  s2[(r >> 1) & 1] += d;
                  q2[(r >> 1) & 1] = __builtin_elementwise_fma(d, d, q2[(r >> 1) & 1]);
                }
                const f32x2_t st = s2[0] + s2[1], qt = q2[0] + q2[1];
                ps[j] += st.x + st.y;
                pq[j] += qt.x + qt.y;
              }
            } else {
              // a seam or gap rows: register r of this lane holds frame 8 (r >> 2) + 4 lh + (r & 3) -> bit r of the lane's mask
              const uint32_t x = bits >> (4 * lh);
              const uint32_t lm = (x & 0xfu) | ((x >> 4) & 0xf0u) | ((x >> 8) & 0xf00u) | ((x >> 12) & 0xf000u);
              // pivot = the lane's FIRST frame of the utterance, in whichever fragment of the tile that frame lies.  (Until round 5
              // it was only taken in the utterance's first fragment: a lane half without a frame there - an utterance that starts in
              // the last rows of a fragment - kept the PREVIOUS utterance's pivot for the rest of the tile.  Harmless between
              // utterances of like scale; next to one whose activations are 1e5 x larger the sums about that pivot cancelled and the
              // embedding depended on its batch neighbour: tests/test_gpu_xvector.py::test_pooled_moments_ignore_the_neighbour.)
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
              f32x2_t s2[2] = {{0.f, 0.f}, {0.f, 0.f}}, q2[2] = {{0.f, 0.f}, {0.f, 0.f}};
              const f32x2_t pva = {pv[0], pv[0]}, pvb = {pv[1], pv[1]};
  #pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const int t0 = (int)(lm << (31 - r)) >> 31, t1 = (int)(lm << (30 - r)) >> 31;      // all ones where the frame is in the run
                f32x2_t da = (f32x2_t){u[0][r], u[0][r + 1]} - pva, db = (f32x2_t){u[1][r], u[1][r + 1]} - pvb;
                da.x = __int_as_float(__float_as_int(da.x) & t0); da.y = __int_as_float(__float_as_int(da.y) & t1);
                db.x = __int_as_float(__float_as_int(db.x) & t0); db.y = __int_as_float(__float_as_int(db.y) & t1);
                s2[0] += da; q2[0] = __builtin_elementwise_fma(da, da, q2[0]);
                s2[1] += db; q2[1] = __builtin_elementwise_fma(db, db, q2[1]);
              }
              ps[0] += s2[0].x + s2[0].y; pq[0] += q2[0].x + q2[0].y;
              ps[1] += s2[1].x + s2[1].y; pq[1] += q2[1].x + q2[1].y;
            }
          }
          if (first_unit) fine(1 + i);
        }
        publish();
        if (first_unit) fine(5);
      }
      if constexpr (ABL == 4) __builtin_amdgcn_s_barrier();
      stamp();                             // 8, 10, 12: pooling epilogue of the unit done
    }
    if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

ChainTilePlan chain_tile_plan(int rows, bool allow_tail) {
  ChainTilePlan plan;
  const int n = rows / 128;
  plan.n128 = n;
  // ASV_AMD_CHAIN_TAIL (read once; results agree to the order of the f32 sums of the pooled moments): 0 = 128-frame tiles always,
  // 2 = also cut the last round of a multi-round batch into 96-frame tiles (measured: see asv_internal.h)
  static const int mode = getenv("ASV_AMD_CHAIN_TAIL") != nullptr ? atoi(getenv("ASV_AMD_CHAIN_TAIL")) : 1;
  if (!allow_tail || mode == 0) return plan;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (n < cus) {                                     // less than one round: the smallest tile that still fits one round
    for (int r : {64, 96}) {
      const int t = (rows + r - 1) / r;
      if (t <= cus) { plan.n128 = 0; plan.n_tail = t; plan.tail_rows = r; return plan; }
    }
    return plan;
  }
  if (mode == 2) {
    const int tail = n % cus, as96 = (tail * 128 + 95) / 96;
    if (tail > 0 && as96 <= cus) { plan.n128 = n - tail; plan.n_tail = as96; plan.tail_rows = 96; }
  }
  return plan;
}

int launch_tdnn_chain(const TdnnChainParams &p0, hipStream_t s) {
  TdnnChainParams p = p0;
  constexpr int CM = 128;
  ASV_REQUIRE(p.rows % CM == 0 && p.rows >= CM, "tdnn(chain): rows %d not a multiple of %d", p.rows, CM);
  ASV_REQUIRE(p.cin_pad % CBK == 0 && p.cin_pad >= CBK && p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS, "tdnn(chain): first layer with %d channels / %d taps", p.cin_pad, p.n_taps);
  ASV_REQUIRE((unsigned long long)p.rows * (unsigned long long)p.ldx * 2ull < (1ull << 32), "tdnn(chain): input matrix beyond 32-bit offsets");
  ASV_REQUIRE(p.first.wfrag && p.last.wfrag && p.last.bias && p.n_mid >= 0 && p.n_mid <= 2 && p.last.cout_pad % 64 == 0, "tdnn(chain): incomplete layer description");
  ASV_REQUIRE(p.pool_partial && p.row_seg && p.pool_slots >= 1, "tdnn(chain): the last layer feeds the fused pooling (partials / row map missing)");
  for (int t = 0; t < p.n_taps; ++t) ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(chain): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  if (p.n128 == 0 && p.n_tail == 0) p.n128 = p.rows / CM;           // callers without a plan: 128-frame tiles throughout
  ASV_REQUIRE(p.n_tail == 0 || p.tail_rows == 96 || p.tail_rows == 64, "tdnn(chain): tail tiles of %d frames", p.tail_rows);
  ASV_REQUIRE(p.n128 * 128 + p.n_tail * p.tail_rows >= p.rows && p.n128 * 128 <= p.rows, "tdnn(chain): tile plan %d x 128 + %d x %d does not cover %d rows", p.n128, p.n_tail,
              p.tail_rows, p.rows);
  const dim3 block(512);
  p.row_base = 0; p.tile_base = 0;
#ifdef ASV_WITH_ABLATION
  // Developer build only (libasv_amd_dev.so, `make dev`): the ablation instantiations (results are garbage), the first pooling
  // epilogue and the four-wave kernel.  The switches are read ONCE per process; with ASV_AMD_LIVE_TUNE=1 (tools/chain_ab.py:
  // in-process interleaved A/B) at every launch.  The product library has none of this code: no environment variable reaches it.
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  auto read_int = [](const char *name) { const char *v = getenv(name); return v != nullptr ? atoi(v) : -1; };
  static const int poolv0 = read_int("ASV_AMD_CHAIN_POOLV"), abl0 = read_int("ASV_AMD_CHAIN_ABL");
  const int poolv = live ? read_int("ASV_AMD_CHAIN_POOLV") : poolv0, abl = live ? read_int("ASV_AMD_CHAIN_ABL") : abl0;
  // ASV_AMD_CHAIN_WAVES=4: the 4-wave form (tools/kernels_tdnn_chain4.hip) where it applies.  Measured 0 - 1 % SLOWER than this
  // kernel (profiles/r3k_*): kept as the reproducible A/B of that design.
  static const int waves0 = read_int("ASV_AMD_CHAIN_WAVES");
  const int waves = live ? read_int("ASV_AMD_CHAIN_WAVES") : waves0;
  if (p.n_tail == 0) {
    const dim3 grid(p.n128);
    if (waves == 4 && abl <= 0 && poolv != 0 && !(p.dbg != nullptr && p.dbg_fine) && tdnn_chain4_supported(p)) return launch_tdnn_chain4(p, s);
    if (p.et == ET_BF16 && !(p.dbg != nullptr && p.dbg_fine) && (abl == 1 || abl == 2 || (abl == 4 && p.last.cout_pad % 512 == 0) || poolv == 0)) {
      if (abl == 1) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 1>), grid, block, 0, s, p);
      else if (abl == 2) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 2>), grid, block, 0, s, p);
      else if (abl == 4) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 4>), grid, block, 0, s, p);   // every wave runs the same number of units
      else hipLaunchKernelGGL(tdnn_chain_kernel<0>, grid, block, 0, s, p);
      ASV_HIP_CHECK(hipGetLastError());
      return ASV_OK;
    }
  }
#endif
  if (p.n128 > 0) {
    const dim3 grid(p.n128);
    if (p.et == ET_F16) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_F16>), grid, block, 0, s, p);
    else if (p.dbg != nullptr && p.dbg_fine) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 3>), grid, block, 0, s, p);   // stamps only: results valid
    else hipLaunchKernelGGL(tdnn_chain_kernel<1>, grid, block, 0, s, p);
  }
  if (p.n_tail > 0) {
    // the smaller tiles (chain_tile_plan): behind the full rounds, if any, on the same stream
    p.row_base = p.n128 * 128; p.tile_base = p.n128;
    p.dbg = nullptr;                                   // (the stamp buffer is laid out for the 128-frame launch)
    const dim3 grid(p.n_tail);
    if (p.tail_rows == 96) {
      if (p.et == ET_F16) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_F16, 0, 3>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 0, 3>), grid, block, 0, s, p);
    } else {
      if (p.et == ET_F16) hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_F16, 0, 2>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((tdnn_chain_kernel<1, ET_BF16, 0, 2>), grid, block, 0, s, p);
    }
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
