// 16-bit TDNN / 1x1-conv implicit GEMM, 256 x 256 tiles, BOTH operands through LDS-DMA, four phases per K-tile with counted
// waits - the structure of the 8-phase GEMM template of /opt/skills/guides/cdna_hip_programming.md (":612-700", T2 - T5), built for
// this library's layer shapes (round 5).  It replaces, for the layers it takes, kernels_tdnn_v3.hip's 128 x 256
// tile (feature window through LDS, weight fragments straight from L2, two self-synchronising workgroups per CU).
//
//   * One workgroup of 8 waves (2 along the frames x 4 along the channels) per CU, a wave owns 128 frames x 64 channels = 4 x 2
//     accumulators of v_mfma_f32_32x32x16 (D = W . X^T: lane = frame, registers = channels - the operand order and the epilogue of
//     kernels_tdnn_v3.hip, so the outputs are bit-identical to that kernel's: same products, same k order).
//   * K walks (64-channel chunk) x (tap): a K-tile is 256 frames x 64 channels of X, read at row offset d_tap (a tap is nothing but
//     a shifted row window; rows beyond the matrix ends are clamped onto its zero gap rows), and 256 channels x 64 of the weights in
//     their plain [cout][tap][cin] order.  Each operand tile is staged as TWO half-tiles of 128 rows x 128 bytes (16 KiB, 2 LDS-DMA
//     instructions per wave): HB0 | HA0 | HB1 | HA1.  The rows of a half-tile are not contiguous in the matrix: half-tile HA0 holds,
//     for each of the two wave rows, the FIRST 64 of its 128 frames, HA1 the second 64; HB0 the first 32 of every wave column's 64
//     channels, HB1 the second 32.  A wave therefore needs HB0 + HA0 for its first accumulator quadrant, HB1 for the second, HA1
//     for the third, and the half-tiles become free in that order.
//   * LDS: 2 buffers x 4 half-tiles = 128 KiB (+ 3 KiB epilogue constants).  16-byte slots XOR-swizzled by (row >> 1) & 7 on the
//     DMA SOURCE address and on the ds_read_b128 address (the LDS image of a DMA is lane-linear).
//   * Production schedule: TWO phases per K-tile, each {fragment reads | two half-tiles staged | lgkmcnt(0) + counted vmcnt | barrier |
//     16 MFMAs at priority 1 | barrier}:
//         PA: read HB0 (4) + HA0 (8) + HB1 (4), stage HB1 + HA1 of K-tile kt+1, vmcnt(8) - HA1 of this K-tile has landed -, Q(A0,B0) Q(A0,B1)
//         PB: read HA1 (8),                      stage HB0 + HA0 of K-tile kt+2, vmcnt(6) - HB0, HA0, HB1 of K-tile kt+1 -,  Q(A1,B1) Q(A1,B0)
//     Counted vmcnt, never 0 in the loop: two or three half-tiles stay in flight across every barrier.  The two wave rows run
//     STAGGERED by one barrier (the second row executes one extra barrier in front of the loop, the first one behind it): on every
//     SIMD one wave is inside its MFMA cluster while its partner issues reads and DMA - worth 15 - 19 % (tools/p8_probe).
//     Why this is race free (the rules of the guide, ":660-669"):
//       RAW  a staged half-tile is read one phase after the wait that retires it: the wait stands in front of the phase's first
//            barrier, the reads it covers start in the next phase; with the stagger, the lagging row has executed ITS wait before the
//            leading row passes that phase's second barrier.
//       WAR  the reads of a phase are retired (lgkmcnt(0)) IN FRONT OF the phase's first barrier, so a half-tile may be restaged one
//            phase after it was read: HB0 / HA0 read in PA -> restaged in PB; HA1 read in PB -> restaged in the next PA (HB1 read in
//            PA -> restaged in the next PA: two phases).  A staging wave has passed the reading phase's second barrier, hence every
//            wave of BOTH rows has passed its first one, hence executed the lgkmcnt(0) in front of it.
//     The first two versions ran FOUR phases per K-tile (8 MFMAs between barriers, one half-tile staged per phase, vmcnt(6) once per
//     K-tile - the guide's template as it stands; reads per phase 12 / 4 / 8 / 0, then 8 / 4 / 8 / 4 with the next K-tile's HB0 read
//     in the last phase into a second register set).  It stays in the developer build (variants 40 / 41 / 44): the same rate to
//     +-4 % on every shape (profiles/r5l_p8_shapes.txt) - its MFMA + barrier skeleton and this one's are equally fast: the number of
//     barriers is not what the loop's ~20 % of non-MFMA cycles are - on 254 registers and an even number of K-tiles only.
//   * Production = tdnn_gemm_p8p_kernel further down: this schedule inside a persistent tile loop (the next tile's first K-tile is
//     requested in front of the epilogue).  The one-tile kernel here stays for A/B (variant 50) and carries the measurement variants.
//   * Epilogue: kernels_tdnn_v3.hip's (bias -> ReLU -> folded BN, packed 16-bit, wave-private LDS transpose, 16-byte row stores).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int P8_HALF = 128 * 128;                  // bytes of a half-tile
constexpr int P8_OFF_B0 = 0, P8_OFF_A0 = P8_HALF, P8_OFF_B1 = 2 * P8_HALF, P8_OFF_A1 = 3 * P8_HALF;
constexpr int P8_BUF = 4 * P8_HALF;                 // 64 KiB per K-tile buffer
constexpr int P8_PARAM_OFF = 2 * P8_BUF;
constexpr int P8_LDS_BYTES = 2 * P8_BUF + 3 * 256 * 4;
constexpr int P8_ROWB = 128;

typedef __attribute__((address_space(3))) unsigned char p8_lds_byte;

// one LDS-DMA instruction: 64 lanes x 16 bytes from (scalar base + per-lane 32-bit byte offset) to LDS [M0 .. M0 + 1024).
// M0 is declared clobbered instead of saved and restored (nothing else in this kernel reads it: two SALU operations less per piece).
__device__ __forceinline__ void p8_glds(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory", "m0");
}

// VAR (measurement variants of the developer build; 0 = the production kernel):
//   1 no stagger between the wave rows    2 no s_setprio around the MFMA clusters (+0 ... +5 % with it)    3 neither
//   4 MFMA + barriers only (no reads, no DMA in the loop: the skeleton's ceiling; results are garbage)
//   5 no DMA in the loop (fragment reads of stale LDS; garbage)    6 no fragment reads in the loop (garbage)
// ONE_TAP: a 1-tap layer (row offset 0): the DMA source offsets of the feature rows are loop constants
template <int ET, int VAR, bool ONE_TAP, int PH>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_p8_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  static_assert(PH == 2 || PH == 4, "phases per K-tile");
  constexpr bool STAGGER = !(VAR == 1 || VAR == 3), PRIO = !(VAR == 2 || VAR == 3), SKELETON = VAR == 4;
  constexpr bool NO_DMA = SKELETON || VAR == 5, NO_READS = SKELETON || VAR == 6;
  // VAR 7 (results valid): s_memtime stamps of wave 0 into p.partial[workgroup][8]: 0 start, 1 first K-tile in LDS, 2 K loop done, 3 end
  auto stamp = [&](int k) {
    if constexpr (VAR == 7) {
      if (threadIdx.x == 0) reinterpret_cast<unsigned long long *>(p.partial)[(size_t)blockIdx.x * 8 + k] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp(0);
  __shared__ __attribute__((aligned(16))) unsigned char lds[P8_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * 256;
  const int n0 = (tile % n_tiles) * 256;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *wg = reinterpret_cast<const unsigned char *>(p.w);
  const uint32_t x_pitch = (uint32_t)p.ldx * 2u;
  const int cin_pad = p.cin_pad;
  const int n_taps = ONE_TAP ? 1 : p.n_taps;
  const uint32_t w_pitch = (uint32_t)n_taps * (uint32_t)cin_pad * 2u;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(p8_lds_byte *)lds);
  const int nkt = (cin_pad / 64) * n_taps;

  // per-channel epilogue constants (bias | scale | shift of the tile's 256 channels) -> LDS by LDS-DMA, one 1 KiB piece each, issued by
  // waves 0 - 2 in front of the prologue's pieces (they are then older than everything the prologue's counted wait leaves in flight).
  // As compiler-visible loads they cost a memory round trip of their own in front of the first DMA piece - hipcc waits vmcnt(0) around
  // every load it knows of when untracked (inline-asm) pieces are in flight.
  float *lds_par = reinterpret_cast<float *>(lds + P8_PARAM_OFF);
  auto stage_params = [&]() {
    if (wave < 3) {
      const float *src = (wave == 0) ? p.bias : (wave == 1 ? p.scale : p.shift);
      if (src != nullptr) {
        p8_glds(src + n0, (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)P8_PARAM_OFF + (uint32_t)wave * 1024u));
      } else {
        const float dflt = (wave == 1) ? 1.0f : 0.0f;
        *reinterpret_cast<float4 *>(lds_par + wave * 256 + lane * 4) = make_float4(dflt, dflt, dflt, dflt);
      }
    }
  };

  // ---- LDS-DMA pieces: a half-tile = 16 pieces of 8 rows x 128 B; this wave issues pieces 2 wave and 2 wave + 1 of every half-tile.
  // Half-tile row r <-> matrix row:  A: frame m0 + (r >> 6) * 128 + (r & 63) (+ 64 in HA1);  B: channel n0 + (r >> 5) * 64 + (r & 31) (+ 32 in HB1)
  const int g_row = lane >> 3, g_slot = lane & 7;
  int a_row[2];                 // matrix row of this lane's row in HA0 (tap offset and + 64 for HA1 are added per piece, then clamped)
  uint32_t a_slot[2];           // byte offset of the 16-byte slot this lane fetches (the swizzle lives on the source side)
  uint32_t a_voff[2][2];        // ONE_TAP: the whole per-lane source offset of HA0 / HA1
  uint32_t b_off[2];            // byte offset into the weights of this lane's 16 bytes in HB0 (+ 32 rows for HB1, + (tap, chunk) per K-tile)
  const int last_row = p.rows - 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + g_row;
    const uint32_t slot16 = (uint32_t)(g_slot ^ ((r >> 1) & 7)) * 16u;
    a_row[i] = m0 + (r >> 6) * 128 + (r & 63);
    a_slot[i] = slot16;
    a_voff[0][i] = (uint32_t)a_row[i] * x_pitch + slot16;
    a_voff[1][i] = (uint32_t)min(a_row[i] + 64, last_row) * x_pitch + slot16;
    b_off[i] = (uint32_t)(n0 + (r >> 5) * 64 + (r & 31)) * w_pitch + slot16;
  }
  // tap t in lane t (v_readlane in the loop: no s_load + lgkmcnt(0) there); built from the scalar kernel arguments - indexing the
  // argument array by the lane is a GLOBAL load, and the compiler's vmcnt(0) in front of its first use drained the first DMA pieces
  int v_taps = p.taps[0];
  if (!ONE_TAP) {
#pragma unroll
    for (int t = 1; t < ASV_MAX_TAPS; ++t) v_taps = (lane == t) ? p.taps[t] : v_taps;
  }

  // stage half-tile `which` (0 HB0, 1 HA0, 2 HB1, 3 HA1) of K-tile (chunk c, tap t) into buffer b
  auto stage = [&](int which, int c, int t, int b, bool in_loop) {
    if (NO_DMA && in_loop) return;
    const uint32_t dst0 = lds_base + (uint32_t)b * P8_BUF + (uint32_t)which * P8_HALF + (uint32_t)wave * 2048u;
    if (which & 1) {                                 // an A half-tile
      const unsigned char *base = xg + (size_t)c * 128;
      if (ONE_TAP) {
#pragma unroll
        for (int i = 0; i < 2; ++i) p8_glds(base, a_voff[which == 3][i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
      } else {
        const int d = __builtin_amdgcn_readlane(v_taps, t) + (which == 3 ? 64 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = min(max(a_row[i] + d, 0), last_row);
          p8_glds(base, (uint32_t)row * x_pitch + a_slot[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
        }
      }
    } else {
      const unsigned char *base = wg + ((size_t)t * cin_pad + (size_t)c * 64) * 2 + (which == 2 ? (size_t)32 * w_pitch : 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) p8_glds(base, b_off[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
    }
  };

  // ---- fragment reads: 16 bytes of row (lr), k-slot (2 kg + lh) ^ swizzle
  const uint32_t sw = (uint32_t)((lr >> 1) & 7);
  const uint32_t a_base = (uint32_t)(wm * 64 + lr) * P8_ROWB, b_base = (uint32_t)(wn * 32 + lr) * P8_ROWB;
  uint32_t a_addr[4], b_addr[4];          // per k-group; the K-tile buffer (bit 16) is added per K-tile
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const uint32_t s16 = (((uint32_t)(kg * 2 + lh)) ^ sw) * 16u;
    a_addr[kg] = a_base + s16;
    b_addr[kg] = b_base + s16;
  }

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  if constexpr (PH == 2) {
    // ================= two phases per K-tile (16 MFMAs between barriers) =================
    //     PA: read HB0 (4) + HA0 (8) + HB1 (4), stage HB1 + HA1 of K-tile kt+1, lgkmcnt(0), vmcnt(8) - HA1 of THIS K-tile has landed -, barrier, Q(A0,B0) Q(A0,B1), barrier
    //     PB: read HA1 (8),                      stage HB0 + HA0 of K-tile kt+2, lgkmcnt(0), vmcnt(6) - HB0 HA0 HB1 of K-tile kt+1 have landed -, barrier, Q(A1,B1) Q(A1,B0), barrier
    // Half the barriers per MFMA of the four-phase form (SQ counters of that form: its MFMA + barriers skeleton keeps the matrix pipe
    // busy 0.56 - 0.61 of the cycles - ~110 cycles go by around every barrier beside 256 of MFMA).  The reads of a phase are retired
    // (lgkmcnt(0)) IN FRONT OF the phase's first barrier, so a half-tile may be restaged one phase after it was read (the guide's
    // WAR rule, second form): HB0 / HA0 read in PA -> restaged in PB; HA1 read in PB -> restaged in the next PA; HB1 read in PA ->
    // restaged in the next PA.  RAW as before: a wait stands in front of a phase's first barrier, the reads it covers come a phase
    // later.  Prefetch distance: two phases (~1.2 k cycles) for the feature half-tiles, one for HB1 (weights: L2-resident).
    stage_params();
    stage(0, 0, 0, 0, false); stage(1, 0, 0, 0, false); stage(2, 0, 0, 0, false); stage(3, 0, 0, 0, false);
    {
      int c1 = 0, t1 = 1;
      if (t1 == n_taps) { t1 = 0; c1 = 1; }
      if (nkt > 1) {
        stage(0, c1, t1, 1, false); stage(1, c1, t1, 1, false);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    uint4 wb0[4], wb1[4], xa[2][4];
    if (NO_READS) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        wb0[kg] = make_uint4(lane, 1, 2, 3); wb1[kg] = make_uint4(3, lane, 1, 0);
        xa[0][kg] = make_uint4(1, 1, lane, 1); xa[1][kg] = make_uint4(2, 2, 2, lane);
      }
    }
    stamp(1);
    if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();      // this wave row runs one barrier behind the other from here on
    auto mma_q = [&](const uint4 (&wfr)[4], int j, int i0) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) acc[i0 + i2][j] = mfma16<ET>(wfr[kg], xa[i2][kg], acc[i0 + i2][j]);
    };
    auto barrier = [&]() {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    int c1 = 0, t1 = 0, c2 = 0, t2 = 0;     // (chunk, tap) of K-tile kt + 1 and of K-tile kt + 2
    auto adv = [&](int &c, int &t) { if (++t == n_taps) { t = 0; ++c; } };
    adv(c1, t1); adv(c2, t2); adv(c2, t2);
    // TAIL 0: K-tiles kt + 1 and kt + 2 exist; 1: kt + 1 is the last; 2: kt is the last
    auto ktile2 = [&](int kt, auto tail_c) {
      constexpr int TAIL = decltype(tail_c)::value;
      const int b = kt & 1;
      const unsigned char *L = lds + (uint32_t)b * P8_BUF;
      // ---- PA
      if (!NO_READS) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) wb0[kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_B0 + b_addr[kg]);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A0 + i2 * (32 * P8_ROWB) + a_addr[kg]);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) wb1[kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_B1 + b_addr[kg]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TAIL <= 1) { stage(2, c1, t1, b ^ 1, true); stage(3, c1, t1, b ^ 1, true); }      // HB1, HA1 of K-tile kt + 1
      if (TAIL <= 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");            // HA1 of this K-tile has landed; this phase's reads are back
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      mma_q(wb0, 0, 0);
      mma_q(wb1, 1, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
      // ---- PB
      if (!NO_READS) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A1 + i2 * (32 * P8_ROWB) + a_addr[kg]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (TAIL == 0) { stage(0, c2, t2, b, true); stage(1, c2, t2, b, true); }                // HB0, HA0 of K-tile kt + 2
      if (TAIL == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");            // HB0, HA0, HB1 of K-tile kt + 1 have landed
      else if (TAIL == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      mma_q(wb1, 1, 2);
      mma_q(wb0, 0, 2);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
      adv(c1, t1); adv(c2, t2);
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 2 < nkt; ++kt) ktile2(kt, J0{});
    if (kt + 1 < nkt) { ktile2(kt, J1{}); ++kt; }
    ktile2(kt, J2{});
  } else {
  // ---- prologue: K-tile 0 complete + the first three half-tiles of K-tile 1 in flight
  {
    int c1 = 0, t1 = 1;
    if (t1 == n_taps) { t1 = 0; c1 = 1; }
    stage_params();
    stage(0, 0, 0, 0, false); stage(1, 0, 0, 0, false); stage(2, 0, 0, 0, false); stage(3, 0, 0, 0, false);
    if (nkt > 1) {
      stage(0, c1, t1, 1, false); stage(1, c1, t1, 1, false); stage(2, c1, t1, 1, false);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // weight fragments of HB0 (two sets: the next K-tile's are read in P4, while the current ones feed P4's MFMAs) and HB1, frame
  // fragments of the A half-tile in use
  uint4 wb0[2][4], wb1[4], xa[2][4];
  if (NO_READS) {
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      wb0[0][kg] = make_uint4(lane, 1, 2, 3); wb0[1][kg] = make_uint4(lane, 5, 2, 3); wb1[kg] = make_uint4(3, lane, 1, 0);
      xa[0][kg] = make_uint4(1, 1, lane, 1); xa[1][kg] = make_uint4(2, 2, 2, lane);
    }
  } else {
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) wb0[0][kg] = *reinterpret_cast<const uint4 *>(lds + P8_OFF_B0 + b_addr[kg]);     // HB0 of K-tile 0
  }
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();        // this wave row runs one barrier behind the other from here on

  auto mma_q = [&](const uint4 (&wfr)[4], int j, int i0) {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) acc[i0 + i2][j] = mfma16<ET>(wfr[kg], xa[i2][kg], acc[i0 + i2][j]);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // K-tile kt in buffer kt & 1 (= PAR, also the register set of its HB0 fragments).
  // TAIL 0: K-tiles kt + 1 and kt + 2 exist; 1: kt + 1 is the last; 2: kt is the last.
  int cs = 0, ts = 0;                       // (chunk, tap) of K-tile kt + 1, then of kt + 2 (the staging cursor runs ahead)
  auto advance = [&]() { if (++ts == n_taps) { ts = 0; ++cs; } };
  advance();                                // K-tile 1
  auto ktile = [&](auto tail_c, auto par_c) {
    constexpr int TAIL = decltype(tail_c)::value, PAR = decltype(par_c)::value;
    const unsigned char *L = lds + PAR * P8_BUF, *Ln = lds + (PAR ^ 1) * P8_BUF;
    // ---- P1: read HA0 (8), stage HA1 of K-tile kt + 1
    if (!NO_READS) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A0 + i2 * (32 * P8_ROWB) + a_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL <= 1) stage(3, cs, ts, PAR ^ 1, true);
    if (TAIL == 0) advance();                                    // the cursor moves on to K-tile kt + 2
    barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma_q(wb0[PAR], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    barrier();
    // ---- P2: read HB1 (4), stage HB0 of K-tile kt + 2 (HB0 of this K-tile was read in P4 of the previous one)
    if (!NO_READS) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) wb1[kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_B1 + b_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL == 0) stage(0, cs, ts, PAR, true);
    barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma_q(wb1, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    barrier();
    // ---- P3: read HA1 (8), stage HA0 of K-tile kt + 2; HB0 of K-tile kt + 1 (staged 5 | 3 half-tiles ago) has landed behind this wait
    if (!NO_READS) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A1 + i2 * (32 * P8_ROWB) + a_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL == 0) {
      stage(1, cs, ts, PAR, true);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if (TAIL == 1) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mma_q(wb1, 1, 2);
    __builtin_amdgcn_sched_barrier(0);
    barrier();
    // ---- P4: read HB0 of K-tile kt + 1 (4) into the other register set, stage HB1 of K-tile kt + 2; K-tile kt + 1 has landed behind this wait
    if (!NO_READS && TAIL <= 1) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) wb0[PAR ^ 1][kg] = *reinterpret_cast<const uint4 *>(Ln + P8_OFF_B0 + b_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TAIL == 0) {
      stage(2, cs, ts, PAR, true);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");           // everything but the last three half-tiles: K-tile kt + 1 is in LDS
    } else if (TAIL == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the HB0 reads retire here: two phases before that half-tile is restaged
    __builtin_amdgcn_sched_barrier(0);
    mma_q(wb0[PAR], 0, 2);
    __builtin_amdgcn_sched_barrier(0);
    barrier();
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  // nkt is even (tdnn_p8_supported): pairs of K-tiles, one per register set of the HB0 fragments, then the two closing ones.  (One
  // straight line of code on purpose: with the three possible remainders of an odd / even count as if - else alternatives the
  // register allocator kept a copy of the 128 accumulator registers per alternative - 560 spilled registers.)
  for (int kt = 0; kt + 2 < nkt; kt += 2) { ktile(I0{}, I0{}); ktile(I0{}, I1{}); }
  ktile(I1{}, I0{});
  ktile(I2{}, I1{});
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();        // the rows meet again: every wave is done with the buffers
  asm volatile("" ::: "memory");
  stamp(2);

  // ---- epilogue (kernels_tdnn_v3.hip) ---------------------------------------------------------
  // acc[i][j][r]: frame = m0 + wm*128 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  unsigned char *scr = lds + wave * (128 * P8_ROWB);     // [128 frames][64 channels] 16-bit, 128-B rows, swizzled slots
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  uint32_t vmask = 0;                                     // bit i: this lane's frame of m-fragment i is a real frame
#pragma unroll
  for (int i = 0; i < 4; ++i) vmask |= ((p.row_valid[(m0 + wm * 128 + i * 32) >> 5] >> lr) & 1u) << i;
  auto swz = [](int row, int slot) { return slot ^ ((row >> 1) & 7); };
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;                 // channel inside the tile
      const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
      const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
      const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 512 + chl);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      const int slot = j * 4 + q;                // channel offset j*32 + 8*q + 4*lh -> 16-B slot, 8-B half lh
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (vmask >> i) & 1u;
        const int frow = i * 32 + lr;            // row inside the wave's scratch tile
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], true);
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        pk.x = valid ? pk.x : 0u;                  // gap rows are zeros: on the packed pairs, 2 selects per 4 values
        pk.y = valid ? pk.y : 0u;
        // odd rows keep their two 8-byte halves swapped so rows r, r+1 (same slot) hit different banks
        *reinterpret_cast<uint2 *>(scr + frow * P8_ROWB + swz(frow, slot) * 16 + ((lh ^ (frow & 1)) * 8)) = pk;
      }
    }
  }
  // the scratch tile belongs to this wave only: LDS ops of one wave complete in order
  {
    unsigned char *yg = reinterpret_cast<unsigned char *>(p.y);
    const size_t y_pitch = (size_t)p.ldy * 2;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int piece = it * 64 + lane, frow = piece >> 3, slot = piece & 7;
      uint4 v = *reinterpret_cast<const uint4 *>(scr + frow * P8_ROWB + swz(frow, slot) * 16);
      if (frow & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      const int ch = n0 + wn * 64 + slot * 8;
      const int row = m0 + wm * 128 + frow;
      if (ch < p.cout_store) *reinterpret_cast<uint4 *>(yg + (size_t)row * y_pitch + (size_t)ch * 2) = v;
    }
  }
  if constexpr (VAR == 7) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(3);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the kernel above (two-phase schedule): as many workgroups as the chip has CUs, each walks its tiles (it,
// it + grid, ...).  Per-tile stamps of the one-tile form (tools/p8_probe, ECAPA's 1024 -> 1024 layer): prologue 7.8 k cycles
// (parameters + the first K-tile from HBM with nothing to overlap), K loop 38.8 k (2.4 k per K-tile of 2.05 k of MFMA issue),
// epilogue + store drain 7.4 k, and ~8 k between the workgroups of a CU: 28 % of a tile outside its K loop.  Here
//   * the NEXT tile's first K-tile (4 half-tiles -> buffer 0) and its parameters are requested BEFORE the epilogue of this tile; the
//     epilogue's LDS transposition runs in two passes of 64 rows per wave inside buffer 1 (64 KiB), so the two do not meet;
//   * the epilogue constants AND the tile's row-validity words arrive by LDS-DMA into one of two parameter slots: the kernel has
//     no compiler-visible load at all (a visible load next to untracked DMA pieces costs a vmcnt(0));
//   * the stores of a tile drain under the next tile's K loop; the wait in front of that loop is vmcnt(4) (everything but the two
//     half-tiles of its second K-tile, requested behind the stores).
constexpr int P8P_PAR_SLOT = 4096;                  // bias | scale | shift (3 x 1 KiB) | row-validity words (1 KiB piece, 32 B used)
constexpr int P8P_LDS_BYTES = 2 * P8_BUF + 2 * P8P_PAR_SLOT;

template <int ET, bool ONE_TAP>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_p8p_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[P8P_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int total = m_tiles * n_tiles;
  const int grid = gridDim.x;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *wg = reinterpret_cast<const unsigned char *>(p.w);
  const uint32_t x_pitch = (uint32_t)p.ldx * 2u;
  const int cin_pad = p.cin_pad;
  const int n_taps = ONE_TAP ? 1 : p.n_taps;
  const uint32_t w_pitch = (uint32_t)n_taps * (uint32_t)cin_pad * 2u;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(p8_lds_byte *)lds);
  const int nkt = (cin_pad / 64) * n_taps;
  const int last_row = p.rows - 1;
  const int g_row = lane >> 3, g_slot = lane & 7;

  // per-tile, per-lane DMA source offsets (see the one-tile kernel)
  struct TileAddr { int m0, n0; int a_row[2]; uint32_t a_slot[2], a_voff[2][2], b_off[2]; };
  auto tile_addr = [&](int it) {
    TileAddr t;
    const int tile = xcd_swizzle(it, total);
    t.m0 = (tile / n_tiles) * 256;
    t.n0 = (tile % n_tiles) * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + g_row;
      const uint32_t slot16 = (uint32_t)(g_slot ^ ((r >> 1) & 7)) * 16u;
      t.a_row[i] = t.m0 + (r >> 6) * 128 + (r & 63);
      t.a_slot[i] = slot16;
      t.a_voff[0][i] = (uint32_t)t.a_row[i] * x_pitch + slot16;
      t.a_voff[1][i] = (uint32_t)min(t.a_row[i] + 64, last_row) * x_pitch + slot16;
      t.b_off[i] = (uint32_t)(t.n0 + (r >> 5) * 64 + (r & 31)) * w_pitch + slot16;
    }
    return t;
  };
  int v_taps = p.taps[0];
  if (!ONE_TAP) {
#pragma unroll
    for (int t = 1; t < ASV_MAX_TAPS; ++t) v_taps = (lane == t) ? p.taps[t] : v_taps;
  }
  auto stage = [&](const TileAddr &T, int which, int c, int t, int b) {
    const uint32_t dst0 = lds_base + (uint32_t)b * P8_BUF + (uint32_t)which * P8_HALF + (uint32_t)wave * 2048u;
    if (which & 1) {
      const unsigned char *base = xg + (size_t)c * 128;
      if (ONE_TAP) {
#pragma unroll
        for (int i = 0; i < 2; ++i) p8_glds(base, T.a_voff[which == 3][i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
      } else {
        const int d = __builtin_amdgcn_readlane(v_taps, t) + (which == 3 ? 64 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = min(max(T.a_row[i] + d, 0), last_row);
          p8_glds(base, (uint32_t)row * x_pitch + T.a_slot[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
        }
      }
    } else {
      const unsigned char *base = wg + ((size_t)t * cin_pad + (size_t)c * 64) * 2 + (which == 2 ? (size_t)32 * w_pitch : 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) p8_glds(base, T.b_off[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
    }
  };
  // bias | scale | shift of the tile's 256 channels and the validity words of its 256 rows -> parameter slot `slot`: one piece each
  // from waves 0 .. 3 (the defaults of an absent scale / shift are written by the wave itself)
  auto stage_tile_params = [&](const TileAddr &T, int slot) {
    const uint32_t dst = lds_base + 2u * P8_BUF + (uint32_t)slot * P8P_PAR_SLOT + (uint32_t)wave * 1024u;
    if (wave < 3) {
      const float *src = (wave == 0) ? p.bias : (wave == 1 ? p.scale : p.shift);
      if (src != nullptr) {
        p8_glds(src + T.n0, (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane(dst));
      } else {
        const float dflt = (wave == 1) ? 1.0f : 0.0f;
        *reinterpret_cast<float4 *>(lds + 2 * P8_BUF + slot * P8P_PAR_SLOT + wave * 1024 + lane * 16) = make_float4(dflt, dflt, dflt, dflt);
      }
    } else if (wave == 3) {
      // 8 words = 32 bytes: lanes 0 and 1 fetch them, the others re-read the first 16 bytes (valid memory; their LDS bytes are never used)
      p8_glds(p.row_valid + (T.m0 >> 5), lane < 2 ? (uint32_t)lane * 16u : 0u, __builtin_amdgcn_readfirstlane(dst));
    }
  };

  const uint32_t sw = (uint32_t)((lr >> 1) & 7);
  const uint32_t a_base = (uint32_t)(wm * 64 + lr) * P8_ROWB, b_base = (uint32_t)(wn * 32 + lr) * P8_ROWB;
  uint32_t a_addr[4], b_addr[4];
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) {
    const uint32_t s16 = (((uint32_t)(kg * 2 + lh)) ^ sw) * 16u;
    a_addr[kg] = a_base + s16;
    b_addr[kg] = b_base + s16;
  }
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto kt_ct = [&](int kt, int &c, int &t) { c = kt / n_taps; t = kt - c * n_taps; };
  auto swz = [](int row, int slot) { return slot ^ ((row >> 1) & 7); };
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  unsigned char *yg = reinterpret_cast<unsigned char *>(p.y);
  const size_t y_pitch = (size_t)p.ldy * 2;

  int it = blockIdx.x;
  int slot = 0;
  TileAddr cur = tile_addr(it);
  // ---- request the first tile: parameters, K-tile 0 -> buffer 0, the first two half-tiles of K-tile 1 -> buffer 1
  stage_tile_params(cur, 0);
  stage(cur, 0, 0, 0, 0); stage(cur, 1, 0, 0, 0); stage(cur, 2, 0, 0, 0); stage(cur, 3, 0, 0, 0);
  if (nkt > 1) {
    int c1, t1;
    kt_ct(1, c1, t1);
    stage(cur, 0, c1, t1, 1); stage(cur, 1, c1, t1, 1);
  }
#pragma unroll 1
  while (true) {
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    uint4 wb0[4], wb1[4], xa[2][4];
    if (wm == 1) __builtin_amdgcn_s_barrier();                 // this wave row runs one barrier behind the other inside the K loop
    auto mma_q = [&](const uint4 (&wfr)[4], int j, int i0) {
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) acc[i0 + i2][j] = mfma16<ET>(wfr[kg], xa[i2][kg], acc[i0 + i2][j]);
    };
    int c1 = 0, t1 = 0, c2 = 0, t2 = 0;
    auto adv = [&](int &c, int &t) { if (++t == n_taps) { t = 0; ++c; } };
    adv(c1, t1); adv(c2, t2); adv(c2, t2);
    auto ktile2 = [&](int kt, auto tail_c) {
      constexpr int TAIL = decltype(tail_c)::value;
      const int b = kt & 1;
      const unsigned char *L = lds + (uint32_t)b * P8_BUF;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) wb0[kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_B0 + b_addr[kg]);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A0 + i2 * (32 * P8_ROWB) + a_addr[kg]);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) wb1[kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_B1 + b_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
      if (TAIL <= 1) { stage(cur, 2, c1, t1, b ^ 1); stage(cur, 3, c1, t1, b ^ 1); }
      if (TAIL <= 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma_q(wb0, 0, 0);
      mma_q(wb1, 1, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
#pragma unroll
      for (int kg = 0; kg < 4; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = *reinterpret_cast<const uint4 *>(L + P8_OFF_A1 + i2 * (32 * P8_ROWB) + a_addr[kg]);
      __builtin_amdgcn_sched_barrier(0);
      if (TAIL == 0) { stage(cur, 0, c2, t2, b); stage(cur, 1, c2, t2, b); }
      if (TAIL == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else if (TAIL == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma_q(wb1, 1, 2);
      mma_q(wb0, 0, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
      adv(c1, t1); adv(c2, t2);
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 2 < nkt; ++kt) ktile2(kt, J0{});
    if (kt + 1 < nkt) { ktile2(kt, J1{}); ++kt; }
    ktile2(kt, J2{});
    if (wm == 0) __builtin_amdgcn_s_barrier();                 // the rows meet again: nobody reads the buffers any more, no DMA in flight
    asm volatile("" ::: "memory");

    // ---- the next tile's parameters and first K-tile, requested in front of this tile's epilogue
    const bool has_next = it + grid < total;
    TileAddr nxt = cur;
    if (has_next) {
      nxt = tile_addr(it + grid);
      stage_tile_params(nxt, slot ^ 1);
      stage(nxt, 0, 0, 0, 0); stage(nxt, 1, 0, 0, 0); stage(nxt, 2, 0, 0, 0); stage(nxt, 3, 0, 0, 0);
    }
    // ---- epilogue of `cur` (kernels_tdnn_v3.hip's), two passes of 64 rows per wave through buffer 1
    {
      const float *par = reinterpret_cast<const float *>(lds + 2 * P8_BUF + slot * P8P_PAR_SLOT);
      const uint32_t *vw = reinterpret_cast<const uint32_t *>(lds + 2 * P8_BUF + slot * P8P_PAR_SLOT + 3072);
      unsigned char *scr = lds + P8_BUF + wave * (64 * P8_ROWB);        // [64 frames][64 channels] 16-bit, 128-B rows, swizzled slots
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t vmask = 0;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) vmask |= ((vw[wm * 4 + h * 2 + i2] >> lr) & 1u) << i2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;
            const float4 b4 = *reinterpret_cast<const float4 *>(par + chl);
            const float4 sc4 = *reinterpret_cast<const float4 *>(par + 256 + chl);
            const float4 sh4 = *reinterpret_cast<const float4 *>(par + 512 + chl);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const int sl = j * 4 + q;
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
              const bool valid = (vmask >> i2) & 1u;
              const int frow = i2 * 32 + lr;
              float y[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) y[e] = tdnn_epilogue_fast(acc[h * 2 + i2][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], true);
              uint2 pk;
              pk.x = pack_h16x2<ET>(y[0], y[1]);
              pk.y = pack_h16x2<ET>(y[2], y[3]);
              pk.x = valid ? pk.x : 0u;
              pk.y = valid ? pk.y : 0u;
              *reinterpret_cast<uint2 *>(scr + frow * P8_ROWB + swz(frow, sl) * 16 + ((lh ^ (frow & 1)) * 8)) = pk;
            }
          }
        }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int piece = s8 * 64 + lane, frow = piece >> 3, sl = piece & 7;
          uint4 v = *reinterpret_cast<const uint4 *>(scr + frow * P8_ROWB + swz(frow, sl) * 16);
          if (frow & 1) v = make_uint4(v.z, v.w, v.x, v.y);
          const int ch = cur.n0 + wn * 64 + sl * 8;
          const int row = cur.m0 + wm * 128 + h * 64 + frow;
          if (ch < p.cout_store) *reinterpret_cast<uint4 *>(yg + (size_t)row * y_pitch + (size_t)ch * 2) = v;
        }
      }
    }
    if (!has_next) break;
    barrier();                                                  // every wave is done with its scratch in buffer 1
    if (nkt > 1) {
      int c1n, t1n;
      kt_ct(1, c1n, t1n);
      stage(nxt, 0, c1n, t1n, 1); stage(nxt, 1, c1n, t1n, 1);
    }
    cur = nxt;
    slot ^= 1;
    it += grid;
  }
}

}  // namespace

// Layers the 8-phase kernel takes: 16-bit rows, the plain epilogue (affine -> [ReLU] -> folded BN), whole 64-channel chunks.
bool tdnn_p8_supported(const TdnnKernelParams &p, int et, bool out_f32) {
  const bool fits32 = (unsigned long long)p.rows * (unsigned long long)p.ldx * 2ull < (1ull << 32) &&
                      (unsigned long long)round_up(p.cout_store, 256) * (unsigned long long)p.n_taps * (unsigned long long)p.cin_pad * 2ull < (1ull << 32);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  return p.w != nullptr && et != ET_F32 && !out_f32 && fits32 && fast && p.x2 == nullptr && p.seg_bias == nullptr && p.seg_scale == nullptr && p.res == nullptr &&
         p.pool_partial == nullptr && p.rows % 256 == 0 && p.rows >= 256 && p.cin_pad % 64 == 0 && p.cin_pad >= 64 && p.cout_store % 8 == 0 && p.cout_store >= 192 &&
         p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS && p.row_valid != nullptr;
}

int launch_tdnn_p8_variant(const TdnnKernelParams &p, int variant, hipStream_t s) {
  ASV_REQUIRE(tdnn_p8_supported(p, p.et, false), "tdnn(p8): layer shape not supported (rows %d cin %d cout %d taps %d)", p.rows, p.cin_pad, p.cout_store, p.n_taps);
  const int m_tiles = p.rows / 256, n_tiles = round_up(p.cout_store, 256) / 256;
  const dim3 grid(m_tiles * n_tiles), block(512);
  const bool f16 = p.et == ET_F16;
  const bool one = p.n_taps == 1 && p.taps[0] == 0;
#define ASV_P8(ETV, VARV) ASV_P8P(ETV, VARV, 2)
#define ASV_P8P(ETV, VARV, PHV) do { if (one) hipLaunchKernelGGL((tdnn_gemm_p8_kernel<ETV, VARV, true, PHV>), grid, block, 0, s, p, m_tiles, n_tiles); \
                               else hipLaunchKernelGGL((tdnn_gemm_p8_kernel<ETV, VARV, false, PHV>), grid, block, 0, s, p, m_tiles, n_tiles); } while (0)
  switch (variant) {
#ifdef ASV_WITH_ABLATION
    case 1: ASV_P8(ET_BF16, 1); break;
    case 2: ASV_P8(ET_BF16, 2); break;
    case 4: ASV_P8(ET_BF16, 4); break;
    case 5: ASV_P8(ET_BF16, 5); break;
    case 6: ASV_P8(ET_BF16, 6); break;
    case 7: ASV_P8(ET_BF16, 7); break;
    case 40: case 41: case 44:                       // the four-phase form (round 5, second version) and its variants: an even number of K-tiles
      ASV_REQUIRE(((p.cin_pad / 64) * p.n_taps) % 2 == 0, "tdnn(p8): the four-phase form takes an even number of K-tiles");
      if (variant == 40) ASV_P8P(ET_BF16, 0, 4); else if (variant == 41) ASV_P8P(ET_BF16, 1, 4); else ASV_P8P(ET_BF16, 4, 4);
      break;
#endif
    case 50:                                         // the one-tile two-phase form (A/B)
      if (f16) ASV_P8(ET_F16, 0);
      else ASV_P8(ET_BF16, 0);
      break;
    default: {
      // persistent form: one workgroup per CU
      static int cus = 0;
      if (cus == 0) {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        cus = n > 0 ? n : 256;
      }
      const dim3 pgrid(std::min(m_tiles * n_tiles, cus));
      if (one) { if (f16) hipLaunchKernelGGL((tdnn_gemm_p8p_kernel<ET_F16, true>), pgrid, block, 0, s, p, m_tiles, n_tiles);
                 else hipLaunchKernelGGL((tdnn_gemm_p8p_kernel<ET_BF16, true>), pgrid, block, 0, s, p, m_tiles, n_tiles); }
      else { if (f16) hipLaunchKernelGGL((tdnn_gemm_p8p_kernel<ET_F16, false>), pgrid, block, 0, s, p, m_tiles, n_tiles);
             else hipLaunchKernelGGL((tdnn_gemm_p8p_kernel<ET_BF16, false>), pgrid, block, 0, s, p, m_tiles, n_tiles); }
    }
  }
#undef ASV_P8
#undef ASV_P8P
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_tdnn_p8(const TdnnKernelParams &p, hipStream_t s) { return launch_tdnn_p8_variant(p, 0, s); }

}  // namespace asv
