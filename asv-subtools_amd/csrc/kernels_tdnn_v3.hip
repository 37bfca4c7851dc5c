// bf16 TDNN / 1x1-conv implicit GEMM, large-tile variant 3: feature window through LDS,
// weight fragments straight from L2 into registers.
//
// Measured on the variant-2 kernel (tools/gemm_ablate, 52k frames x 512 -> 512, 3 taps): the pure
// MFMA + barrier skeleton needs 71 us, the ds_read_b128 stream alone 91 us, the LDS-DMA stream alone
// 59 us - the kernel was LDS-read bound (6 fragment reads per 8 MFMAs at ~128 B/clk/CU), not MFMA
// or HBM/L2 bound.  This variant removes a third of the LDS reads and most LDS-DMA traffic:
//
//   * weights are pre-packed on the host in MFMA-fragment order
//         Wf[n_frag32][tap][chunk64][k_group16][lane][8 x bf16]
//     so a wave fetches the fragment of one k-group with ONE perfectly coalesced 1 KiB
//     global_load_dwordx4 (the two M-waves that share a channel range hit in L1).  Each fragment
//     register is re-loaded for the NEXT step right after its MFMAs were issued (prefetch distance:
//     one step, no extra registers).
//   * only the feature window (264 frames x 64 channels, shared by all taps of a chunk) goes through
//     LDS, by LDS-DMA into a 3-stage ring (prefetch distance: two chunks).  The window is read-only for
//     the whole chunk, so the tap steps of a chunk run WITHOUT any barrier: one barrier per chunk,
//     and a counted `s_waitcnt vmcnt(12)` that waits for the window of the next chunk but not for the
//     weight fragments / window pieces issued during the last step.
//   * 4 LDS fragment reads per 8 MFMAs; epilogue as in variant 2 (4 consecutive channels per lane,
//     LDS-staged 16-byte stores).
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int BN = 256;
constexpr int ROWB = 128;
constexpr int N_STAGES = 4;
constexpr int BK = 64;
static_assert(BN == kBigTileN, "weight padding must match the N tile");
// WM = number of 128-frame wave rows per workgroup:
//   WM = 2: 256 x 256 tile, 8 waves, 135 KiB LDS, one workgroup per CU
//   WM = 1: 128 x 256 tile, 4 waves,  71 KiB LDS, TWO workgroups per CU - the prologue (first window + weight
//           fragments in flight) and the epilogue of one workgroup overlap with the main loop of the other
//   WM = 0:  64 x 256 tile, 4 waves (wave tile 64 x 64 = 2 x 2 accumulators), 39 KiB LDS, <= 168 VGPRs: THREE
//           workgroups per CU.  For layers whose tile count does not fill whole rounds of 512 workgroups: 816
//           tiles of 128 rows are 2 rounds (1.59 used), 1632 tiles of 64 rows on 768 slots are 3 half-rounds.
//   WM = 3: 128 x 128 tile, 4 waves = 2 x 2 of 64 rows x 64 channels, two workgroups per CU: layers with 97..128 output
//           channels and a deep K (ECAPA's attention bottleneck 1536 -> 128 + per-utterance bias + tanh, the im2col'd
//           576 -> 128 convolution of the ResNet trunk).  On half-filled 256-channel tiles they were slower than on the
//           128 x 128 register-staged tile (231 vs 172 us, r2c); with 128-row x 32-channel waves every row read fed one
//           MFMA only and the LDS was the limit (138 us, r2l).
template <int WM> struct Geom3 {
  static constexpr int MF = (WM == 0 || WM == 3) ? 2 : 4;   // 32-frame accumulator fragments per wave
  static constexpr int WNS = WM == 3 ? 2 : 4;          // waves along the channels (64 each)
  static constexpr int BM = WM == 0 ? 64 : (WM == 3 ? 128 : 128 * WM);
  static constexpr int WIN = BM + 2 * kHalo;           // 264 | 136
  static constexpr int A_STAGE = WIN * ROWB;           // 33792 | 17408
  static constexpr int A_GROUPS = WIN / 8;             // 33 | 17 eight-row groups
  static constexpr int WAVES = (WM == 0 || WM == 3) ? 4 : 4 * WM;
  static constexpr int SCRATCH = MF * 32 * ROWB;       // epilogue scratch per wave: [MF * 32 frames][64 channels] bf16
  static constexpr int PIECES = (A_GROUPS + WAVES - 1) / WAVES;   // LDS-DMA pieces per wave per window: 5
  static constexpr int RING_BYTES = N_STAGES * A_STAGE; // 4 window stages (135168 | 69632 B) >= epilogue scratch, 16 KiB per wave
  static constexpr int PARAM_OFF = RING_BYTES;         // bias | scale | shift of the tile's 256 channels (3 KiB)
  static constexpr int LDS_BYTES = RING_BYTES + 3 * 256 * 4;
  static_assert(WAVES * SCRATCH <= RING_BYTES, "epilogue scratch must fit in the ring");
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// one LDS-DMA instruction (see kernels_tdnn_v2.hip for why this is inline asm)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
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

// same with a scalar base + 32-bit per-lane byte offset (no 64-bit VALU address arithmetic per piece)
__device__ __forceinline__ void glds16_s(const void *sbase, uint32_t voff, uint32_t lds_dst) {
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

// ABL: 0 full, 1 full + short loop for a partial last chunk (cin % 64 != 0), 2 MFMA only (no loads of any kind in the loop), 4 no epilogue stores,
//      5 full + per-workgroup phase timestamps into p.partial, 6 MFMA only + timestamps
#define STAMP3(k)                                                                                                   \
  if constexpr (ABL >= 5) {                                                                                         \
    if (tid == 0) {                                                                                                 \
      unsigned long long *dbg = reinterpret_cast<unsigned long long *>(p.partial) + (size_t)blockIdx.x * 8;          \
      dbg[k] = __builtin_amdgcn_s_memrealtime();                                                                    \
      if (k == 1) dbg[5] = __builtin_amdgcn_s_memtime();                                                          \
      if (k == 2) dbg[6] = __builtin_amdgcn_s_memtime();                                                          \
      if (k == 0) dbg[4] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
    }                                                                                                               \
  }
template <int ABL, bool GENERIC, bool POOL, int WM, int ET = ET_BF16>
__global__ __launch_bounds__(Geom3<WM>::WAVES * 64, WM == 0 ? 3 : 2) void tdnn_gemm_big3_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  using G = Geom3<WM>;
  constexpr bool MFMA_ONLY = (ABL == 2 || ABL == 6);
  constexpr bool NO_WF = MFMA_ONLY || (ABL >= 16 && (ABL & 1)), NO_X = MFMA_ONLY || (ABL >= 16 && (ABL & 2)), NO_DMA = MFMA_ONLY || (ABL >= 16 && (ABL & 4));
  constexpr bool X_DUMMY = (ABL >= 16 && (ABL & 8));   // LDS reads issued but their data never feeds an MFMA
  constexpr int BM = G::BM, A_STAGE = G::A_STAGE, A_GROUPS = G::A_GROUPS, MF = G::MF;
  static_assert(!(POOL && (WM == 0 || WM == 3)), "the fused pooling epilogue works on 128-row half tiles");
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / G::WNS, wn = wave % G::WNS;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = p.row_begin + (tile / n_tiles) * BM;
  const int n0 = (tile % n_tiles) * (G::WNS * 64);

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *zero = reinterpret_cast<const unsigned char *>(p.zero16);
  const size_t x_pitch = (size_t)p.ldx * 2;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte *)lds);
  const int g_row = lane >> 3, g_slot = lane & 7;

  const int nchunks = (p.cin_pad + BK - 1) / BK;
  const int n_taps = p.n_taps;
  STAMP3(0);
  if (p.tune) {
    // which of the (up to) two co-resident workgroups of this CU am I: the first one gets LDS base 0
    const int second = (__builtin_amdgcn_s_getreg(14342) != 0);
    const int prio = (p.tune >> (second ? 2 : 0)) & 3;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    if (second) for (int k = 0; k < ((p.tune >> 8) & 0xff); ++k) __builtin_amdgcn_s_sleep(127);   // ~127*64 cycles each
  }

  // Per-channel epilogue constants go to LDS now: fetching them from L2 inside the epilogue put a chain of
  // dependent ~1 us loads behind every tile (the accumulators leave no registers to prefetch them into).
  float *lds_par = reinterpret_cast<float *>(lds + G::PARAM_OFF);
  if (tid < 192) {
    const int which = tid >> 6, idx = (tid & 63) * 4;
    float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = (which == 0) ? p.bias : (which == 1 ? p.scale : p.shift);
    if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + n0 + idx);
    *reinterpret_cast<float4 *>(lds_par + which * 256 + idx) = v;
  }

  // feature window of chunk c -> ring stage st: 33 | 17 eight-row groups, one LDS-DMA instruction each.  Every
  // wave issues PIECES of them (the odd last group is issued by all waves - identical bytes - so that the
  // VMEM queue looks the same in every wave).  Rows beyond the ends of the matrix are clamped to its first /
  // last row, which are gap rows (zeros) in every frames / grid buffer; per piece only a 32-bit byte offset is
  // kept and the chunk advances the scalar base, so a refill costs ~4 instructions per piece.
  uint32_t a_off[G::PIECES];
#pragma unroll
  for (int i = 0; i < G::PIECES; ++i) {
    const int grp = min(wave + i * G::WAVES, A_GROUPS - 1);
    const int w = grp * 8 + g_row;
    const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
    a_off[i] = (uint32_t)row * (uint32_t)x_pitch + (uint32_t)swz(w, g_slot) * 16u;
  }
  auto issue_A = [&](int c, int st) {
    const unsigned char *base = xg + (size_t)c * (BK * 2);
    const bool tail = (c + 1) * BK > p.cin_pad;              // only the last chunk of a cin that is not a multiple of 64
#pragma unroll
    for (int i = 0; i < G::PIECES; ++i) {
      const int grp = min(wave + i * G::WAVES, A_GROUPS - 1);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + st * A_STAGE + grp * 1024);
      if (!tail) {
        glds16_s(base, a_off[i], dst);
      } else {
        const int w = grp * 8 + g_row;
        const bool ok = c * BK + swz(w, g_slot) * 8 < p.cin_pad;
        glds16(ok ? base + a_off[i] : zero, dst);
      }
    }
  };

  // weight fragments: this wave's two 32-channel fragments, [tap][chunk][k-group] blocks of 1 KiB
  const size_t frag_stride = (size_t)n_taps * nchunks * 4096;            // bytes per 32-channel fragment
  const unsigned char *wf_base0 = reinterpret_cast<const unsigned char *>(p.wfrag) + (size_t)((n0 + wn * 64) / 32) * frag_stride + (size_t)lane * 16;
  const unsigned char *wf_base1 = wf_base0 + frag_stride;
  uint4 wf[4][2];
  auto load_wf = [&](int kg, int c, int t) {
    const size_t off = ((size_t)t * nchunks + c) * 4096 + (size_t)kg * 1024;
    wf[kg][0] = *reinterpret_cast<const uint4 *>(wf_base0 + off);
    wf[kg][1] = *reinterpret_cast<const uint4 *>(wf_base1 + off);
  };

  // Plain epilogue (ABL 3 = the first form, for in-process A/B): the gap-row mask is applied to the packed bf16 pairs (2 selects
  // per 4 values instead of 4) - a saturating MFMA stream leaves the SIMD's VALU no issue slot, so every VALU instruction of an
  // epilogue is paid in full (LABLOG.md, round 2 item 1).  Starting the accumulators from the bias saves another 128 additions
  // per wave tile (736 -> 449 VALU operations, +0.2 % / +0.6 % per x-vector / ECAPA step in the A/B of profiles/r2q_*) but moves
  // the bias to the front of the f32 sum: the outputs are then no longer bit-identical to the generic tile's, which layer
  // shapes beyond this kernel's 32-bit row offsets fall back to - an utterance's bits would depend on the batch it is in.
  // Not taken: BIAS_IN_ACC stays off.
  constexpr bool PACKED_MASK = !GENERIC && !POOL && ABL != 3;
  constexpr bool BIAS_IN_ACC = false;
  f32x16_t acc[MF][2];
  if constexpr (!BIAS_IN_ACC) {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  struct XFrags { uint4 x[MF]; };
  auto load_x = [&](const unsigned char *Ab, int d, int kg, XFrags &f) {
    const int slot = kg * 2 + lh;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int w = wm * (MF * 32) + i * 32 + lr + kHalo + d;
      f.x[i] = *reinterpret_cast<const uint4 *>(Ab + w * ROWB + swz(w, slot) * 16);
    }
  };
  auto load_x1 = [&](const unsigned char *Ab, int d, int kg, int i, XFrags &f) {
    const int w = wm * (MF * 32) + i * 32 + lr + kHalo + d;
    f.x[i] = *reinterpret_cast<const uint4 *>(Ab + w * ROWB + swz(w, kg * 2 + lh) * 16);
  };
  auto mma2 = [&](const XFrags &f, int kg, int j, int i0) {
#pragma unroll
    for (int i = i0; i < i0 + 2; ++i)
      acc[i][j] = mfma16<ET>(wf[kg][j], f.x[i], acc[i][j]);
  };
  // ---- prologue: three windows in flight, weight fragments of step 0.  Only window 0 and the fragments are
  // waited for: VMEM retires in order, so vmcnt(2 * PIECES) leaves exactly windows 1 and 2 outstanding.
  // Tried and dropped (r2i, in-process A/B): the fragment fetches as inline-asm loads with counted s_waitcnt and the window DMA
  // issued behind a step's last fetch - what doubled the rate of the Res2NetBlock kernel's K loop (kernels_res2.hip: there the
  // compiler-visible loads sat right behind a 48 KiB DMA).  Here the window is fetched three chunks ahead and hipcc's own
  // waits next to the pinned MFMA / ds_read interleave are the better schedule: 711 vs 721 us per x-vector step, 3181 vs
  // 3334 us per ECAPA step in favour of the compiler-visible loads (profiles/r2i_ab_big3_*.txt); same on the chain kernel.
  // Tried and dropped (r2e, in-process A/B on ECAPA): 1-tap layers meeting at the workgroup barrier every SECOND chunk (all four
  // stages in flight, two windows read between barriers): 3556 vs 3532 us per step - the per-chunk barrier is not what holds
  // the 1024 -> 1024 layers at 0.37 of the bf16 peak.
  issue_A(0, 0);
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) load_wf(kg, 0, 0);
  if (nchunks > 2) {
    issue_A(1, 1);
    issue_A(2, 2);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::PIECES) : "memory");
  } else {
    if (nchunks > 1) issue_A(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (BIAS_IN_ACC) {
    // acc[i][j][4 q + e] belongs to channel wn*64 + j*32 + 8 q + 4 lh + e of the tile, for every frame fragment i
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + wn * 64 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int i = 0; i < MF; ++i) {
          acc[i][j][q * 4 + 0] = b4.x; acc[i][j][q * 4 + 1] = b4.y; acc[i][j][q * 4 + 2] = b4.z; acc[i][j][q * 4 + 3] = b4.w;
        }
      }
  }

  STAMP3(1);
  // tap offsets live in one VGPR (lane t = tap t) and are fetched with v_readlane: indexing the kernel-argument
  // array costs an s_load + s_waitcnt lgkmcnt(0) per step, and that wait also drains every LDS read in flight
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  const int d_first = __builtin_amdgcn_readlane(v_taps, 0);
  int s = 0;
  XFrags x0, x1;
  load_x(lds, d_first, 0, x0);                               // the only LDS latency nothing hides
  // chunks of 64 input channels; a last chunk with only 16 / 32 / 48 of them (cin = 80: the fbank input layer)
  // runs 1 / 2 / 3 k-groups per tap in the short loop behind this one instead of 4 mostly-zero ones
  const int tail_groups = (ABL == 1) ? ((p.cin_pad % BK) / 16) : 0;    // ABL 1 = the instantiation with the short tail loop
  const int nfull = tail_groups ? nchunks - 1 : nchunks;
  for (int c = 0; c < nfull; ++c) {
    const unsigned char *Ab = lds + (c % N_STAGES) * A_STAGE;
    for (int t = 0; t < n_taps; ++t, ++s) {
      // Fragments of the NEXT step are fetched unconditionally (the last step re-reads its own: valid memory,
      // never used).  A conditional fetch makes the compiler assume the short queue at every wait and the
      // resulting vmcnt(1)/vmcnt(0) stall each step on loads issued a few instructions earlier.
      const bool last_tap = (t + 1 == n_taps);
      int cn = c, tn = t + 1;
      if (last_tap) { tn = 0; cn = c + 1; }
      if (cn == nchunks) { cn = c; tn = t; }
      const int d = __builtin_amdgcn_readlane(v_taps, t);
      // k-group g = the 8 MFMAs of g with the 4 LDS reads of group g+1 (each with its 2 address VALU ops) and
      // the 2 fragment fetches of the next step's group g threaded between them: issued as one block after
      // the MFMAs they cost their full issue time (tools/loop_probe.hip: 324 vs 271 cycles per group), one
      // read in front of every MFMA pair hides completely.  sched_barrier(0) after every pair pins exactly this
      // order (sched_group_barrier picks MFMAs in an order of its own and the first MFMA of the next group
      // then waits for the read issued last).
      XFrags xd;
      auto sink = [&]() { if (X_DUMMY) asm volatile("" ::"v"(xd.x[0].x), "v"(xd.x[1].y), "v"(xd.x[MF - 2].z), "v"(xd.x[MF - 1].w)); };
      auto group = [&](const XFrags &xc, int kg, XFrags &xn, const unsigned char *An, int dn, int kgn) {
        const size_t woff = ((size_t)tn * nchunks + cn) * 4096 + (size_t)kg * 1024;
#pragma unroll
        for (int q = 0; q < MF; ++q) {
          // pair q: channel fragment j = q / (MF/2) against frame fragments 2 * (q % (MF/2)) and + 1; a channel
          // fragment is re-fetched (for the next step) as soon as its last MFMA has been issued
          if (!NO_X) load_x1(An, dn, kgn, q, X_DUMMY ? xd : xn);
          mma2(xc, kg, q / (MF / 2), (q % (MF / 2)) * 2);
          if (!NO_WF && q == MF / 2 - 1) wf[kg][0] = *reinterpret_cast<const uint4 *>(wf_base0 + woff);
          if (!NO_WF && q == MF - 1) wf[kg][1] = *reinterpret_cast<const uint4 *>(wf_base1 + woff);
          __builtin_amdgcn_sched_barrier(0);
        }
        sink();
        if constexpr (ABL == 5) {
          if (tid == 0 && s < 32) {
            unsigned long long *dbg2 = reinterpret_cast<unsigned long long *>(p.partial) + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 128;
            dbg2[s * 4 + kg] = __builtin_amdgcn_s_memtime();
          }
        }
      };
      group(x0, 0, x1, Ab, d, 1);
      group((NO_X || X_DUMMY) ? x0 : x1, 1, x0, Ab, d, 2);
      group(x0, 2, x1, Ab, d, 3);
      // The last k-group already reads the first fragments of the next step.  When that step opens a new chunk
      // the workgroup meets first: every wave has its pieces of window c+1 in LDS (they are older than the
      // youngest 8 VMEM operations, all fragment fetches), and behind the barrier nobody reads window c-1
      // any more, so its stage takes window c+3.  The barrier sits in the shadow of group 2's MFMAs.
      const unsigned char *An = Ab;
      int dn = __builtin_amdgcn_readlane(v_taps, last_tap ? t : t + 1);
      if (last_tap && c + 1 < nchunks) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!NO_DMA && c + 3 < nchunks) issue_A(c + 3, (c + 3) % N_STAGES);
        An = lds + ((c + 1) % N_STAGES) * A_STAGE;
        dn = d_first;
      }
      group((NO_X || X_DUMMY) ? x0 : x1, 3, x0, An, dn, 0);
    }
  }
  if (ABL == 1 && tail_groups) {
    // x0 and wf[0] already hold (tail chunk, tap 0, k-group 0) - prefetched by the last full step, or by the
    // prologue when the layer has no full chunk; the rest alternates between the two register sets
    const int c = nfull;
    const unsigned char *Ab = lds + (c % N_STAGES) * A_STAGE;
    const int n_it = n_taps * tail_groups;
    // two iterations per trip, straight-line: every MFMA is unconditional (an accumulator touched inside a branch
    // costs the register allocator copies of all 128 of them); an iteration that does not exist gets zero weights
    auto prefetch = [&](int i, XFrags &xn, uint4 &w0, uint4 &w1) {
      const bool live = i < n_it;
      const int ii = live ? i : n_it - 1;
      const int t2 = ii / tail_groups, kg2 = ii - t2 * tail_groups;
      load_x(Ab, __builtin_amdgcn_readlane(v_taps, t2), kg2, xn);
      const size_t woff = ((size_t)t2 * nchunks + c) * 4096 + (size_t)kg2 * 1024;
      w0 = *reinterpret_cast<const uint4 *>(wf_base0 + woff);
      w1 = *reinterpret_cast<const uint4 *>(wf_base1 + woff);
      if (!live) { w0 = make_uint4(0, 0, 0, 0); w1 = make_uint4(0, 0, 0, 0); }
    };
    for (int it = 0; it < n_it; it += 2) {
      prefetch(it + 1, x1, wf[1][0], wf[1][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < MF; ++q) mma2(x0, 0, q / (MF / 2), (q % (MF / 2)) * 2);
      __builtin_amdgcn_sched_barrier(0);
      prefetch(it + 2, x0, wf[0][0], wf[0][1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < MF; ++q) mma2(x1, 1, q / (MF / 2), (q % (MF / 2)) * 2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done reading the ring: the epilogue reuses the LDS
  asm volatile("" ::: "memory");
  STAMP3(2);

  if constexpr (POOL) {
    // ---- epilogue with fused statistics pooling (pooling.py:58-67 folded into the producing layer):
    // the layer's output never reaches HBM.  Each wave owns 128 frames x 64 channels.  Per 32-frame
    // fragment it writes u = max(acc + b, lo) * scale (the output minus the BN shift: >= 0, well
    // conditioned for one-pass moments) as f32 to an LDS scratch [32 frames][64 ch], then lane = channel
    // sums the 32 rows.  The row -> segment lookup is wave-uniform, so a segment boundary is a scalar
    // branch: flush (sum (u - pv), sum (u - pv)^2, pivot pv) of the finished segment to the partial buffer
    //   P[half-tile of 128 rows][segment slot][3 stats][channel]
    // and continue.  pool_finish_kernel adds the half-tiles of each segment in row order.
    constexpr int SPITCH = 68;                       // floats per scratch row: 64 + 4 keeps ds_write_b128 conflict-free
    float *scrf = reinterpret_cast<float *>(lds + wave * (32 * SPITCH * 4));
    const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
    const int rbase_w = m0 + wm * 128;
    const int half = rbase_w >> 7;
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && rbase_w + k < p.rows) first_seg = p.row_seg[rbase_w + k];
    const int ch_l = n0 + wn * 64 + lane;
    const int rowseg_lo = p.row_seg[rbase_w + lane], rowseg_hi = p.row_seg[rbase_w + 64 + lane];
    // One-pass moments are taken about a pivot (the segment's first value in this half tile), the way the separate
    // pooling kernel does it: sum (u - pv) and sum (u - pv)^2 stay small for a channel that barely moves in time, where
    // sum u^2 - (sum u)^2 / T would cancel (the reference is two-pass, pooling.py:58-66).  pool_finish_kernel merges the
    // (count, mean, M2) of the half tiles of a segment with the pairwise update of Chan et al.
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
    for (int i = 0; i < 4; ++i) {
      const bool valid = (p.row_valid[(rbase_w + i * 32) >> 5] >> lr) & 1u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;             // channel inside the tile
          const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
          const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
          float4 u;
          u.x = valid ? fmaxf(acc[i][j][q * 4 + 0] + b4.x, act_lo) * sc4.x : 0.0f;
          u.y = valid ? fmaxf(acc[i][j][q * 4 + 1] + b4.y, act_lo) * sc4.y : 0.0f;
          u.z = valid ? fmaxf(acc[i][j][q * 4 + 2] + b4.z, act_lo) * sc4.z : 0.0f;
          u.w = valid ? fmaxf(acc[i][j][q * 4 + 3] + b4.w, act_lo) * sc4.w : 0.0f;
          *reinterpret_cast<float4 *>(scrf + lr * SPITCH + j * 32 + 8 * q + 4 * lh) = u;
        }
      // column sums of this fragment; rows are consumed in order, segments are contiguous in rows.
      // rowseg_lo/hi hold row_seg of the wave's 128 rows (lane l: rows l and 64 + l), so the per-row
      // segment id is a v_readlane with a constant lane: no memory access in the loop.
      const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
      // ~80 % of the 32-frame fragments of a 200-frame batch lie inside one utterance (gap rows belong to nobody and
      // are skipped): those are summed branch-free; a fragment with a gap row or two utterances walks its rows one by one
      const unsigned long long in_frag = 0xffffffffull << ((i & 1) * 32);
      const unsigned long long m_valid = __builtin_amdgcn_ballot_w64(rs_vec >= 0) & in_frag;
      if (m_valid == 0) continue;                                                  // gap rows only
      const int sg0 = __builtin_amdgcn_readlane(rs_vec, __builtin_ctzll(m_valid));
      const unsigned long long m_same = __builtin_amdgcn_ballot_w64(rs_vec == sg0) & in_frag;
      if (m_same == in_frag) {
        if (sg0 != cur_seg) {
          if (cur_seg >= 0) flush();
          cur_seg = sg0; ps = 0.0f; pq = 0.0f; pv = scrf[lane];
        }
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          const float d = scrf[r * SPITCH + lane] - pv;
          ps += d;
          pq = fmaf(d, d, pq);
        }
      } else {
        // rolled on purpose: unrolled, the four fragments' row-by-row code with its flush() copies is tens of KiB of
        // instructions around the fast path (instruction-cache misses in every epilogue)
#pragma unroll 1
        for (int r = 0; r < 32; ++r) {
          const int sg = __builtin_amdgcn_readlane(rs_vec, (i & 1) * 32 + r);          // wave-uniform
          if (sg < 0) continue;                                                          // gap row
          const float v = scrf[r * SPITCH + lane];
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
    if constexpr (ABL >= 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      STAMP3(3);
    }
    return;
  }
  // ---- epilogue --------------------------------------------------------------------------
  // acc[i][j][r]: frame = m0 + wm*128 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  unsigned char *scr = lds + wave * G::SCRATCH;  // [MF * 32 frames][64 channels] bf16, 128-B rows, swizzled slots
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  uint32_t vmask = 0;                            // bit i: this lane's frame of m-fragment i is a real frame
#pragma unroll
  for (int i = 0; i < MF; ++i) vmask |= ((p.row_valid[(m0 + wm * (MF * 32) + i * 32) >> 5] >> lr) & 1u) << i;
  // generic epilogue only: per-utterance bias rows (the hoisted global-context part of ECAPA's attention layer,
  // ecapa_tdnn_xvector.py:176-181), one float4 per (frame, 4 channels) from L2; tanh / sigmoid on v_exp_f32 + v_rcp_f32 (the
  // output is rounded to bf16; libm's tanhf was ~60 VALU operations per value - more than the tile's whole main loop)
  const float *segb[MF];
  auto act_fast = [](float v, int act) -> float {
    switch (act) {
      case ASV_ACT_RELU: return fmaxf(v, 0.0f);
      case ASV_ACT_TANH: return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * v));        // exp overflow -> rcp(inf) = 0 -> 1
      case ASV_ACT_SIGMOID: return __frcp_rn(1.0f + __expf(-v));
      default: return v;
    }
  };
  if constexpr (GENERIC) {
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      segb[i] = nullptr;
      if (p.seg_bias != nullptr && ((vmask >> i) & 1u))
        segb[i] = p.seg_bias + (size_t)p.row_seg[m0 + wm * (MF * 32) + i * 32 + lr] * p.ld_segbias + n0;
    }
  }
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
      for (int i = 0; i < MF; ++i) {
        const bool valid = (vmask >> i) & 1u;
        const int frow = i * 32 + lr;            // row inside the wave's scratch tile
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) {
            float z = acc[i][j][q * 4 + e] + b[e];
            if (segb[i] != nullptr && n0 + chl + e < p.cout_store) z += segb[i][chl + e];
            z = p.affine_first ? act_fast(z * sc[e] + sh[e], p.act1) : act_fast(z, p.act1) * sc[e] + sh[e];
            z = act_fast(z, p.act2);
            y[e] = valid ? z : 0.0f;
          } else if constexpr (BIAS_IN_ACC) {
            y[e] = fmaf(max_lo(acc[i][j][q * 4 + e], act_lo), sc[e], sh[e]);
          } else {
            y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], PACKED_MASK ? true : valid);
          }
        }
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        if constexpr (PACKED_MASK) {               // gap rows are zeros: on the packed pairs, 2 selects per 4 values
          pk.x = valid ? pk.x : 0u;
          pk.y = valid ? pk.y : 0u;
        }
        // odd rows keep their two 8-byte halves swapped so rows r, r+1 (same slot) hit different banks
        *reinterpret_cast<uint2 *>(scr + frow * ROWB + swz(frow, slot) * 16 + ((lh ^ (frow & 1)) * 8)) = pk;
      }
    }
  }
  // the scratch tile belongs to this wave only: LDS ops of one wave complete in order
  {
    unsigned char *yg = reinterpret_cast<unsigned char *>(p.y);
    const size_t y_pitch = (size_t)p.ldy * 2;
#pragma unroll
    for (int it = 0; it < MF * 4; ++it) {
      const int piece = it * 64 + lane, frow = piece >> 3, slot = piece & 7;
      uint4 v = *reinterpret_cast<const uint4 *>(scr + frow * ROWB + swz(frow, slot) * 16);
      if (frow & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      const int ch = n0 + wn * 64 + slot * 8;
      const int row = m0 + wm * (MF * 32) + frow;
      if (ABL == 4) { asm volatile("" ::"v"(v.x), "v"(v.w)); continue; }
      if (ch < p.cout_store) *reinterpret_cast<uint4 *>(yg + (size_t)row * y_pitch + (size_t)ch * 2) = v;
    }
  }
  if constexpr (ABL >= 5) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    STAMP3(3);
  }
}

}  // namespace

bool tdnn_big3_supported(const TdnnKernelParams &p, int et, bool out_f32) {
  const bool bf16 = et != ET_F32;      // either 16-bit element type (p.et selects the instantiation at launch)
  // the window refill addresses rows with 32-bit byte offsets from a scalar base: activation matrices of 4 GiB and more
  // (2 M rows x 1536 channels) go to the kernels with 64-bit addressing
  const bool fits32 = (unsigned long long)p.rows * (unsigned long long)p.ldx * 2ull < (1ull << 32);
  // Tried and dropped (r2c): 128-channel layers with a deep K (ECAPA's attention bottleneck 1536 -> 128, the im2col'd 576 -> 128
  // convolution) on half-filled 256-channel tiles: 600 workgroups are 1.17 rounds of the 512 resident ones and every one of
  // them multiplies 128 channels of zeros: 231 us against 172 us on the 128 x 128 register-staged tile.
  // 97..128 output channels with K >= 512: the 128 x 128 geometry; a per-utterance bias (seg_bias) is part of the generic
  // epilogue (the fused pooling has the plain one)
  const bool wide_enough = p.cout_store >= 192 || (p.cout_store > 96 && p.cout_store <= 128 && p.cin_pad * p.n_taps >= 512 && p.pool_partial == nullptr);
  return p.wfrag != nullptr && fits32 && bf16 && !out_f32 && p.x2 == nullptr && (p.seg_bias == nullptr || (p.pool_partial == nullptr && p.row_seg != nullptr)) && p.seg_scale == nullptr && p.res == nullptr && p.zero16 != nullptr && p.rows % 256 == 0 && p.cout_store % 8 == 0 && wide_enough && p.cin_pad >= 64;
}

// variant = geometry * 100 + ablation code; geometry 0: 128 x 256 tiles (two workgroups per CU), 1: 256 x 256 (one),
// 2: 64 x 256 (three)
int launch_tdnn_big3_variant(const TdnnKernelParams &p, int variant, hipStream_t s) {
  ASV_REQUIRE(p.rows % 256 == 0, "tdnn(big3): rows %d not a multiple of 256", p.rows);
  ASV_REQUIRE(p.wfrag != nullptr, "tdnn(big3): fragment-packed weights missing");
  const int geom = variant / 100;
  variant %= 100;
  const int bm = (geom == 0 || geom == 3) ? 128 : (geom == 1 ? 256 : 64);
  const int row_count = p.row_count > 0 ? p.row_count : p.rows;
  ASV_REQUIRE(p.row_begin % bm == 0 && row_count % bm == 0 && p.row_begin + row_count <= p.rows, "tdnn(big3): row range [%d, +%d) does not fit %d-row tiles", p.row_begin, row_count, bm);
  const int bne = geom == 3 ? 128 : BN;
  const int m_tiles = row_count / bm, n_tiles = round_up(p.cout_store, bne) / bne;
  const dim3 grid(m_tiles * n_tiles), block(geom == 1 ? 512 : 256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr;
  const bool f16 = p.et == ET_F16;       // IEEE-half operands / rows: the production forms only (the ablation and A/B instantiations are bf16)
  ASV_REQUIRE(geom != 3 || p.pool_partial == nullptr, "tdnn(big3): the 128 x 128 geometry has no fused pooling form");
  if (p.pool_partial != nullptr) {
    ASV_REQUIRE(fast && p.row_seg != nullptr && p.pool_slots >= 1 && geom != 2, "tdnn(big3): fused pooling needs the plain epilogue, a row map and 128-row wave tiles");
#ifdef ASV_WITH_ABLATION
    if (geom == 0 && variant == 5) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<5, false, true, 1>), grid, block, 0, s, p, m_tiles, n_tiles);
    else
#endif
    if (geom == 0 && f16) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 1, ET_F16>), grid, block, 0, s, p, m_tiles, n_tiles);
    else if (geom == 0) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 1>), grid, block, 0, s, p, m_tiles, n_tiles);
    else if (f16) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 2, ET_F16>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 2>), grid, block, 0, s, p, m_tiles, n_tiles);
    ASV_HIP_CHECK(hipGetLastError());
    return ASV_OK;
  }
  // The ablation instantiations (parts of the kernel compiled out: tools/gemm_ablate.hip, `make tools`) are built into the
  // tools' own object only (-DASV_WITH_ABLATION); the library carries the production forms and the one A/B variant (3).
#define ASV_BIG3(ABLV, GENV, WMV) do { if (f16 && (ABLV) <= 1) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<((ABLV) <= 1 ? (ABLV) : 0), GENV, false, WMV, ET_F16>), grid, block, 0, s, p, m_tiles, n_tiles); \
                                       else hipLaunchKernelGGL((tdnn_gemm_big3_kernel<ABLV, GENV, false, WMV>), grid, block, 0, s, p, m_tiles, n_tiles); } while (0)
  const bool tail = fast && p.cin_pad % BK != 0;
  if (geom == 0) {
    switch (variant) {
#ifdef ASV_WITH_ABLATION
      case 2: ASV_BIG3(2, false, 1); break;
      case 4: ASV_BIG3(4, false, 1); break;
      case 5: ASV_BIG3(5, false, 1); break;
      case 6: ASV_BIG3(6, false, 1); break;
      case 17: ASV_BIG3(17, false, 1); break;
      case 18: ASV_BIG3(18, false, 1); break;
      case 20: ASV_BIG3(20, false, 1); break;
      case 19: ASV_BIG3(19, false, 1); break;
      case 21: ASV_BIG3(21, false, 1); break;
      case 24: ASV_BIG3(24, false, 1); break;
      case 29: ASV_BIG3(29, false, 1); break;
      case 22: ASV_BIG3(22, false, 1); break;
#endif
      case 3:
        if (fast && !tail && !f16) { ASV_BIG3(3, false, 1); break; }       // A/B: the first form of the plain epilogue
        [[fallthrough]];
      default:
        if (tail) ASV_BIG3(1, false, 1);
        else if (fast) ASV_BIG3(0, false, 1);
        else ASV_BIG3(0, true, 1);
    }
  } else if (geom == 3) {
    if (tail) ASV_BIG3(1, false, 3);
    else if (fast) ASV_BIG3(0, false, 3);
    else ASV_BIG3(0, true, 3);
  } else if (geom == 1) {
    switch (variant) {
#ifdef ASV_WITH_ABLATION
      case 2: ASV_BIG3(2, false, 2); break;
      case 4: ASV_BIG3(4, false, 2); break;
#endif
      default:
        if (fast) ASV_BIG3(0, false, 2);
        else ASV_BIG3(0, true, 2);
    }
  } else {
    switch (variant) {
#ifdef ASV_WITH_ABLATION
      case 2: ASV_BIG3(2, false, 0); break;
      case 5: ASV_BIG3(5, false, 0); break;
      case 6: ASV_BIG3(6, false, 0); break;
#endif
      default:
        if (tail) ASV_BIG3(1, false, 0);
        else if (fast) ASV_BIG3(0, false, 0);
        else ASV_BIG3(0, true, 0);
    }
  }
#undef ASV_BIG3
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

// Geometry choice.  64-row tiles (three workgroups per CU) fix the round quantisation on paper (816 tiles of 128 rows =
// 2 rounds at 80 %, 1632 tiles of 64 rows = 3 half rounds) and their MFMA-only skeleton is 17 % faster, but every wave
// then fetches its 8 KiB of weight fragments per step for 16 instead of 32 MFMAs: at three waves per SIMD that is
// ~62 B/clk/CU from L2, the L1 fill limit, and the full kernel is 3 % slower on the C2 shapes (tools/gemm_ablate).
// They win when the 128-row tiles cannot occupy the chip: measured 24.6 -> 17.5 us at 6656 rows x 512 channels,
// break-even at 13056 rows.
int tdnn_big3_pick_geometry(const TdnnKernelParams &p) {
  if (p.big_one_per_cu) return 1;
  if (p.pool_partial != nullptr) return 0;
  if (p.cout_store <= 128) return 3;
  const long long n_tiles = round_up(p.cout_store, BN) / BN;
  const long long t128 = (long long)(p.rows / 128) * n_tiles;
  return t128 <= 160 ? 2 : 0;
}

// Default entry.  Tried and dropped: giving the rows of a poorly filled last round (C2: 816 tiles = 512 + 304) to the 64-row
// geometry in a second launch (one full round at two workgroups per CU, then a half-height round at three).  The second
// kernel cannot start before the first has drained, the overlap between the tail of round one and the head of round two is
// lost, and the result is 4-6 us slower on every C2 layer (89 vs 85 us on the 3-tap 512 -> 512 layer).  The row-range
// parameters (row_begin / row_count) stay for tools/gemm_ablate.
int launch_tdnn_big3(const TdnnKernelParams &p, hipStream_t s) {
  // developer aid (in-process A/B, tools/chain_ab.py): read once per process; at every launch only with ASV_AMD_LIVE_TUNE=1
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  static const char *tune0 = getenv("ASV_AMD_BIG3_TUNE");
  const char *tune = live ? getenv("ASV_AMD_BIG3_TUNE") : tune0;
  if (tune != nullptr && p.tune == 0) {
    TdnnKernelParams q = p;
    q.tune = atoi(tune);
    const int geom = tdnn_big3_pick_geometry(q);
    return launch_tdnn_big3_variant(q, geom * 100 + ((q.tune & 0x20000) && geom == 0 && q.pool_partial == nullptr ? 3 : 0), s);
  }
  // ASV_AMD_BIG3_GEOM=1|2 (A/B aid): 256 x 256 tiles, one workgroup per CU / 64 x 256, three per CU, for the layers the default
  // 128 x 256 geometry would take
  static const char *geom0 = getenv("ASV_AMD_BIG3_GEOM");
  const char *geom_s = live ? getenv("ASV_AMD_BIG3_GEOM") : geom0;
  int geom = tdnn_big3_pick_geometry(p);
  if (geom_s != nullptr && geom == 0 && p.pool_partial == nullptr && (atoi(geom_s) == 1 || atoi(geom_s) == 2)) geom = atoi(geom_s);
  return launch_tdnn_big3_variant(p, geom * 100, s);
}

}  // namespace asv
