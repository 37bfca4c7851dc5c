// The chain kernel of kernels_tdnn_chain.hip (tdnn3 -> tdnn4 -> tdnn5 -> StatisticsPooling of the standard x-vector, one
// 128-frame tile resident in LDS: model/xvector.py:77-98; components.py:107-149, 410-431; pooling.py:58-67) as FOUR waves of
// 512 registers, one per SIMD, instead of eight waves of 256 - the design VERDICT r2 item 3 (i) / (iv) asked for, built,
// measured, and NOT the default (ASV_AMD_CHAIN_WAVES=4 selects it; tools/chain_ab.py ASV_AMD_CHAIN_WAVES 8 4 is the A/B):
//   * 512-wide layers: a wave owns 128 frames x 128 channels - 4 x 4 accumulator tiles = all 256 AGPRs; per 16-deep K step 4
//     LDS fragment reads + 4 weight-fragment fetches feed 16 matrix instructions (the 8-wave form: 4 + 2 for 8);
//   * last layer: 64-channel units; the pooling arithmetic of unit u - plain VALU operations, branch-free per 32-frame
//     fragment, spread over the slots behind the matrix instructions of unit u + 1's K loop (one value per two instructions,
//     no slot with two dependent operations) - reads unit u's results from a second accumulator set, to which they were
//     copied by eight matrix instructions (0 . 0 + C).  Two forms of every K chunk: plain (the fragment lies inside one
//     utterance: 5 operations per value) and two masked runs (a seam or gap rows: 11).  The accumulators start from the
//     inline constant 0 and the bias joins in the pooling arithmetic.  The wave's last unit is drained after the loop.
// What the measurements say (profiles/r3k_chain4_experiments.txt, r3k_mfma_probe.txt; 640 x 200 frames, cycles per tile):
//                              8 waves      4 waves
//     first window               6.3 k       8 - 11 k   (one wave per SIMD: nothing covers the first fetches; the staging of the
//                                                        last layer's bias / scale in LDS adds 2.5 k)
//     layer A K loop            63.1 k       59.0 k     (49.2 k of matrix issue)
//     store Y (twice)            8.6 k        8.7 k
//     middle K loop             20.5 k       21.0 k
//     last layer                76.2 k       80 k + 8 k drain
//     step                     704 - 731     710 - 764 us   -> 0 - 1 % slower on every box of the pool
//   A lone wave's matrix stream runs at 32.0 cycles per instruction with any operand pattern, 32.8 with one AGPR read + 4 plain
//   VALU operations per two instructions (tools/mfma_bank_probe.hip); the plain chunks of the last layer reach 33.7 in the
//   kernel (ablation 6: 8.6 k per unit) - the design's premise holds.  What it does not survive is everything BETWEEN the
//   instructions, which a second wave per SIMD hides and a lone wave exposes:
//     * every global_load with a 64-bit address per lane: ~35 cycles of idle matrix pipe (-> buffer loads: 65 k -> 59 k in layer A);
//     * a chain of dependent VALU operations inside one slot stalls the next matrix instruction (-> software pipeline over slots);
//     * hipcc sinks the moment sums of a chunk out of the loop into the branch that consumes them (-> inline assembly);
//     * two forms of the chunk behind a branch: hipcc assigns their fragment registers differently, and at every join its
//       wait-count pass must assume the union of both paths' pending fetches: s_waitcnt vmcnt(1) in front of an LDS read whose
//       destination was a weight fragment on the other path - a full L2 round trip per chunk.  With only ONE form in the code
//       a unit takes 10.4 k (plain) or 12.0 k (masked) cycles, with both 12.5 - 13.4 k - against 12.7 k per unit-equivalent
//       for the 8-wave kernel, whose partner waves absorb all of this;
//     * the run logic between chunks (~250 cycles per chunk: scalar branches, the pivot, publication) and the drain of the
//       wave's last unit (8 k) have nothing to hide behind either.
//   Hand-allocating the arch VGPRs as well (as the AGPRs are here) would remove the join problem; that is an assembly kernel,
//   not a HIP one.
// Same parameters, weight fragments, pooling partials ([tile][slot][lane half][3][channel]) and debug stamps as the 8-wave
// kernel; results agree with it to f32 rounding of the pooled moments (tests/test_gpu_xvector.py).
#include <cstdlib>
#include <utility>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int CM = 128;                   // frames per workgroup
constexpr int CN = kChainWidth;           // channels of the resident tile (512)
constexpr int CBK = 64;
constexpr int CROWB = 128;                // window row: 64 16-bit elements
constexpr int CSTAGES = 4;
constexpr int CWIN = CM + 2 * kHalo;      // 136
constexpr int CSTAGE = CWIN * CROWB;      // 17408 B
constexpr int CGROUPS = CWIN / 8;         // 17 eight-row DMA pieces
constexpr int NWAVES = 4;
constexpr int CPIECES = (CGROUPS + NWAVES - 1) / NWAVES;   // 5 per wave
constexpr int YROWB = CN * 2;             // 1024 B per Y row
constexpr int Y_BYTES = CM * YROWB;       // 131072
constexpr int SCR_OFF = Y_BYTES;          // biases of the 512-wide layers (3 x 2 KiB, staged once) | scale | shift of the layer in flight (unfolded BatchNorm only)
constexpr int LAST_OFF = Y_BYTES + 3 * 2048 + 2 * 2048;   // bias | BN scale of the last layer (cout_pad floats each), staged once
constexpr int CHAIN4_LDS = 163840;
constexpr int kMaxLastWidth = (CHAIN4_LDS - LAST_OFF) / 8;   // 2816 output channels
static_assert(CSTAGES * CSTAGE <= Y_BYTES, "the window ring lives inside the Y region");

typedef __attribute__((address_space(3))) unsigned char chain_lds_byte;
template <int V> struct IC { static constexpr int value = V; };
template <int... Is, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(IC<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

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

// The accumulators are the wave's 256 AGPRs, managed BY HAND: tile T (0..15) = a[16 T .. 16 T + 15], every matrix instruction, every
// read and every write of them is inline assembly naming the registers.  (Left to hipcc, the two alternating accumulator sets of
// the last layer ended up half in VGPRs, shuffled through thousands of v_accvgpr_mov, with 300 - 450 spilled registers.)
// 512-wide layers: tile i * 4 + j = frames i * 32.., channels j * 32.. of the wave's 128; last layer: j = 2 * set + channel half.
// The compiler does not know these are matrix instructions, so the hazards it would cover are handled here: mfma_settle()
// between the last matrix instruction and VALU reads of its result (CDNA3 ISA 4.5: 11 wait states for 8 passes, 18 for 16), a
// short s_nop between v_accvgpr_write and a matrix instruction reading it.  Nothing else may live in AGPRs: the build checks
// that the kernel has no spills and that every AGPR mention sits in these helpers (tools/kernel_resources.py).
typedef unsigned int chain_u32x4 __attribute__((ext_vector_type(4)));
template <int ET, int T>
__device__ __forceinline__ void mfma_tile(const uint4 a, const uint4 b) {                  // tile T += a . b
  const chain_u32x4 va = __builtin_bit_cast(chain_u32x4, a), vb = __builtin_bit_cast(chain_u32x4, b);
  if constexpr (ET == ET_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(va), "v"(vb), "n"(T * 16), "n"(T * 16 + 15));
  else asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(va), "v"(vb), "n"(T * 16), "n"(T * 16 + 15));
}
template <int ET, int T>
__device__ __forceinline__ void mfma_tile_zero(const uint4 a, const uint4 b) {             // tile T = a . b
  const chain_u32x4 va = __builtin_bit_cast(chain_u32x4, a), vb = __builtin_bit_cast(chain_u32x4, b);
  if constexpr (ET == ET_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, 0" ::"v"(va), "v"(vb), "n"(T * 16), "n"(T * 16 + 15));
  else asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, 0" ::"v"(va), "v"(vb), "n"(T * 16), "n"(T * 16 + 15));
}
// Weight fragments come in through buffer loads: one SGPR descriptor per fetch site + the lane's 32-bit offset + a scalar offset.
// A global_load with a 64-bit address per lane costs a lone wave two carry-chained VALU operations and a longer issue
// (measured: ~35 cycles of matrix-pipe idle per fetch; with two waves per SIMD the partner covers it).
typedef unsigned int chain_u32x4b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wfrag_rsrc(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 wfrag_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// tile DST = tile SRC, as ONE matrix instruction (0 . 0 + C): a 16-register copy inside the accumulator file that costs no VALU
// issue (128 v_accvgpr_mov per unit would)
template <int ET, int DST, int SRC>
__device__ __forceinline__ void mfma_tile_copy(const uint4 zero) {
  const chain_u32x4 vz = __builtin_bit_cast(chain_u32x4, zero);
  if constexpr (ET == ET_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 a[%1:%2], %0, %0, a[%3:%4]" ::"v"(vz), "n"(DST * 16), "n"(DST * 16 + 15), "n"(SRC * 16), "n"(SRC * 16 + 15));
  else asm volatile("v_mfma_f32_32x32x16_f16 a[%1:%2], %0, %0, a[%3:%4]" ::"v"(vz), "n"(DST * 16), "n"(DST * 16 + 15), "n"(SRC * 16), "n"(SRC * 16 + 15));
}
template <int R>
__device__ __forceinline__ float agpr_read() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "n"(R));
  return v;
}
template <int R>
__device__ __forceinline__ void agpr_write(float v) { asm volatile("v_accvgpr_write_b32 a[%1], %0" ::"v"(v), "n"(R)); }
// all ones where bit R of m is set.  Opaque on purpose: hipcc turns the plain form into 64 precomputed lane masks per fragment
// (SGPR pairs, spilled to VGPR lanes and read back with two v_readlane per use)
template <int R>
__device__ __forceinline__ int bit_mask(uint32_t m) {
  int t;
  asm volatile("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(m), "n"(R));
  return t;
}
// s += d, q += d * d of the pooling arithmetic, pinned where they are written: left as plain C++, hipcc sinks the whole chain of a
// chunk (32 dependent operations, ~250 cycles) behind the chunk, into the branch that consumes the sums
__device__ __forceinline__ void moments_add(float &s, float &q, float d) {
  asm volatile("v_add_f32 %0, %0, %2\n\tv_fmac_f32 %1, %2, %2" : "+v"(s), "+v"(q) : "v"(d));
}
__device__ __forceinline__ void mfma_settle() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory"); }
__device__ __forceinline__ void agpr_claim() { asm volatile("" ::: "a0", "a255"); }        // the kernel descriptor must cover all 256

// ABL (developer aid, ASV_AMD_CHAIN_ABL with ASV_AMD_LIVE_TUNE=1; results are garbage): 1 = every weight-fragment fetch of the K
// loops reads the layer's first chunk (always in L2), 2 = no weight fetches in the K loops, 3 = no LDS fragment reads in the K
// loops, 4 = neither (the matrix stream alone); 5 = all loads, but only the plain form of the last layer's chunks in the code
// (instruction footprint), 6 = 5 without the run logic between the chunks, 7 = 5 without the stores of the pooling partials
template <int ET, int ABL = 0>
__global__ __launch_bounds__(256, 1) void tdnn_chain4_kernel(const TdnnChainParams p) {
  constexpr bool W_LOADS = ABL != 2 && ABL != 4, X_LOADS = ABL != 3 && ABL != 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[CHAIN4_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // 0..3: 128-channel slice of the 512-wide layers
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * CM;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(chain_lds_byte *)lds);
  float *par = reinterpret_cast<float *>(lds + SCR_OFF);
  const uint32_t lane16 = (uint32_t)lane * 16u;

  // developer aid: the stamps of kernels_tdnn_chain.hip ([workgroup][8][32], waves 0..3 used): 1 first window, 2 layer A's K loop,
  // 3 barrier, 4 Y stored, 5 middle K loop, 6 Y stored, 7.. one per unit of the last layer, then the drain
  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.dbg != nullptr && lane == 0 && n_stamp < 14) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + n_stamp] = __builtin_amdgcn_s_memtime();
    ++n_stamp;
  };
  stamp();
  if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 14] = __builtin_amdgcn_s_memrealtime();
  // par: [bias of layer A | middle 0 | middle 1] (all staged here, read by init_acc) | scale | shift of the layer in flight
  auto stage_bias = [&](const float *bias, int slot) {
    if (tid < 128) *reinterpret_cast<float4 *>(par + slot * CN + tid * 4) = bias != nullptr ? *reinterpret_cast<const float4 *>(bias + tid * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto stage_params = [&](const TdnnChainLayer &L) {
    if (L.scale == nullptr) return;                               // folded BatchNorm: the store needs no constants
    {
      const int which = tid >> 7, idx = (tid & 127) * 4;
      const float *src = which == 0 ? L.scale : L.shift;
      float4 v = which == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + idx);
      *reinterpret_cast<float4 *>(par + (3 + which) * CN + idx) = v;
    }
  };
  stage_bias(p.first.bias, 0);
  for (int m = 0; m < p.n_mid; ++m) stage_bias(p.mid[m].bias, 1 + m);
  float *last_bias = reinterpret_cast<float *>(lds + LAST_OFF), *last_scale = last_bias + p.last.cout_pad;
  for (int k = tid * 4; k < p.last.cout_pad; k += 1024) {
    *reinterpret_cast<float4 *>(last_bias + k) = *reinterpret_cast<const float4 *>(p.last.bias + k);
    *reinterpret_cast<float4 *>(last_scale + k) = p.last.scale != nullptr ? *reinterpret_cast<const float4 *>(p.last.scale + k) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  // the last layer's view of the tile's rows, fetched here, behind the first window's latency
  int first_seg = -1;
#pragma unroll
  for (int k = 0; k < kHalo + 1; ++k)
    if (first_seg < 0 && m0 + k < p.rows) first_seg = p.row_seg[m0 + k];
  const int rowseg_lo = p.row_seg[m0 + lane], rowseg_hi = p.row_seg[m0 + 64 + lane];

  uint4 wf[4][4];
  struct XFrags { uint4 x[4]; };
  agpr_claim();
  // 512-wide layers, D = W X^T: tile i * 4 + j, register 4 q + e = frame i * 32 + lr, channel j * 32 + 8 q + 4 lh + e of the wave's 128
  auto init_acc = [&](const float *bias128) {                    // LDS
    static_for<16>([&](auto Jq) {
      constexpr int j = decltype(Jq)::value >> 2, q = decltype(Jq)::value & 3;
      const float4 b4 = *reinterpret_cast<const float4 *>(bias128 + j * 32 + 8 * q + 4 * lh);
      static_for<4>([&](auto Ic) {
        constexpr int R = (decltype(Ic)::value * 4 + j) * 16 + q * 4;
        agpr_write<R + 0>(b4.x); agpr_write<R + 1>(b4.y); agpr_write<R + 2>(b4.z); agpr_write<R + 3>(b4.w);
      });
    });
    asm volatile("s_nop 4" ::: "memory");
  };
  // the 16 matrix instructions of one k-group share: channel fragment j against the four frame fragments
  auto mma4 = [&](const XFrags &xc, const uint4 w, auto Jc) {
    constexpr int j = decltype(Jc)::value;
    mfma_tile<ET, 0 * 4 + j>(w, xc.x[0]); mfma_tile<ET, 1 * 4 + j>(w, xc.x[1]); mfma_tile<ET, 2 * 4 + j>(w, xc.x[2]); mfma_tile<ET, 3 * 4 + j>(w, xc.x[3]);
  };

  // weight fragments of a 512-input layer: [32-channel fragment][8 chunks][4 k-groups][lane] 16 B
  constexpr size_t kFragStride512 = (size_t)(CN / CBK) * 4096;
  auto first_frags_of = [&](int layer) -> const unsigned char * {       // layer = index into mid[], n_mid = the last layer: what this wave needs first
    return layer < p.n_mid ? reinterpret_cast<const unsigned char *>(p.mid[layer].wfrag) + (size_t)(wave * 4) * kFragStride512
                           : reinterpret_cast<const unsigned char *>(p.last.wfrag) + (size_t)(wave * 2) * kFragStride512;
  };
  const unsigned char *w_after_first = first_frags_of(0);

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
      const int grp = min(wave + i * NWAVES, CGROUPS - 1);
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      a_off[i] = (uint32_t)row * (uint32_t)x_pitch + (uint32_t)cswz(w, g_slot) * 16u;
    }
    auto issue_A = [&](int c, int st) {
      const unsigned char *base = xg + (size_t)c * (CBK * 2);
#pragma unroll
      for (int i = 0; i < CPIECES; ++i) {
        const int grp = min(wave + i * NWAVES, CGROUPS - 1);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + st * CSTAGE + grp * 1024);
        chain_glds16_s(base, a_off[i], dst);
      }
    };
    const size_t frag_stride = (size_t)n_taps * nchunks * 4096;
    const unsigned char *wA = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 4) * frag_stride;   // wave-uniform
    auto x_base = [&](int c, int d) -> uint32_t {
      const int wrow = lr + kHalo + d;
      return (uint32_t)((c % CSTAGES) * CSTAGE + wrow * CROWB + ((lh ^ ((wrow >> 1) & 7)) << 4));
    };
    auto load_x4 = [&](uint32_t xb, int kg, int i, XFrags &f) {
      f.x[i] = *reinterpret_cast<const uint4 *>(lds + (xb ^ (uint32_t)(kg << 5)) + i * 4096);
    };
    issue_A(0, 0);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[kg][j] = wfrag_load(wfrag_rsrc(wA), lane16 + kg * 1024, (uint32_t)(j * frag_stride));
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
    init_acc(par + wave * 128);
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    const int d_first = __builtin_amdgcn_readlane(v_taps, 0);
    XFrags x0, x1;
    uint32_t xb = x_base(0, d_first);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_x4(xb, 0, i, x0);
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      for (int t = 0; t < n_taps; ++t) {
        const bool last_tap = (t + 1 == n_taps);
        int cn = c, tn = t + 1;
        if (last_tap) { tn = 0; cn = c + 1; }
        const bool more = cn < nchunks;
        if (!more) { cn = c; tn = t; }                       // the last step reads its own window fragments again (never used)
        // fragments of the next step; the last step fetches the first ones of the NEXT layer (same registers, same order)
        const unsigned char *wsrc = ABL == 1 ? wA : more ? wA + ((size_t)tn * nchunks + cn) * 4096 : w_after_first;
        const uint32_t wstr = (uint32_t)((more || ABL == 1) ? frag_stride : kFragStride512);
        const __amdgpu_buffer_rsrc_t wres = wfrag_rsrc(wsrc);
        // one k-group: 16 matrix instructions; the four fragments of the next k-group are read behind the first eight, every
        // weight fragment is re-fetched for the next step behind its last use
        auto group = [&](const XFrags &xc, int kg, XFrags &xn, uint32_t xbn, int kgn) {
          static_for<4>([&](auto Jc) {
            constexpr int j = decltype(Jc)::value;
            if (X_LOADS && j < 2) { load_x4(xbn, kgn, 2 * j, xn); load_x4(xbn, kgn, 2 * j + 1, xn); }
            mma4(xc, wf[kg][j], Jc);
            if (W_LOADS) wf[kg][j] = wfrag_load(wres, lane16 + kg * 1024, j * wstr);
            __builtin_amdgcn_sched_barrier(0);
          });
        };
        group(x0, 0, x1, xb, 1);
        group(x1, 1, x0, xb, 2);
        group(x0, 2, x1, xb, 3);
        if (last_tap && c + 1 < nchunks) {
          // the youngest 12 VMEM operations are this step's fragment fetches; window c + 1 (issued two chunks ago) is older
          asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
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

  // epilogue of a 512-wide layer: [ReLU], [BN unless folded into the consumer], 16-bit -> Y (row-major, 16-byte slots
  // XOR-swizzled by row & 15); the bias is in the accumulators already (kernels_tdnn_chain.hip store_Y)
  auto store_Y = [&](int relu, bool affine) {
    unsigned char *yrow = lds + lr * YROWB + lh * 8;
    const int rx = lr & 15;
    mfma_settle();
    // (the volatile register reads keep hipcc from unswitching: one loop nest per form, the branch outside)
    auto body = [&](auto Ac, auto Rc) {
      constexpr bool AFF = decltype(Ac)::value != 0, RELU = decltype(Rc)::value != 0;
      static_for<16>([&](auto Jq) {
        constexpr int j = decltype(Jq)::value >> 2, q = decltype(Jq)::value & 3;
        unsigned char *dst = yrow + (((wave * 16 + j * 4 + q) ^ rx) << 4);
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (AFF) {
          const int chl = wave * 128 + j * 32 + 8 * q + 4 * lh;
          const float4 sc4 = *reinterpret_cast<const float4 *>(par + 3 * CN + chl);
          const float4 sh4 = *reinterpret_cast<const float4 *>(par + 4 * CN + chl);
          sc[0] = sc4.x; sc[1] = sc4.y; sc[2] = sc4.z; sc[3] = sc4.w; sh[0] = sh4.x; sh[1] = sh4.y; sh[2] = sh4.z; sh[3] = sh4.w;
        }
        static_for<4>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value, R = (i * 4 + j) * 16 + q * 4;
          float y[4] = {agpr_read<R>(), agpr_read<R + 1>(), agpr_read<R + 2>(), agpr_read<R + 3>()};
          uint2 pk;
          if constexpr (!AFF) {
            pk.x = pack_h16x2<ET>(y[0], y[1]);
            pk.y = pack_h16x2<ET>(y[2], y[3]);
            if constexpr (RELU) { pk.x = relu_h16x2(pk.x); pk.y = relu_h16x2(pk.y); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = fmaf(RELU ? max_lo(y[e], 0.0f) : y[e], sc[e], sh[e]);
            pk.x = pack_h16x2<ET>(y[0], y[1]);
            pk.y = pack_h16x2<ET>(y[2], y[3]);
          }
          *reinterpret_cast<uint2 *>(dst + i * 32 * YROWB) = pk;
        });
      });
    };
    if (!affine) { if (relu) body(IC<0>{}, IC<1>{}); else body(IC<0>{}, IC<0>{}); }
    else { if (relu) body(IC<1>{}, IC<1>{}); else body(IC<1>{}, IC<0>{}); }
  };

  // fragment i of the rows of Y for k-group kg of chunk c: slot (c*8 + kg*2 + lh) ^ (lr & 15)
  const uint32_t yb = (uint32_t)(lr * YROWB);
  const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
  auto load_y4 = [&](int c, int kg, int i, XFrags &f) {
    f.x[i] = *reinterpret_cast<const uint4 *>(lds + yb + ((((uint32_t)(c * 8 + kg * 2)) ^ sx) << 4) + i * 32 * YROWB);
  };

  stamp();                                 // 2: main loop of layer A done (every window piece was waited for at its chunk; in flight: the next layer's fragments)

  // ================================ middle layers: Y -> Y ================================
  // m = -1: only the tail of layer A (its accumulators -> Y); one copy of the store code for all 512-wide layers
#pragma unroll 1
  for (int m = -1; m < p.n_mid; ++m) {
    const TdnnChainLayer &L = m < 0 ? p.first : p.mid[m];
    if (m >= 0) {
    stage_params(L);
    const size_t frag_stride = kFragStride512;
    const unsigned char *wb = first_frags_of(m);                  // its first fragments are in wf already (fetched by the previous layer's last step)
    const unsigned char *w_after = first_frags_of(m + 1);
    init_acc(par + (1 + m) * CN + wave * 128);
    XFrags x0, x1;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_y4(0, 0, i, x0);
#pragma unroll 1
    for (int c = 0; c < CN / CBK; ++c) {
      const int cn = min(c + 1, CN / CBK - 1);
      const unsigned char *wsrc = ABL == 1 ? wb : c + 1 < CN / CBK ? wb + (size_t)(c + 1) * 4096 : w_after;
      const __amdgpu_buffer_rsrc_t wres = wfrag_rsrc(wsrc);
      auto group = [&](const XFrags &xc, int kg, XFrags &xn, int c2, int kgn) {
        static_for<4>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if (X_LOADS && j < 2) { load_y4(c2, kgn, 2 * j, xn); load_y4(c2, kgn, 2 * j + 1, xn); }
          mma4(xc, wf[kg][j], Jc);
          if (W_LOADS) wf[kg][j] = wfrag_load(wres, lane16 + kg * 1024, (uint32_t)(j * frag_stride));
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      group(x0, 0, x1, c, 1);
      group(x1, 1, x0, c, 2);
      group(x0, 2, x1, c, 3);
      group(x1, 3, x0, cn, 0);
    }
    stamp();                               // 5: main loop of the middle layer done
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody reads the ring / the old Y any more (and the staged constants are visible)
    asm volatile("" ::: "memory");
    if (m < 0) stamp();                    // 3
    store_Y(L.relu, L.scale != nullptr);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();                               // 4 / 6: Y of the layer complete
  }

  // ================================ last layer + fused statistics pooling ================================
  {
    const TdnnChainLayer &L = p.last;
    const float act_lo = L.relu ? 0.0f : -INFINITY;
    const int half = m0 >> 7;
    const size_t frag_stride = kFragStride512;
    // What the 32-frame fragments of this tile are (the same for every unit):
    //   0 = gap rows only; 1 = all 32 rows inside ONE utterance (fsa); 2 = rows of utterance fsa (mask fma), then possibly rows of
    //   a second one (fsb, mask fmb), gap rows anywhere.  Three utterances inside 32 rows cannot happen: the launcher takes this
    //   kernel only when every utterance has >= 32 frames.
    int fmode[4], fsa[4], fsb[4];
    uint32_t lma[4], lmb[4];                  // per lane: which of its 16 registers of the fragment belong to the first / second utterance
    // register r of a lane holds frame 8 (r >> 2) + 4 lh + (r & 3) of its fragment: the lane's 16 bits of a 32-row mask
    auto lane_mask = [&](uint32_t m) -> uint32_t {
      const uint32_t x = m >> (4 * lh);
      return (x & 0xfu) | ((x >> 4) & 0xf0u) | ((x >> 8) & 0xf00u) | ((x >> 12) & 0xf000u);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
      const int shift = (i & 1) * 32;
      const uint32_t rem = (uint32_t)(__builtin_amdgcn_ballot_w64(rs_vec >= 0) >> shift);
      fmode[i] = 0; fsa[i] = -1; fsb[i] = -1; lma[i] = 0; lmb[i] = 0;
      if (rem != 0) {
        const int sg = __builtin_amdgcn_readlane(rs_vec, shift + __builtin_ctz(rem));
        const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(rs_vec == sg) >> shift) & rem;
        fmode[i] = bits == 0xffffffffu ? 1 : 2;
        fsa[i] = sg; lma[i] = lane_mask(bits);
        const uint32_t rest = rem & ~bits;
        if (rest != 0) {
          const int sg2 = __builtin_amdgcn_readlane(rs_vec, shift + __builtin_ctz(rest));
          fsb[i] = sg2;
          lmb[i] = lane_mask((uint32_t)(__builtin_amdgcn_ballot_w64(rs_vec == sg2) >> shift) & rest);
        }
      }
    }
    // Pooling state of the unit whose accumulators are being summed (the "previous" unit, channel base cbp):
    //   P[tile][segment slot][lh][3 = sum (u - pv), sum (u - pv)^2, pv][channel], u = act(acc + bias), the BN scale at publication.
    // The pivot pv only has to be near the data (pool_finish_kernel merges about the recorded one): it is the lane's last value of
    // the fragment in which the utterance first shows up in this tile, or the running one when that fragment is a seam.
    float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
    int cur_seg = -1;
    int cbp = 0;
    float scp[2] = {1.f, 1.f}, bp[2] = {0.f, 0.f};
    // partials of this tile: one descriptor, lane offset (lh, channel), scalar offset (slot, unit, moment)
    const __amdgpu_buffer_rsrc_t pres = wfrag_rsrc(p.pool_partial + (size_t)half * p.pool_slots * 6 * p.ld_partial);
    const uint32_t pvoff = (uint32_t)(lh * 3 * p.ld_partial + lr) * 4u;
    const uint32_t pld = (uint32_t)p.ld_partial * 4u;
    auto publish = [&]() {
      const int slot = cur_seg - first_seg;
      if constexpr (ABL == 7) { asm volatile("" ::"v"(ps[0]), "v"(ps[1]), "v"(pq[0]), "v"(pq[1]), "v"(pv[0]), "v"(pv[1])); return; }   // ablation: no stores
      if (cur_seg >= 0 && slot >= 0 && slot < p.pool_slots) {
        const uint32_t so = ((uint32_t)slot * 6u * (uint32_t)p.ld_partial + (uint32_t)cbp) * 4u;
        // (every unit lies inside the padded width: launcher)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ps[j] * scp[j]), pres, pvoff + j * 128, so, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pq[j] * scp[j] * scp[j]), pres, pvoff + j * 128, so + pld, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv[j] * scp[j]), pres, pvoff + j * 128, so + 2 * pld, 0);
        }
      }
    };

    // One unit: the K loop of 64 output channels (cb ..) into tiles i * 4 + {0, 1}, with the pooling arithmetic of the previous
    // unit (copied to tiles i * 4 + {2, 3}) in the shadow of its matrix instructions.  K chunk c carries the 16 values of fragment c >> 1, channel
    // half c & 1: one value per pair of matrix instructions, 6 plain VALU operations (12 in a fragment with a seam or gap rows).
    // On entry wf[..][0..1] and x0 hold the unit's first fragments (fetched by the previous unit's - or layer's - last chunk); its
    // last chunk fetches those of unit cb_next.
    XFrags x0, x1;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_y4(0, 0, i, x0);
    auto unit = [&](const int cb, const int cb_next, const bool has_prev) {
      constexpr int S = 0, P = 1;                 // accumulate in tiles i * 4 + {0, 1}; the previous unit's results sit in i * 4 + {2, 3}
      const unsigned char *wb = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb / 32) * frag_stride;
      const unsigned char *wb_next = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb_next / 32) * frag_stride;
      // this unit's bias / BN scale: needed when it has become the "previous" one
      const float nb0 = last_bias[cb + lr], nb1 = last_bias[cb + 32 + lr], ns0 = last_scale[cb + lr], ns1 = last_scale[cb + 32 + lr];      // LDS
      float hold_s = 0.f, hold_q = 0.f;                       // second run of a seam fragment, channel half 0, until half 1 is through
      // ablation 8 (results valid): wave 0 stamps its third unit - chunk entry, before / behind the 32 matrix instructions - into the
      // slots of the (absent) waves 4..7: [workgroup][4 * 32 + 3 c + k]
      const bool u_is_third = cb / (NWAVES * 64) == 2;
      auto fine = [&](int k) {
        if constexpr (ABL == 8 || ABL == 10)
          if (p.dbg != nullptr && lane == 0 && wave == 0 && u_is_third) p.dbg[((size_t)blockIdx.x * 8 + 4) * 32 + k] = __builtin_amdgcn_s_memtime();
      };
      static_for<8>([&](auto Cc) {
        constexpr int c = decltype(Cc)::value, fi = c >> 1, jj = c & 1;
        fine(3 * c);
        const unsigned char *wsrc = ABL == 1 ? wb : c + 1 < 8 ? wb + (size_t)(c + 1) * 4096 : wb_next;
        const __amdgpu_buffer_rsrc_t wres = wfrag_rsrc(wsrc);
        const int mode = (has_prev && ABL != 6) ? fmode[fi] : 0;           // ablation 6: the chunks alone, no run logic around them
        // (a taken branch costs a lone wave ~50 cycles of idle matrix pipe: the common case - the utterance goes on, plain chunk -
        // falls through everything here)
        if (jj == 0 && __builtin_expect(mode != 0 && fsa[fi] != cur_seg, 0)) {
          publish();
          cur_seg = fsa[fi];
          ps[0] = 0.f; ps[1] = 0.f; pq[0] = 0.f; pq[1] = 0.f;
          pv[0] = max_lo(agpr_read<(fi * 4 + 2 * P) * 16 + 15>() + bp[0], act_lo);
          pv[1] = max_lo(agpr_read<(fi * 4 + 2 * P + 1) * 16 + 15>() + bp[1], act_lo);
        }
        float s_a = 0.f, q_a = 0.f, s_b = 0.f, q_b = 0.f;
        // d = act(acc + bias) - pivot = max(acc + c1, c2)
        const float c1 = bp[jj] - pv[jj], c2 = act_lo - pv[jj];
        // The K chunk; TWO = the fragment holds a seam or gap rows: two masked runs about the same pivot.  A lone wave issues in
        // order, and a VALU instruction that needs the result of the one before it stalls the next matrix instruction with it:
        // the arithmetic of value n is spread over the slots of values n .. n + 2, so that no slot holds two dependent operations
        //   plain: A(n) read n, max n-1          B(n) add n, sum / sum of squares n-1
        //   TWO:   A(n) read n, masks n, max n-1, first run's sums n-2     B(n) add n, and n-1, second run's sums n-2
        auto chunk = [&](auto Tc) {
          constexpr bool TWO = decltype(Tc)::value != 0;
          const uint32_t lm_a = lma[fi], lm_b = lmb[fi];
          float xv[16], yv[16], dv[16], da_[16], db_[16];
          int ta[16], tb[16];
          static_for<16>([&](auto Nc) {
            constexpr int n = decltype(Nc)::value, kg = n >> 2, q = n & 3, j = q >> 1, i0 = (q & 1) * 2;
            constexpr int kgn = (kg + 1) & 3, c2x = kg == 3 ? (c + 1) & 7 : c;      // the last k-group of the unit reads chunk 0 again: the next unit's
            const XFrags &xc = (kg & 1) ? x1 : x0;
            XFrags &xn = (kg & 1) ? x0 : x1;
            // slot A
            if (X_LOADS && q < 2) { load_y4(c2x, kgn, 2 * q, xn); load_y4(c2x, kgn, 2 * q + 1, xn); }
            if (c == 0 && kg == 0) mfma_tile_zero<ET, i0 * 4 + 2 * S + j>(xc.x[i0], wf[kg][j]);
            else mfma_tile<ET, i0 * 4 + 2 * S + j>(xc.x[i0], wf[kg][j]);
            xv[n] = agpr_read<(fi * 4 + 2 * P + jj) * 16 + n>();
            if constexpr (TWO) { ta[n] = bit_mask<n>(lm_a); tb[n] = bit_mask<n>(lm_b); }
            if constexpr (n >= 1) dv[n - 1] = max_lo(yv[n - 1], c2);
            if constexpr (TWO && n >= 2) moments_add(s_a, q_a, da_[n - 2]);
            __builtin_amdgcn_sched_barrier(0);
            // slot B
            if (c == 0 && kg == 0) mfma_tile_zero<ET, (i0 + 1) * 4 + 2 * S + j>(xc.x[i0 + 1], wf[kg][j]);
            else mfma_tile<ET, (i0 + 1) * 4 + 2 * S + j>(xc.x[i0 + 1], wf[kg][j]);
            if (W_LOADS && q == 1) wf[kg][0] = wfrag_load(wres, lane16 + kg * 1024, 0u);
            if (W_LOADS && q == 3) wf[kg][1] = wfrag_load(wres, lane16 + kg * 1024, (uint32_t)frag_stride);
            yv[n] = xv[n] + c1;
            if constexpr (!TWO && n >= 1) moments_add(s_a, q_a, dv[n - 1]);
            if constexpr (TWO && n >= 1) {
              da_[n - 1] = __int_as_float(__float_as_int(dv[n - 1]) & ta[n - 1]);
              db_[n - 1] = __int_as_float(__float_as_int(dv[n - 1]) & tb[n - 1]);
            }
            if constexpr (TWO && n >= 2) moments_add(s_b, q_b, db_[n - 2]);
            __builtin_amdgcn_sched_barrier(0);
          });
          // the pipeline's tail (no matrix instruction to hide behind: ~3 operations, 9 in a TWO chunk)
          dv[15] = max_lo(yv[15], c2);
          if constexpr (!TWO) moments_add(s_a, q_a, dv[15]);
          else {
            da_[15] = __int_as_float(__float_as_int(dv[15]) & ta[15]);
            db_[15] = __int_as_float(__float_as_int(dv[15]) & tb[15]);
            moments_add(s_a, q_a, da_[14]); moments_add(s_b, q_b, db_[14]);
            moments_add(s_a, q_a, da_[15]); moments_add(s_b, q_b, db_[15]);
          }
        };
        fine(3 * c + 1);
        if constexpr ((ABL >= 5 && ABL <= 7) || ABL == 10) chunk(IC<0>{});   // ablation: no second form of the chunk in the code
        else if constexpr (ABL == 9) chunk(IC<1>{});         // ablation: only the two-run form
        else { if (__builtin_expect(mode == 2, 0)) chunk(IC<1>{}); else chunk(IC<0>{}); }
        fine(3 * c + 2);
        {
          const float keep = mode != 0 ? 1.0f : 0.0f;           // wave-uniform
          ps[jj] = fmaf(keep, s_a, ps[jj]); pq[jj] = fmaf(keep, q_a, pq[jj]);
          if (__builtin_expect(mode == 2 && fsb[fi] >= 0, 0)) {
            if (jj == 0) { hold_s = s_b; hold_q = q_b; }
            else {                               // the first utterance of the fragment is complete; the second goes on with the same pivot
              publish();
              cur_seg = fsb[fi];
              ps[0] = hold_s; pq[0] = hold_q; ps[1] = s_b; pq[1] = q_b;
            }
          }
        }
      });
      if (has_prev) { publish(); cur_seg = -1; }
      fine(24);
      // this unit's accumulators become the "previous" ones: eight tile copies through the matrix pipe (+3 % of its work; the code of
      // all units is the same - with two alternating sets and twice the code the chunks ran 20 % slower: instruction fetch)
      {
        const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        static_for<8>([&](auto Tc) {
          constexpr int i = decltype(Tc)::value >> 1, j = decltype(Tc)::value & 1;
          mfma_tile_copy<ET, i * 4 + 2 + j, i * 4 + j>(zero);
        });
      }
      mfma_settle();                              // the next unit (or the drain) reads them with VALU instructions
      // this unit is the next one's "previous"
      cbp = cb;
      bp[0] = nb0; bp[1] = nb1; scp[0] = ns0; scp[1] = ns1;
    };

    const int units = L.cout_pad / (64 * NWAVES);              // per wave
#pragma unroll 1
    for (int u = 0; u < units; ++u) {
      const int cb = (u * NWAVES + wave) * 64;
      unit(cb, u + 1 < units ? cb + NWAVES * 64 : cb, u > 0);
      stamp();                             // 7 .. 12
    }
    // drain: the wave's last unit; no matrix work left to hide behind.  One body, the fragment's 32 values copied
    // to fixed registers first.
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
      const int mode = f == 0 ? fmode[0] : f == 1 ? fmode[1] : f == 2 ? fmode[2] : fmode[3];
      if (mode == 0) continue;
      const int sa = f == 0 ? fsa[0] : f == 1 ? fsa[1] : f == 2 ? fsa[2] : fsa[3];
      const int sb = f == 0 ? fsb[0] : f == 1 ? fsb[1] : f == 2 ? fsb[2] : fsb[3];
      const uint32_t lm_a = f == 0 ? lma[0] : f == 1 ? lma[1] : f == 2 ? lma[2] : lma[3];
      const uint32_t lm_b = f == 0 ? lmb[0] : f == 1 ? lmb[1] : f == 2 ? lmb[2] : lmb[3];
      float u[2][16];
      static_for<4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        if (f == i) {
          static_for<32>([&](auto Rc) {
            constexpr int j = decltype(Rc)::value >> 4, r = decltype(Rc)::value & 15;
            u[j][r] = max_lo(agpr_read<(i * 4 + 2 + j) * 16 + r>() + bp[j], act_lo);
          });
        }
      });
      if (sa != cur_seg) {
        publish();
        cur_seg = sa;
        ps[0] = 0.f; ps[1] = 0.f; pq[0] = 0.f; pq[1] = 0.f;
        pv[0] = u[0][15]; pv[1] = u[1][15];
      }
      float sb_[2] = {0.f, 0.f}, qb_[2] = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float s_a = 0.f, q_a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = u[j][r] - pv[j];
          const float da = __int_as_float(__float_as_int(d) & __builtin_amdgcn_sbfe(lm_a, r, 1));
          const float db = __int_as_float(__float_as_int(d) & __builtin_amdgcn_sbfe(lm_b, r, 1));
          s_a += da; q_a = fmaf(da, da, q_a);
          sb_[j] += db; qb_[j] = fmaf(db, db, qb_[j]);
        }
        ps[j] += s_a; pq[j] += q_a;
      }
      if (sb >= 0) {
        publish();
        cur_seg = sb;
        ps[0] = sb_[0]; pq[0] = qb_[0]; ps[1] = sb_[1]; pq[1] = qb_[1];
      }
    }
    publish();
    stamp();                               // 13
    if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

// every wave gets whole 64-channel units, the last layer's constants fit the LDS left, and no 32-frame fragment can hold rows of three utterances
bool tdnn_chain4_supported(const TdnnChainParams &p) {
  return p.last.cout_pad % (64 * NWAVES) == 0 && p.last.cout_pad >= 64 * NWAVES && p.last.cout_pad <= kMaxLastWidth && p.ld_partial >= p.last.cout_pad && p.min_seg_len >= 32 &&
         (unsigned long long)p.pool_slots * 6ull * (unsigned long long)p.ld_partial * 4ull < (1ull << 31);
}

int launch_tdnn_chain4(const TdnnChainParams &p, hipStream_t s) {
  ASV_REQUIRE(tdnn_chain4_supported(p), "tdnn(chain4): %d output channels / shortest utterance %d frames not supported", p.last.cout_pad, p.min_seg_len);
  const dim3 grid(p.rows / CM), block(64 * NWAVES);
  static const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;
  // the ablations whose results the header quotes stay built (4, 5, 6, 9); 1 - 3, 7, 8, 10 exist in the source for re-runs
  const int abl = live && getenv("ASV_AMD_CHAIN4_ABL") != nullptr ? atoi(getenv("ASV_AMD_CHAIN4_ABL")) : 0;
  if (p.et == ET_F16) hipLaunchKernelGGL((tdnn_chain4_kernel<ET_F16>), grid, block, 0, s, p);
  else if (abl == 4) hipLaunchKernelGGL((tdnn_chain4_kernel<ET_BF16, 4>), grid, block, 0, s, p);
  else if (abl == 5) hipLaunchKernelGGL((tdnn_chain4_kernel<ET_BF16, 5>), grid, block, 0, s, p);
  else if (abl == 6) hipLaunchKernelGGL((tdnn_chain4_kernel<ET_BF16, 6>), grid, block, 0, s, p);
  else if (abl == 9) hipLaunchKernelGGL((tdnn_chain4_kernel<ET_BF16, 9>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((tdnn_chain4_kernel<ET_BF16>), grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
