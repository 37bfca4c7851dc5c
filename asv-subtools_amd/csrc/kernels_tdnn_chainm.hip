// The layer chain of kernels_tdnn_chainx.hip - tdnn -> [1-tap 512 -> 512]* -> 1-tap + fused statistics pooling in ONE kernel
// (model/xvector.py:77-98: tdnn3 -> tdnn4 -> tdnn5 -> StatisticsPooling; components.py:107-149, 410-431; pooling.py:58-67) - for
// the "f32m" form of the f32x precision mode (round 6): the two CORRECTION products of the operand split run on gfx950's
// block-scaled 8-bit matrix instruction instead of the 16-bit one.
//
//     w x  ~  w_hi x_hi                       v_mfma_f32_32x32x16_f16: IEEE-half halves, the products are exact in the f32 accumulator
//           + w_hi x_lo + w_lo x_hi           ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 channels: its two 32-deep K blocks carry
//                                             [e4m3(w_hi) . e5m2(x_lo)] and [e4m3(w_lo) . e5m2(x_hi)], the block scales (powers of
//                                             two, E8M0) undo the scaling of the 8-bit operands
//
// The corrections are 2^-11 of a product; rounding THEM to 3 - 4 significant bits leaves an error of ~2^-15 per product - measured
// (CPU emulation of the whole x-vector against the f32 forward, three weight seeds; tools/emulate_f32m.py): embeddings 0.9 - 1.0e-5
// of the reference's against 0.9e-7 with three half products and 2.5 - 3.4e-4 for the f16 mode; the north star's gate is 1e-4.  The
// matrix pipe runs the scaled 8-bit instruction at twice the 16-bit rate (K = 64 in the time of two K = 16 instructions): per 32
// channels and accumulator 2 + 1 instructions = 4 time units instead of 6 (tools/mx_probe.hip on the device: 72 ms against 115 ms
// for the same products on every CU, random operands).  e5m2 for the activation side because it has the exponent range of the half
// itself: no range limit beyond the mode's own (|x| < 57344 here: the watch below).
//
// Operand layout of the scaled instruction (pinned on the device by tools/mx_layout.hip): lane l holds row / column l & 31; its
// registers 0-3 belong to K block 0, registers 4-7 to K block 1, A and B positions pair (half, register, byte) with the same; the scale
// byte of a lane < 32 applies to block 0 of its row / column, of a lane >= 32 to block 1.  Here: byte q of block b of lane half lh =
// channel 16 lh + q of the 32-channel group; block 0 = (w_hi8, x_lo8), block 1 = (w_lo8, x_hi8).
//
// Everything else is kernels_tdnn_chainx.hip: 64 frames per workgroup, 8 waves, wave w = channels 64 w .. 64 w + 63 of the resident
// 512-channel tile (2 x 2 accumulators), the tile resident in LDS between the layers - as its half image Yh (64 KiB) and ONE 8-bit
// image Y8 (64 KiB: per row and 32-channel group [x_lo8 16 | x_lo8 16 | x_hi8 16 | x_hi8 16]) where chainx keeps two half images -,
// last layer in 64-channel units with swapped operands and the register-only pooling epilogue.  The K loop works in 32-channel pairs:
// 8 half instructions (two k-groups) + 4 scaled ones; ALL weight fragments of a pair are fetched a whole step ahead into the second of
// two register sets, the rows from LDS two phases ahead: halves (2 B), e4m3(w_lo) (1 B) and e4m3(w_hi) (1 B, the second plane of
// pack_tdnn_weight_mx8; CHAINM_WHI8_REG = 1 makes it in registers instead: 24 conversions per step, 2.5 % slower).  One workgroup (512
// threads, 136 KiB of LDS) per CU.  Layer A's rows come as f32 (converted in place, a ring of 3 windows, a barrier per chunk) or as IMAGES from
// the producing layer's epilogue (kernels_tdnn_x3m.hip: a ring of 10 windows, a barrier per four chunks) - see there.
//
// Where the time goes (round 6, s_memtime stamps: profiles/r6i, r6o, r6p, r6r, r6s, r6u, r6x): per 64-frame tile 114.7 k cycles of matrix work per
// SIMD, 206 k measured at first, 192 k now.  A K step (one pair, 1024 matrix cycles for the two waves of a SIMD) takes ~1350 cycles in the
// barrier-free Y loops and ~1450 in layer A with image rows (1900 with f32 rows and a barrier per chunk).  Not the cause: the window
// conversion, the window DMA, the order of the fragment array (a copy in step order changed nothing), the fetch distance of the fragments
// (350 -> 1250 cycles: nothing), the stream from L2 as such (every step fetching the SAME fragments: -5 %), the LDS reads (0 %).  The
// cause, by experiment builds (-DCHAINM_EXP, r6x) and tools/vmem_probe.hip (r6z): the NUMBER of fragment loads a CU issues per step - 8 x 1
// KiB per wave, whatever level serves them (a bare probe with this kernel's mix - 8 loads + 16 LDS reads per 16 matrix instructions - runs at
// 60 - 63 % of the bf16 datasheet rate on trivial operands; spreading the loads over the step, 71 % against 55 % in the probe without the LDS
// reads, measured SLOWER in the kernel: tools/chainm_spread_loads.patch, profiles/r6z_chainm_spread_loads_ab.txt); with half of those loads the loop's cycles drop 9 % and the shader clock RISES 8 % (power),
// together -13 % per workgroup.  4 bytes per weight is what this form costs; more frames per fetched byte is the lever and LDS capacity
// (2 KiB per resident frame) stops it - the 96-frame kernel (x_hi8 made in registers to fit) paid more in conversions than it gained.
// Alternating the issue priority between the two waves of a SIMD step by step gives 3 % (r6p).
#include <cstdlib>

#include "device_utils.h"

#ifndef CHAINM_EXP
#define CHAINM_EXP 0
#endif
#ifndef CHAINM_WHI8_REG
#define CHAINM_WHI8_REG 0      // 1: e4m3(w_hi) made in registers from the half fragments (24 conversions per step) instead of fetched (rounds 6i - 6r)
#endif

namespace asv {
namespace {

constexpr int MM = 64;                     // frames per workgroup
constexpr int MN = kChainWidth;            // channels of the resident tile (512)
constexpr int MROW = 128;                  // window row: 32 f32; image row: [hi16 64 B | lo8 32 B | hi8 32 B]
constexpr int MWINR = MM + 2 * kHalo;      // 72 window rows
constexpr int MSTG = MWINR * MROW;         // 9216 B per window buffer (f32 stage, then image, in place)
constexpr int MYROW = MN * 2;              // 1024 B per row of Yh and of Y8
constexpr int MYIMG = MM * MYROW;          // 65536 B: Yh at 0, Y8 at MYIMG
constexpr int MPAR = 2 * MYIMG;            // bias | scale | shift of the layer in flight (6 KiB)
constexpr int CHAINM_LDS = 2 * MYIMG + 8192;
constexpr int MRING = 10;                  // image rows in (p.x_image): window buffers of the ring, see layer A
static_assert(MRING * MSTG <= 2 * MYIMG, "layer A's window buffers live inside the Y region (Yh + Y8)");
static_assert(CHAINM_LDS <= 163840, "160 KiB of LDS per CU");

// E8M0 block scales (2^(byte - 127)) that undo the host's / the epilogue's scaling of the 8-bit operands:
//   w_hi8 = e4m3(w_hi 2^-6), w_lo8 = e4m3(w_lo 2^6)   (pack_tdnn_weight_mx8: w_hi < 2^14, |w_lo| <= 2^-11 |w_hi|)
//   x_lo8 = e5m2(x_lo 2^11), x_hi8 = e5m2(x)
constexpr int kScaleWhi = 127 + 6, kScaleWlo = 127 - 6, kScaleXlo = 127 - 11, kScaleXhi = 127;

typedef __attribute__((address_space(3))) unsigned char chainm_lds_byte;
typedef int mx_v8i __attribute__((ext_vector_type(8)));
struct MTrNo { static constexpr bool value = false; };
struct MTrYes { static constexpr bool value = true; };

__device__ __forceinline__ int mswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void chainm_glds16(const void *gsrc, uint32_t lds_dst) {
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

// Range watch of this form: a packed pair of hi halves -> bits 15 / 31 set iff |half| >= 57344 (0x7b00 + 0x0500 carries into bit 15),
// inf and NaN included: beyond it e5m2(x) has no finite value.  Published as ASV_STATUS_HALF_RANGE like the half split's own watch;
// the callers re-run such a batch on the bf16-halves twin.
__device__ __forceinline__ uint32_t mx_range_bits(uint32_t packed_hi) { return (packed_hi & 0x7fff7fffu) + 0x05000500u; }

// two f32 -> the packed pair of hi halves, and the two 8-bit pairs (low 16 bits of hi8 / lo8 when SEL = false, high 16 bits otherwise)
template <bool SEL>
__device__ __forceinline__ void split_mx(float v0, float v1, uint32_t &hi16, int &hi8, int &lo8, uint32_t &range) {
  hi16 = pack_h16x2<ET_F16>(v0, v1);
  range |= mx_range_bits(hi16);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi16), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi16), "v"(v1));
  hi8 = __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, hi8, SEL);
  lo8 = __builtin_amdgcn_cvt_pk_bf8_f32(r0 * 2048.0f, r1 * 2048.0f, lo8, SEL);
}

// IMG: the first layer's input rows are images (p.x_image).  DEV: the instantiation with the developer aids compiled in (per-step stamps,
// ASV_AMD_CHAINM_ABL, the x_image == 2 protocol); the production instantiations carry none of their tests in the K loops (layer A's loop
// was 269 scalar + 143 vector instructions per 24 matrix instructions with all of them in, round 6).
template <bool IMG, bool DEV>
__global__ __launch_bounds__(512, 2) void tdnn_chainm_kernel(const TdnnChainParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CHAINM_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * MM;
  uint32_t range = 0u;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(chainm_lds_byte *)lds);
  float *par = reinterpret_cast<float *>(lds + MPAR);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const int scale_w = lh ? kScaleWlo : kScaleWhi, scale_x = lh ? kScaleXhi : kScaleXlo;

  // developer aid (ASV_AMD_CHAIN_DBG=1): [workgroup][wave][32] s_memtime stamps at the phase boundaries; 14 / 15: s_memrealtime at start / end
  const int abl = DEV ? p.abl : 0;
  const bool fine = DEV && p.dbg != nullptr && p.dbg_fine;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.dbg != nullptr && lane == 0 && n_stamp < 14) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + n_stamp] = __builtin_amdgcn_s_memtime();
    ++n_stamp;
  };
  stamp();                                                       // 0: start
  if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 14] = __builtin_amdgcn_s_memrealtime();
  auto stage_params = [&](const TdnnChainLayer &L) {
    if (tid < 384) {
      const int which = tid >> 7, idx = (tid & 127) * 4;
      float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = (which == 0) ? L.bias : (which == 1 ? L.scale : L.shift);
      if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + idx);
      *reinterpret_cast<float4 *>(par + which * MN + idx) = v;
    }
  };

  // Operands of a 32-channel pair.  Weights (from L2): this wave's two 32-channel fragments, as halves for the two k-groups and as K block 1
  // (w_lo8) of the scaled instruction - TWO sets, fetched a whole step ahead (round 6: with the half fragments single-buffered and re-fetched
  // ~350 matrix cycles ahead the K loops ran at 0.6 - 0.7 of the matrix rate, profiles/r6i_chainm_phase_stamps.txt).  Rows (from LDS): the
  // two frame fragments per k-group, single-buffered two phases ahead, and their 8-bit K blocks x8[i][block], two sets.
#if CHAINM_WHI8_REG
  struct MW { uint4 h0[2], h1[2], e[2]; };
#else
  struct MW { uint4 h0[2], h1[2], e[2], q[2]; };      // q: K block 0 of the pair's 8-bit weights, e4m3(w_hi 2^-6), from the array's second plane
#endif
  struct MX8 { uint4 x[2][2]; };
  uint4 h0x[2], h1x[2];
#if CHAINM_WHI8_REG
  uint4 wq[2];                              // K block 0 of the pair's weights (w_hi8), made in registers from the two half fragments (whi8 below)
#endif
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
  // instruction q (0..3) of a phase: accumulator (i, j) = (q & 1, q >> 1); an accumulator recurs every 4th instruction
  auto mma_main = [&](const uint4 (&w)[2], const uint4 (&x)[2], int q, auto tr) {
    const int i = q & 1, j = q >> 1;
    if constexpr (decltype(tr)::value) acc[i][j] = mfma16<ET_F16>(x[i], w[j], acc[i][j]);
    else acc[i][j] = mfma16<ET_F16>(w[j], x[i], acc[i][j]);
  };
#if CHAINM_WHI8_REG
  // the 8 half values of a weight fragment -> e4m3(w_hi 2^-6) in two registers (RNE, the same bytes the host would pack): bytes 0-7 of K block
  // 0 from the pair's first k-group, bytes 8-15 from its second
  auto whi8 = [&](const uint4 &f, uint32_t &d0, uint32_t &d1) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    s16x2 a = {0, 0}, b = {0, 0};
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, __builtin_bit_cast(h16x2, f.x), 64.0f, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, __builtin_bit_cast(h16x2, f.y), 64.0f, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, __builtin_bit_cast(h16x2, f.z), 64.0f, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, __builtin_bit_cast(h16x2, f.w), 64.0f, true);
    d0 = __builtin_bit_cast(uint32_t, a); d1 = __builtin_bit_cast(uint32_t, b);
  };
#endif
  auto mma_mx = [&](const MW &w, const MX8 &e, int q, auto tr) {
    const int i = q & 1, j = q >> 1;
#if CHAINM_WHI8_REG
    const mx_v8i a = {(int)wq[j].x, (int)wq[j].y, (int)wq[j].z, (int)wq[j].w, (int)w.e[j].x, (int)w.e[j].y, (int)w.e[j].z, (int)w.e[j].w};
#else
    const mx_v8i a = {(int)w.q[j].x, (int)w.q[j].y, (int)w.q[j].z, (int)w.q[j].w, (int)w.e[j].x, (int)w.e[j].y, (int)w.e[j].z, (int)w.e[j].w};
#endif
    const mx_v8i b = {(int)e.x[i][0].x, (int)e.x[i][0].y, (int)e.x[i][0].z, (int)e.x[i][0].w, (int)e.x[i][1].x, (int)e.x[i][1].y, (int)e.x[i][1].z, (int)e.x[i][1].w};
    // operand formats: 0 = e4m3 (weights), 1 = e5m2 (activations)
    if constexpr (decltype(tr)::value) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc[i][j], 1, 0, 0, scale_x, 0, scale_w);
    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 1, 0, scale_w, 0, scale_x);
  };

  // ================================ phase 1: layer A, f32 window -> image -> products ================================
  stage_params(p.first);
  {
    const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
    const size_t x_pitch = (size_t)p.ldx * 4;
    const int g_row = lane >> 3, g_slot = lane & 7;
    const int nchunks = p.cin_pad / 32;
    const int n_taps = p.n_taps;
    const int nkg = (p.cin_pad / 64) * 4;                       // 16-channel k-groups per tap in the half fragment arrays
    // Window of chunk c -> buffer c % 3 (three buffers of 72 rows x 128 B inside the not yet written Y region): piece w by wave w, the ninth
    // piece by wave 0 (rows beyond the matrix ends are clamped onto zero gap rows).  A buffer is f32 stage AND image: the conversion runs
    // IN PLACE - the four threads of a row sit in one wave, their reads of the whole row retire before any of them writes - and, from
    // chunk 2 on, interleaved with the matrix instructions of the step that follows the chunk barrier instead of in front of them (round
    // 6: with the conversion between barrier and K loop every wave of the CU left the matrix pipe idle for it once per chunk - layer A's
    // loop took 79 k cycles for 49 k of matrix work, profiles/r6i_chainm_phase_stamps.txt).
    auto piece_off = [&](int grp) -> size_t {
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      return (size_t)row * x_pitch + (size_t)mswz(w, g_slot) * 16u;
    };
    const size_t off_a = piece_off(wave), off_b = piece_off(8);
    auto issue_A = [&](int c, int buf) {
      const unsigned char *base = xg + (size_t)c * MROW;
      chainm_glds16(base + off_a, __builtin_amdgcn_readfirstlane(lds_base + buf * MSTG + wave * 1024));
      if (wave == 0) chainm_glds16(base + off_b, __builtin_amdgcn_readfirstlane(lds_base + buf * MSTG + 8 * 1024));
    };
    // f32 rows -> image rows [hi halves of 32 channels (slots 0-3) | x_lo8 (slots 4, 5 = lane halves 0, 1) | x_hi8 (slots 6, 7)], in place; thread (w, q)
    // converts channels 8 q .. 8 q + 7 of row w: cv_load reads its 32 bytes, cv_store writes its 16 + 8 + 8
    uint4 cva = make_uint4(0, 0, 0, 0), cvb = make_uint4(0, 0, 0, 0);
    auto cv_load = [&](int buf) {
      if (tid < MWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        const unsigned char *src = lds + buf * MSTG + w * MROW;
        cva = *reinterpret_cast<const uint4 *>(src + mswz(w, 2 * q) * 16);
        cvb = *reinterpret_cast<const uint4 *>(src + mswz(w, 2 * q + 1) * 16);
      }
    };
    auto cv_store = [&](int buf) {
      if (tid < MWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        uint4 hi;
        int h8a = 0, l8a = 0, h8b = 0, l8b = 0;
        split_mx<false>(__uint_as_float(cva.x), __uint_as_float(cva.y), hi.x, h8a, l8a, range);
        split_mx<true>(__uint_as_float(cva.z), __uint_as_float(cva.w), hi.y, h8a, l8a, range);
        split_mx<false>(__uint_as_float(cvb.x), __uint_as_float(cvb.y), hi.z, h8b, l8b, range);
        split_mx<true>(__uint_as_float(cvb.z), __uint_as_float(cvb.w), hi.w, h8b, l8b, range);
        unsigned char *dst = lds + buf * MSTG + w * MROW;
        *reinterpret_cast<uint4 *>(dst + mswz(w, q) * 16) = hi;
        // byte b of an 8-bit slot of lane half lh = channel (b < 8 ? 8 lh + b : 16 + 8 lh + b - 8): the order of the lane's two half fragments
        *reinterpret_cast<uint2 *>(dst + mswz(w, 4 + (q & 1)) * 16 + (q >> 1) * 8) = make_uint2((uint32_t)l8a, (uint32_t)l8b);
        *reinterpret_cast<uint2 *>(dst + mswz(w, 6 + (q & 1)) * 16 + (q >> 1) * 8) = make_uint2((uint32_t)h8a, (uint32_t)h8b);
      }
    };
    // fragment offsets of pair (c, t) in the tap-major arrays of pack_tdnn_weight_frags / _mx8.  (A copy in this loop's own order - consecutive
    // steps fetching consecutive kilobytes instead of addresses 32 KiB apart - was measured without effect: profiles/r6o_*.)
    auto off_h = [&](int c, int t) -> size_t { return ((size_t)t * nkg + (size_t)c * 2) * 1024; };
    auto off_8 = [&](int c, int t) -> size_t { return ((size_t)t * nchunks + c) * 1024; };
    const size_t frag_stride = (size_t)n_taps * nkg * 1024;                    // half fragments: bytes per 32-channel output fragment
    const size_t frag8_stride = (size_t)n_taps * nchunks * 1024;               // 8-bit fragments (w_lo8): [tap][32-channel group][lane][16]
    const unsigned char *wh = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 2) * frag_stride + lane16;
    const unsigned char *w8 = reinterpret_cast<const unsigned char *>(p.first.w8) + (size_t)(wave * 2) * frag8_stride + lane16;
    const size_t plane8 = (size_t)(kChainWidth / 32) * frag8_stride;           // the w_hi8 plane follows the w_lo8 plane (pack_tdnn_weight_mx8)
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    // LDS byte address of this lane's row of the image in buffer `buf` for tap t, and its swizzle term (blind to + 32 rows)
    auto x_row = [&](int buf, int t, uint32_t &base, int &sw) {
      const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t);
      sw = (wrow >> 1) & 7;
      base = (uint32_t)(buf * MSTG + wrow * MROW);
    };
    // the weights of pair (c, t): 6 fragment loads
    auto ld_w = [&](int c, int t, MW &w) {
      const size_t offh = off_h(c, t), off8 = off_8(c, t);
      w.h0[0] = *reinterpret_cast<const uint4 *>(wh + offh); w.h0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh);
      w.h1[0] = *reinterpret_cast<const uint4 *>(wh + offh + 1024); w.h1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh + 1024);
      w.e[0] = *reinterpret_cast<const uint4 *>(w8 + off8); w.e[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off8);
#if !CHAINM_WHI8_REG
      w.q[0] = *reinterpret_cast<const uint4 *>(w8 + plane8 + off8); w.q[1] = *reinterpret_cast<const uint4 *>(w8 + plane8 + frag8_stride + off8);
#endif
    };
    // f32 rows: windows 0, 1, 2 in flight; 0 and 1 converted at once (the K loop's first barrier follows the first chunk); a ring of 3 buffers,
    // a workgroup barrier per chunk (the conversion of window c + 2 and the DMA of window c + 3 hang on it).
    // Image rows (p.x_image: the producing layer's epilogue made them, kernels_tdnn_x3m.hip): nothing to convert - a ring of MRING buffers
    // filled SIX chunks ahead, and a barrier only in front of chunks 1, 4, 8, 12, ...: at the barrier in front of chunk g every wave has
    // waited for its own pieces of the windows up to g + 3 (the youngest of them issued three steps before), so the group's windows are
    // complete; the window issued behind it, c + 6, takes the buffer of chunk c - 4, which lies in front of the group barrier every wave
    // has passed.  4 barriers in layer A instead of 16.
    const bool grouped = IMG && (!DEV || p.x_image == 1);          // (x_image == 2, developer build of the kernel: image rows under the per-chunk protocol of the f32 rows)
    const int ring = grouped ? MRING : 3;
    MW w0, w1;
    MX8 e0, e1;
    issue_A(0, 0);
    if (grouped) {
      ld_w(0, 0, w0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CHAINM_WHI8_REG ? 6 : 8) : "memory");      // window 0 is older than the fragment loads
      for (int k = 1; k < 6; ++k) if (k < nchunks) issue_A(k, k);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      if (nchunks > 1) issue_A(1, 1);
      if (nchunks > 2) issue_A(2, 2);
      ld_w(0, 0, w0);
      // windows 0 and 1 have landed: everything but the youngest 6 (the fragments) + this wave's pieces of window 2 ... kept simple: all of it
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (!IMG) {
        cv_load(0); cv_store(0);
        if (nchunks > 1) { cv_load(1); cv_store(1); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    init_acc(p.first.bias + wave * 64, p.first.w_scale, MTrNo{});
    {
      uint32_t base; int sw;
      x_row(0, 0, base, sw);
      const uint32_t a0 = base + (uint32_t)((lh ^ sw) << 4), a1 = base + (uint32_t)(((2 + lh) ^ sw) << 4);
      const uint32_t a4 = base + (uint32_t)(((4 + lh) ^ sw) << 4), a6 = base + (uint32_t)(((6 + lh) ^ sw) << 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        h0x[i] = *reinterpret_cast<const uint4 *>(lds + a0 + i * 32 * MROW);
        h1x[i] = *reinterpret_cast<const uint4 *>(lds + a1 + i * 32 * MROW);
        e0.x[i][0] = *reinterpret_cast<const uint4 *>(lds + a4 + i * 32 * MROW);
        e0.x[i][1] = *reinterpret_cast<const uint4 *>(lds + a6 + i * 32 * MROW);
      }
    }
    const int P = nchunks * n_taps;
    int c = 0, t = 0, cb = 0;                                     // cb = c % ring: the buffer of chunk c's image
    stamp();                                                     // 1: layer A's prologue (three windows, two conversions, first fetches)
    // pair n = (chunk c, tap t): 4 + 4 half instructions and 4 scaled ones; the fetches of pair n + 1 are pinned between them
    auto step = [&](const MW &wc, MW &wn, const MX8 &ec, MX8 &en, int n) {
      if (fine && lane == 0 && n < 16) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 16 + n] = __builtin_amdgcn_s_memtime();
      // The two waves of a SIMD (w and w + 4) take turns at the higher issue priority, step by step: left alone the older wave of a pair runs
      // its steps in ~1280 cycles and the younger in ~1900 (profiles/r6o_chainm_layerA_step_stamps.txt, no barrier), and a chunk barrier
      // then runs at the pace of the slower one.  ASV_AMD_CHAINM_ABL bit 3: off.
      if ((abl & 8) == 0) { if (((n ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
      int c2 = c, t2 = t + 1;
      if (t2 == n_taps) { t2 = 0; c2 = c + 1; }
      const bool more = n + 1 < P;
      if (!more) { c2 = c; t2 = t; }                               // the last pair re-fetches itself (valid memory, never used)
      const bool enter = more && c2 != c;
      const int cb1 = cb + 1 == ring ? 0 : cb + 1, cb2 = cb == 0 ? 2 : cb - 1;        // (c + 1) % ring; f32 rows: (c + 2) % 3
      const bool cv = !IMG && enter && c + 2 < nchunks && (abl & 1) == 0;
      const int ahead = grouped ? 6 : 3;                          // the window issued behind this step's fetches: c + ahead
      if (enter && (abl & 4) == 0 && (!grouped || c == 0 || ((c + 1) & 3) == 0)) {
        // Entering chunk c + 1, at the start of the LAST step of chunk c (this step's operands were fetched in the previous one): image
        // c + 1 is complete (converted during the step behind the previous chunk barrier: lgkmcnt), window c + 2 has landed (issued a
        // chunk ago; the only vector-memory operations behind it that may still be in flight are this step's own weight fragments,
        // needed now anyway), nobody reads image c any more.  Behind the barrier: window c + 3 goes into image c's buffer (issued
        // behind this step's weight fetches), window c + 2 is converted in place between this step's matrix instructions.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      const size_t off8 = off_8(c2, t2);
      const size_t offh = off_h(c2, t2);
      uint32_t base; int sw;
      x_row(c2 != c ? cb1 : cb, t2, base, sw);
      const uint32_t ax0 = base + (uint32_t)(((4 + lh) ^ sw) << 4), ax1 = base + (uint32_t)(((6 + lh) ^ sw) << 4);
      const uint32_t ah0 = base + (uint32_t)(((lh) ^ sw) << 4), ah1 = base + (uint32_t)(((2 + lh) ^ sw) << 4);
      // phase 1: k-group 0 of the pair; ALL weight fragments of the next pair, its 8-bit rows
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == 0) { wn.h0[0] = *reinterpret_cast<const uint4 *>(wh + offh); wn.h0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh); }
        if (q == 0) { wn.h1[0] = *reinterpret_cast<const uint4 *>(wh + offh + 1024); wn.h1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh + 1024); }
        if (q == 1) { wn.e[0] = *reinterpret_cast<const uint4 *>(w8 + off8); wn.e[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off8); }
#if !CHAINM_WHI8_REG
        if (q == 1) { wn.q[0] = *reinterpret_cast<const uint4 *>(w8 + plane8 + off8); wn.q[1] = *reinterpret_cast<const uint4 *>(w8 + plane8 + frag8_stride + off8); }
#endif
        if (q == 1 && enter && c + ahead < nchunks && (abl & 2) == 0) issue_A(c + ahead, grouped ? (cb + 6 >= MRING ? cb + 6 - MRING : cb + 6) : cb);
        if constexpr (!IMG) { if (q == 1 && cv) cv_load(cb2); }
        if (q == 2) { en.x[0][0] = *reinterpret_cast<const uint4 *>(lds + ax0); en.x[0][1] = *reinterpret_cast<const uint4 *>(lds + ax1); }
#if CHAINM_WHI8_REG
        if (q == 2) { whi8(wc.h0[0], wq[0].x, wq[0].y); whi8(wc.h0[1], wq[1].x, wq[1].y); }
#endif
        if (q == 3) { en.x[1][0] = *reinterpret_cast<const uint4 *>(lds + ax0 + 32 * MROW); en.x[1][1] = *reinterpret_cast<const uint4 *>(lds + ax1 + 32 * MROW); }
        mma_main(wc.h0, h0x, q, MTrNo{});
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 2: k-group 1; the rows of the next pair's k-group 0 into the registers phase 1 has just read
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_main(wc.h1, h1x, q, MTrNo{});
        if (q == 0) { h0x[0] = *reinterpret_cast<const uint4 *>(lds + ah0); h0x[1] = *reinterpret_cast<const uint4 *>(lds + ah0 + 32 * MROW); }
#if CHAINM_WHI8_REG
        if (q == 1) { whi8(wc.h1[0], wq[0].z, wq[0].w); whi8(wc.h1[1], wq[1].z, wq[1].w); }
#endif
        if constexpr (!IMG) { if (q == 2 && cv) cv_store(cb2); }
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 3: the corrections; the rows of the next pair's k-group 1
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_mx(wc, ec, q, MTrNo{});
        if (q == 0) { h1x[0] = *reinterpret_cast<const uint4 *>(lds + ah1); h1x[1] = *reinterpret_cast<const uint4 *>(lds + ah1 + 32 * MROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (c2 != c) cb = cb1;
      c = c2; t = t2;
    };
    for (int n = 0; n < P; n += 2) {
      step(w0, w1, e0, e1, n);
      if (n + 1 < P) step(w1, w0, e1, e0, n + 1);
    }
  }

  // epilogue of a 512-wide layer: acc / w_scale -> [ReLU] -> [folded BN unless it sits in the next layer's weights] -> hi halves into Yh,
  // the two 8-bit values into Y8 (16-byte slots XOR-swizzled by row & 15)
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
          const float4 sc4 = *reinterpret_cast<const float4 *>(par + MN + chl);
          const float4 sh4 = *reinterpret_cast<const float4 *>(par + 2 * MN + chl);
          sc[0] = sc4.x; sc[1] = sc4.y; sc[2] = sc4.z; sc[3] = sc4.w;
          sh[0] = sh4.x; sh[1] = sh4.y; sh[2] = sh4.z; sh[3] = sh4.w;
        }
        // Yh: 4 consecutive channels = 8 bytes inside the 16-byte slot (wave * 8 + j * 4 + q), half lh
        const uint32_t slot_off = (uint32_t)((((wave * 8 + j * 4 + q) ^ rx) << 4) + lh * 8);
        // Y8: the 32-channel group (wave * 2 + j) owns slots 4 g .. 4 g + 3 = [lo8 of lane half 0 | lo8 of lane half 1 | hi8 of 0 | hi8 of 1], a
        // half's 16 bytes = channels 8 lhK .. + 7 then 16 + 8 lhK .. + 7 (the order of the two half fragments); channels 8 q + 4 lh + e of
        // this lane: lhK = q & 1, bytes 8 (q >> 1) + 4 lh + e
        const int grp = (wave * 2 + j) * 4 + (q & 1);
        const uint32_t lo_off = (uint32_t)(((grp ^ rx) << 4) + (q >> 1) * 8 + lh * 4), hi_off = (uint32_t)((((grp + 2) ^ rx) << 4) + (q >> 1) * 8 + lh * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = max_lo(acc[i][j][q * 4 + e] * unscale, act_lo);
            y[e] = affine ? fmaf(v, sc[e], sh[e]) : v;
          }
          uint2 hi;
          int h8 = 0, l8 = 0;
          split_mx<false>(y[0], y[1], hi.x, h8, l8, range);
          split_mx<true>(y[2], y[3], hi.y, h8, l8, range);
          unsigned char *row = lds + (i * 32 + lr) * MYROW;
          *reinterpret_cast<uint2 *>(row + slot_off) = hi;
          *reinterpret_cast<uint32_t *>(row + MYIMG + lo_off) = (uint32_t)l8;
          *reinterpret_cast<uint32_t *>(row + MYIMG + hi_off) = (uint32_t)h8;
        }
      }
  };

  // main loop of a layer whose input is Y: K = 512 = 16 pairs of 32 channels, no barrier.  wbh / wb8: wave-uniform bases of the half / 8-bit
  // fragment arrays of this wave's (or unit's) first 32-channel output fragment; the second follows at + 32 KiB in both.
  auto yloop = [&](const unsigned char *wbh, const unsigned char *wb8, size_t plane8, const float *bias64, float w_scale, auto tr) {
    constexpr size_t fs = (size_t)(MN / 16) * 1024, fs8 = (size_t)(MN / 32) * 1024;
    const uint32_t yb = (uint32_t)(lr * MYROW);
    const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
    MW w0, w1;
    MX8 e0, e1;
    auto ld_w = [&](int n, MW &w) {
      const size_t offh = (size_t)(n * 2) * 1024 + lane16, off8 = (size_t)n * 1024 + lane16;
      w.h0[0] = *reinterpret_cast<const uint4 *>(wbh + offh); w.h0[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh);
      w.h1[0] = *reinterpret_cast<const uint4 *>(wbh + offh + 1024); w.h1[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh + 1024);
      w.e[0] = *reinterpret_cast<const uint4 *>(wb8 + off8); w.e[1] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + off8);
#if !CHAINM_WHI8_REG
      w.q[0] = *reinterpret_cast<const uint4 *>(wb8 + plane8 + off8); w.q[1] = *reinterpret_cast<const uint4 *>(wb8 + plane8 + fs8 + off8);
#endif
    };
    ld_w(0, w0);
    {
      const uint32_t a0 = yb + ((0u ^ sx) << 4), a1 = yb + ((2u ^ sx) << 4);                  // slots (2 kg + lh) ^ (lr & 15), kg = 0, 1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        h0x[i] = *reinterpret_cast<const uint4 *>(lds + a0 + i * 32 * MYROW);
        h1x[i] = *reinterpret_cast<const uint4 *>(lds + a1 + i * 32 * MYROW);
        e0.x[i][0] = *reinterpret_cast<const uint4 *>(lds + MYIMG + a0 + i * 32 * MYROW);
        e0.x[i][1] = *reinterpret_cast<const uint4 *>(lds + MYIMG + a1 + i * 32 * MYROW);
      }
    }
    init_acc(bias64, w_scale, tr);
    auto step = [&](const MW &wc, MW &wn, const MX8 &ec, MX8 &en, int nn) {      // computes the pair in (wc, h0x, h1x, ec); fetches pair nn
      if ((abl & 8) == 0) { if (((nn ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
      // (developer instantiation, results garbage: abl bit 4 = every step fetches the SAME weight fragments - the loads stay, the stream from
      //  L2 goes; bit 5 = every step reads the same rows of Y)
      // (CHAINM_EXP, experiment builds only - never the product: bit 0 = no fetch of the two 8-bit weight blocks in the Y loops, bit 1 = none of k-group 1's halves)
      const int nw = (abl & 16) ? 0 : nn, nx = (abl & 32) ? 0 : nn;
      const size_t off8 = (size_t)nw * 1024 + lane16;
      const size_t offh = (size_t)(nw * 2) * 1024 + lane16;
      const uint32_t ax0 = MYIMG + yb + ((((uint32_t)(nx * 4)) ^ sx) << 4), ax1 = MYIMG + yb + ((((uint32_t)(nx * 4 + 2)) ^ sx) << 4);
      const uint32_t ah0 = yb + ((((uint32_t)(nx * 4)) ^ sx) << 4), ah1 = yb + ((((uint32_t)(nx * 4 + 2)) ^ sx) << 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == 0) { wn.h0[0] = *reinterpret_cast<const uint4 *>(wbh + offh); wn.h0[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh); }
        if (q == 0 && !(CHAINM_EXP & 2)) { wn.h1[0] = *reinterpret_cast<const uint4 *>(wbh + offh + 1024); wn.h1[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh + 1024); }
        if (q == 1 && !(CHAINM_EXP & 1)) { wn.e[0] = *reinterpret_cast<const uint4 *>(wb8 + off8); wn.e[1] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + off8); }
#if !CHAINM_WHI8_REG
        if (q == 1 && !(CHAINM_EXP & 1)) { wn.q[0] = *reinterpret_cast<const uint4 *>(wb8 + plane8 + off8); wn.q[1] = *reinterpret_cast<const uint4 *>(wb8 + plane8 + fs8 + off8); }
#endif
        if (q == 2) { en.x[0][0] = *reinterpret_cast<const uint4 *>(lds + ax0); en.x[0][1] = *reinterpret_cast<const uint4 *>(lds + ax1); }
#if CHAINM_WHI8_REG
        if (q == 2) { whi8(wc.h0[0], wq[0].x, wq[0].y); whi8(wc.h0[1], wq[1].x, wq[1].y); }
#endif
        if (q == 3) { en.x[1][0] = *reinterpret_cast<const uint4 *>(lds + ax0 + 32 * MYROW); en.x[1][1] = *reinterpret_cast<const uint4 *>(lds + ax1 + 32 * MYROW); }
        mma_main(wc.h0, h0x, q, tr);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_main(wc.h1, h1x, q, tr);
        if (q == 0) { h0x[0] = *reinterpret_cast<const uint4 *>(lds + ah0); h0x[1] = *reinterpret_cast<const uint4 *>(lds + ah0 + 32 * MYROW); }
#if CHAINM_WHI8_REG
        if (q == 1) { whi8(wc.h1[0], wq[0].z, wq[0].w); whi8(wc.h1[1], wq[1].z, wq[1].w); }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_mx(wc, ec, q, tr);
        if (q == 0) { h1x[0] = *reinterpret_cast<const uint4 *>(lds + ah1); h1x[1] = *reinterpret_cast<const uint4 *>(lds + ah1 + 32 * MYROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int n = 0; n < MN / 32; n += 2) {
      step(w0, w1, e0, e1, n + 1);
      step(w1, w0, e1, e0, min(n + 2, MN / 32 - 1));               // the last pair re-fetches itself (never used)
    }
  };

  __builtin_amdgcn_s_setprio(0);
  stamp();                                                       // 2: layer A's K loop
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done with layer A's stages and images: Y may be written
  asm volatile("" ::: "memory");
  stamp();                                                       // 3: the wait for the slowest wave
  store_Y(p.first.relu, p.first.scale != nullptr, 1.0f / p.first.w_scale);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // Y complete; the constants of layer A are dead
  asm volatile("" ::: "memory");
  stamp();                                                       // 4: layer A's epilogue (split, Yh / Y8) + barrier

  // ================================ middle layers: Y -> Y ================================
#pragma unroll 1
  for (int m = 0; m < p.n_mid; ++m) {
    const TdnnChainLayer &L = p.mid[m];
    stage_params(L);
    const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(wave * 2) * ((size_t)(MN / 16) * 1024);
    const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(wave * 2) * ((size_t)(MN / 32) * 1024);
    yloop(wbh, wb8, (size_t)(L.cout_pad / 32) * ((size_t)(MN / 32) * 1024), L.bias + wave * 64, L.w_scale, MTrNo{});
    stamp();                                                     // 5: a middle layer's K loop
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody reads the old Y any more (and the staged constants are visible)
    asm volatile("" ::: "memory");
    store_Y(L.relu, L.scale != nullptr, 1.0f / L.w_scale);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();                                                     // 6: its barriers + epilogue
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
#pragma unroll 1
    for (int cb = wave * 64; cb < L.cout_pad; cb += 512) {
      const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb / 32) * ((size_t)(MN / 16) * 1024);
      const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(cb / 32) * ((size_t)(MN / 32) * 1024);
      yloop(wbh, wb8, (size_t)(L.cout_pad / 32) * ((size_t)(MN / 32) * 1024), L.bias + cb, L.w_scale, MTrYes{});
      stamp();                                                   // 7, 9, 11: a unit's K loop
      // Pooling epilogue, registers only (kernels_tdnn_chainx.hip, the same arithmetic): acc[i][j][r] = channel cb + j*32 + lr, frame
      // i*32 + 8 (r >> 2) + 4 lh + (r & 3); a lane sums its own frames per utterance about the pivot of its FIRST frame of that utterance,
      // the two lane halves publish P[tile of 64 rows][segment slot][lh][3 = sum (u - pv), sum (u - pv)^2, pv][channel] with the BN scale
      // applied at publication; pool_finish_kernel merges the parts and adds the BN shift.
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
          const bool need = !have && lm != 0;                    // (the stale-pivot rule of round 5: tests/test_gpu_xvector.py::test_pooled_moments_ignore_the_neighbour)
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
      stamp();                                                   // 8, 10, 12: its pooling epilogue
    }
  }
  if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 15] = __builtin_amdgcn_s_memrealtime();
  x3_publish_range(range, p.status);
}

}  // namespace

int launch_tdnn_chainm(const TdnnChainParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows % MM == 0 && p.rows >= MM, "tdnn(chainm): rows %d not a multiple of %d", p.rows, MM);
  ASV_REQUIRE(p.cin_pad % 64 == 0 && p.cin_pad >= 64 && p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS, "tdnn(chainm): first layer with %d channels / %d taps", p.cin_pad, p.n_taps);
  ASV_REQUIRE(p.first.wfrag && p.first.w8 && p.last.wfrag && p.last.w8 && p.last.bias && p.n_mid >= 0 && p.n_mid <= 2 && p.last.cout_pad % 64 == 0,
              "tdnn(chainm): incomplete layer description");
  for (int m = 0; m < p.n_mid; ++m) ASV_REQUIRE(p.mid[m].wfrag && p.mid[m].w8, "tdnn(chainm): middle layer %d without 8-bit weights", m);
  ASV_REQUIRE(p.first.w_scale > 0.0f && p.last.w_scale > 0.0f && p.et == ET_F16, "tdnn(chainm): the half split with scaled weights only");
  ASV_REQUIRE(p.pool_partial && p.row_seg && p.pool_slots >= 1, "tdnn(chainm): the last layer feeds the fused pooling (partials / row map missing)");
  for (int t = 0; t < p.n_taps; ++t) ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(chainm): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const dim3 grid(p.rows / MM), block(512);
  const bool dev = p.dbg != nullptr || p.abl != 0 || p.x_image == 2;
  if (p.x_image) {
    if (dev) hipLaunchKernelGGL((tdnn_chainm_kernel<true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((tdnn_chainm_kernel<true, false>), grid, block, 0, s, p);
  } else {
    if (dev) hipLaunchKernelGGL((tdnn_chainm_kernel<false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((tdnn_chainm_kernel<false, false>), grid, block, 0, s, p);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
