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
// 8 half instructions (two k-groups) + 4 scaled ones; the main operands of a k-group are single-buffered and re-fetched two phases
// ahead, the 8-bit operands double-buffered a whole pair ahead.  One workgroup (512 threads, 136 KiB of LDS) per CU.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int MM = 64;                     // frames per workgroup
constexpr int MN = kChainWidth;            // channels of the resident tile (512)
constexpr int MROW = 128;                  // window row: 32 f32; image row: [hi16 64 B | lo8 32 B | hi8 32 B]
constexpr int MWINR = MM + 2 * kHalo;      // 72 window rows
constexpr int MSTG = MWINR * MROW;         // 9216 B per f32 stage / per image
constexpr int MYROW = MN * 2;              // 1024 B per row of Yh and of Y8
constexpr int MYIMG = MM * MYROW;          // 65536 B: Yh at 0, Y8 at MYIMG
constexpr int MPAR = 2 * MYIMG;            // bias | scale | shift of the layer in flight (6 KiB)
constexpr int CHAINM_LDS = 2 * MYIMG + 8192;
static_assert(4 * MSTG <= MYIMG, "layer A's stages and images live inside the Y region");
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

  auto stage_params = [&](const TdnnChainLayer &L) {
    if (tid < 384) {
      const int which = tid >> 7, idx = (tid & 127) * 4;
      float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float *src = (which == 0) ? L.bias : (which == 1 ? L.scale : L.shift);
      if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + idx);
      *reinterpret_cast<float4 *>(par + which * MN + idx) = v;
    }
  };

  struct MH { uint4 w[2], x[2]; };          // main operands of one 16-channel k-group: this wave's two weight fragments, the two frame fragments
  struct ME { uint4 w[2][2], x[2][2]; };    // 8-bit operands of a 32-channel pair: w[j][K block], x[i][K block]
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
  auto mma_main = [&](const MH &h, int q, auto tr) {
    const int i = q & 1, j = q >> 1;
    if constexpr (decltype(tr)::value) acc[i][j] = mfma16<ET_F16>(h.x[i], h.w[j], acc[i][j]);
    else acc[i][j] = mfma16<ET_F16>(h.w[j], h.x[i], acc[i][j]);
  };
  auto mma_mx = [&](const ME &e, int q, auto tr) {
    const int i = q & 1, j = q >> 1;
    const mx_v8i a = {(int)e.w[j][0].x, (int)e.w[j][0].y, (int)e.w[j][0].z, (int)e.w[j][0].w, (int)e.w[j][1].x, (int)e.w[j][1].y, (int)e.w[j][1].z, (int)e.w[j][1].w};
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
    // window of chunk c -> f32 stage c & 1: piece w by wave w, the ninth piece by wave 0 (rows beyond the matrix ends are clamped
    // onto zero gap rows)
    auto piece_off = [&](int grp) -> size_t {
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      return (size_t)row * x_pitch + (size_t)mswz(w, g_slot) * 16u;
    };
    const size_t off_a = piece_off(wave), off_b = piece_off(8);
    auto issue_A = [&](int c) {
      const unsigned char *base = xg + (size_t)c * MROW;
      chainm_glds16(base + off_a, __builtin_amdgcn_readfirstlane(lds_base + (c & 1) * MSTG + wave * 1024));
      if (wave == 0) chainm_glds16(base + off_b, __builtin_amdgcn_readfirstlane(lds_base + (c & 1) * MSTG + 8 * 1024));
    };
    // f32 stage c & 1 -> image c & 1 (at 2 * MSTG): row = [hi halves of 32 channels (slots 0-3) | x_lo8 (slots 4, 5) | x_hi8 (slots 6, 7)];
    // thread (w, q) converts channels 8 q .. 8 q + 7 of row w
    auto convert = [&](int c) {
      if (tid < MWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        const unsigned char *src = lds + (c & 1) * MSTG + w * MROW;
        const uint4 a = *reinterpret_cast<const uint4 *>(src + mswz(w, 2 * q) * 16);
        const uint4 b = *reinterpret_cast<const uint4 *>(src + mswz(w, 2 * q + 1) * 16);
        uint4 hi;
        int h8a = 0, l8a = 0, h8b = 0, l8b = 0;
        split_mx<false>(__uint_as_float(a.x), __uint_as_float(a.y), hi.x, h8a, l8a, range);
        split_mx<true>(__uint_as_float(a.z), __uint_as_float(a.w), hi.y, h8a, l8a, range);
        split_mx<false>(__uint_as_float(b.x), __uint_as_float(b.y), hi.z, h8b, l8b, range);
        split_mx<true>(__uint_as_float(b.z), __uint_as_float(b.w), hi.w, h8b, l8b, range);
        unsigned char *dst = lds + (2 + (c & 1)) * MSTG + w * MROW;
        *reinterpret_cast<uint4 *>(dst + mswz(w, q) * 16) = hi;
        *reinterpret_cast<uint2 *>(dst + mswz(w, 4 + (q >> 1)) * 16 + (q & 1) * 8) = make_uint2((uint32_t)l8a, (uint32_t)l8b);
        *reinterpret_cast<uint2 *>(dst + mswz(w, 6 + (q >> 1)) * 16 + (q & 1) * 8) = make_uint2((uint32_t)h8a, (uint32_t)h8b);
      }
    };
    const size_t frag_stride = (size_t)n_taps * nkg * 1024;                    // half fragments: bytes per 32-channel output fragment
    const size_t frag8_stride = (size_t)n_taps * nchunks * 2048;               // 8-bit fragments: [tap][32-channel group][K block][lane][16]
    const unsigned char *wh = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 2) * frag_stride + lane16;
    const unsigned char *w8 = reinterpret_cast<const unsigned char *>(p.first.w8) + (size_t)(wave * 2) * frag8_stride + lane16;
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    // LDS byte address of this lane's row of image c & 1 for tap t, and its swizzle term (blind to + 32 rows)
    auto x_row = [&](int c, int t, uint32_t &base, int &sw) {
      const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t);
      sw = (wrow >> 1) & 7;
      base = (uint32_t)((2 + (c & 1)) * MSTG + wrow * MROW);
    };
    auto ld_main = [&](int c, int t, int kg, MH &h) {
      const size_t off = ((size_t)t * nkg + (size_t)c * 2 + kg) * 1024;
      h.w[0] = *reinterpret_cast<const uint4 *>(wh + off); h.w[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + off);
      uint32_t base; int sw;
      x_row(c, t, base, sw);
      const uint32_t a = base + (uint32_t)(((kg * 2 + lh) ^ sw) << 4);
      h.x[0] = *reinterpret_cast<const uint4 *>(lds + a); h.x[1] = *reinterpret_cast<const uint4 *>(lds + a + 32 * MROW);
    };
    auto ld_mx = [&](int c, int t, ME &e) {
      const size_t off = ((size_t)t * nchunks + c) * 2048;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        e.w[j][0] = *reinterpret_cast<const uint4 *>(w8 + j * frag8_stride + off);
        e.w[j][1] = *reinterpret_cast<const uint4 *>(w8 + j * frag8_stride + off + 1024);
      }
      uint32_t base; int sw;
      x_row(c, t, base, sw);
      const uint32_t a0 = base + (uint32_t)(((4 + lh) ^ sw) << 4), a1 = base + (uint32_t)(((6 + lh) ^ sw) << 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        e.x[i][0] = *reinterpret_cast<const uint4 *>(lds + a0 + i * 32 * MROW);
        e.x[i][1] = *reinterpret_cast<const uint4 *>(lds + a1 + i * 32 * MROW);
      }
    };
    issue_A(0);
    if (nchunks > 1) issue_A(1);
    // window 0 landed (everything but this wave's pieces of window 1) -> image 0; its stage then takes window 2
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
    init_acc(p.first.bias + wave * 64, p.first.w_scale, MTrNo{});
    MH h0, h1;
    ME e0, e1;
    ld_mx(0, 0, e0);
    ld_main(0, 0, 0, h0);
    ld_main(0, 0, 1, h1);
    const int P = nchunks * n_taps;
    int c = 0, t = 0;
    // pair n = (chunk c, tap t): 4 + 4 half instructions and 4 scaled ones; the fetches of pair n + 1 are pinned between them
    auto step = [&](const ME &ec, ME &en, int n) {
      int c2 = c, t2 = t + 1;
      if (t2 == n_taps) { t2 = 0; c2 = c + 1; }
      const bool more = n + 1 < P;
      if (!more) { c2 = c; t2 = t; }                               // the last pair re-fetches itself (valid memory, never used)
      const bool enter = more && c2 != c;
      if (enter) {
        // entering chunk c + 1 (the protocol of kernels_tdnn_chainx.hip): its image is complete, window c + 2 has landed - older than
        // the youngest 4 vector-memory operations, the half fragments of pair n fetched in the previous step's phases 2 and 3 -;
        // window c + 2 becomes image c & 1 now, its stage takes window c + 3, issued BEHIND this step's first weight fetches
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 2 < nchunks) convert(c + 2);
      }
      const size_t off8 = ((size_t)t2 * nchunks + c2) * 2048;
      const size_t offh = ((size_t)t2 * nkg + (size_t)c2 * 2) * 1024;
      uint32_t base; int sw;
      x_row(c2, t2, base, sw);
      const uint32_t ax0 = base + (uint32_t)(((4 + lh) ^ sw) << 4), ax1 = base + (uint32_t)(((6 + lh) ^ sw) << 4);
      const uint32_t ah0 = base + (uint32_t)(((lh) ^ sw) << 4), ah1 = base + (uint32_t)(((2 + lh) ^ sw) << 4);
      // phase 1: k-group 0 of the pair; the 8-bit operands of the next pair
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == 0) { en.w[0][0] = *reinterpret_cast<const uint4 *>(w8 + off8); en.w[0][1] = *reinterpret_cast<const uint4 *>(w8 + off8 + 1024); }
        if (q == 1) { en.w[1][0] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off8); en.w[1][1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off8 + 1024); }
        if (q == 1 && enter && c + 3 < nchunks) issue_A(c + 3);
        if (q == 2) { en.x[0][0] = *reinterpret_cast<const uint4 *>(lds + ax0); en.x[0][1] = *reinterpret_cast<const uint4 *>(lds + ax1); }
        if (q == 3) { en.x[1][0] = *reinterpret_cast<const uint4 *>(lds + ax0 + 32 * MROW); en.x[1][1] = *reinterpret_cast<const uint4 *>(lds + ax1 + 32 * MROW); }
        mma_main(h0, q, MTrNo{});
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 2: k-group 1; k-group 0 of the next pair into the registers phase 1 has just read
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_main(h1, q, MTrNo{});
        if (q == 0) { h0.w[0] = *reinterpret_cast<const uint4 *>(wh + offh); h0.w[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh); }
        if (q == 1) { h0.x[0] = *reinterpret_cast<const uint4 *>(lds + ah0); h0.x[1] = *reinterpret_cast<const uint4 *>(lds + ah0 + 32 * MROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 3: the corrections; k-group 1 of the next pair
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_mx(ec, q, MTrNo{});
        if (q == 0) { h1.w[0] = *reinterpret_cast<const uint4 *>(wh + offh + 1024); h1.w[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh + 1024); }
        if (q == 1) { h1.x[0] = *reinterpret_cast<const uint4 *>(lds + ah1); h1.x[1] = *reinterpret_cast<const uint4 *>(lds + ah1 + 32 * MROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
      c = c2; t = t2;
    };
    for (int n = 0; n < P; n += 2) {
      step(e0, e1, n);
      if (n + 1 < P) step(e1, e0, n + 1);
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
        // Y8: the 32-channel group (wave * 2 + j) owns slots 4 g .. 4 g + 3 = [lo8 ch 0-15 | lo8 ch 16-31 | hi8 ch 0-15 | hi8 ch 16-31];
        // channels 8 q + 4 lh + e: sixteen-group q >> 1, bytes 8 (q & 1) + 4 lh + e
        const int grp = (wave * 2 + j) * 4 + (q >> 1);
        const uint32_t lo_off = (uint32_t)(((grp ^ rx) << 4) + (q & 1) * 8 + lh * 4), hi_off = (uint32_t)((((grp + 2) ^ rx) << 4) + (q & 1) * 8 + lh * 4);
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
  auto yloop = [&](const unsigned char *wbh, const unsigned char *wb8, const float *bias64, float w_scale, auto tr) {
    constexpr size_t fs = (size_t)(MN / 16) * 1024, fs8 = (size_t)(MN / 32) * 2048;
    const uint32_t yb = (uint32_t)(lr * MYROW);
    const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
    MH h0, h1;
    ME e0, e1;
    auto ld_main = [&](int kg, MH &h) {
      const size_t off = (size_t)kg * 1024 + lane16;
      h.w[0] = *reinterpret_cast<const uint4 *>(wbh + off); h.w[1] = *reinterpret_cast<const uint4 *>(wbh + fs + off);
      const uint32_t a = yb + ((((uint32_t)(kg * 2)) ^ sx) << 4);              // slot (2 kg + lh) ^ (lr & 15)
      h.x[0] = *reinterpret_cast<const uint4 *>(lds + a); h.x[1] = *reinterpret_cast<const uint4 *>(lds + a + 32 * MYROW);
    };
    auto ld_mx = [&](int n, ME &e) {
      const size_t off = (size_t)n * 2048 + lane16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        e.w[j][0] = *reinterpret_cast<const uint4 *>(wb8 + j * fs8 + off);
        e.w[j][1] = *reinterpret_cast<const uint4 *>(wb8 + j * fs8 + off + 1024);
      }
      const uint32_t a0 = MYIMG + yb + ((((uint32_t)(n * 4)) ^ sx) << 4), a1 = MYIMG + yb + ((((uint32_t)(n * 4 + 2)) ^ sx) << 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        e.x[i][0] = *reinterpret_cast<const uint4 *>(lds + a0 + i * 32 * MYROW);
        e.x[i][1] = *reinterpret_cast<const uint4 *>(lds + a1 + i * 32 * MYROW);
      }
    };
    ld_mx(0, e0);
    ld_main(0, h0);
    ld_main(1, h1);
    init_acc(bias64, w_scale, tr);
    auto step = [&](const ME &ec, ME &en, int nn) {                // computes the pair in (h0, h1, ec); fetches pair nn
      const size_t off8 = (size_t)nn * 2048 + lane16;
      const size_t offh = (size_t)(nn * 2) * 1024 + lane16;
      const uint32_t ax0 = MYIMG + yb + ((((uint32_t)(nn * 4)) ^ sx) << 4), ax1 = MYIMG + yb + ((((uint32_t)(nn * 4 + 2)) ^ sx) << 4);
      const uint32_t ah0 = yb + ((((uint32_t)(nn * 4)) ^ sx) << 4), ah1 = yb + ((((uint32_t)(nn * 4 + 2)) ^ sx) << 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == 0) { en.w[0][0] = *reinterpret_cast<const uint4 *>(wb8 + off8); en.w[0][1] = *reinterpret_cast<const uint4 *>(wb8 + off8 + 1024); }
        if (q == 1) { en.w[1][0] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + off8); en.w[1][1] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + off8 + 1024); }
        if (q == 2) { en.x[0][0] = *reinterpret_cast<const uint4 *>(lds + ax0); en.x[0][1] = *reinterpret_cast<const uint4 *>(lds + ax1); }
        if (q == 3) { en.x[1][0] = *reinterpret_cast<const uint4 *>(lds + ax0 + 32 * MYROW); en.x[1][1] = *reinterpret_cast<const uint4 *>(lds + ax1 + 32 * MYROW); }
        mma_main(h0, q, tr);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_main(h1, q, tr);
        if (q == 0) { h0.w[0] = *reinterpret_cast<const uint4 *>(wbh + offh); h0.w[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh); }
        if (q == 1) { h0.x[0] = *reinterpret_cast<const uint4 *>(lds + ah0); h0.x[1] = *reinterpret_cast<const uint4 *>(lds + ah0 + 32 * MYROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mma_mx(ec, q, tr);
        if (q == 0) { h1.w[0] = *reinterpret_cast<const uint4 *>(wbh + offh + 1024); h1.w[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh + 1024); }
        if (q == 1) { h1.x[0] = *reinterpret_cast<const uint4 *>(lds + ah1); h1.x[1] = *reinterpret_cast<const uint4 *>(lds + ah1 + 32 * MYROW); }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int n = 0; n < MN / 32; n += 2) {
      step(e0, e1, n + 1);
      step(e1, e0, min(n + 2, MN / 32 - 1));                       // the last pair re-fetches itself (never used)
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
    const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(wave * 2) * ((size_t)(MN / 16) * 1024);
    const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(wave * 2) * ((size_t)(MN / 32) * 2048);
    yloop(wbh, wb8, L.bias + wave * 64, L.w_scale, MTrNo{});
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
#pragma unroll 1
    for (int cb = wave * 64; cb < L.cout_pad; cb += 512) {
      const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cb / 32) * ((size_t)(MN / 16) * 1024);
      const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(cb / 32) * ((size_t)(MN / 32) * 2048);
      yloop(wbh, wb8, L.bias + cb, L.w_scale, MTrYes{});
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
    }
  }
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
  hipLaunchKernelGGL(tdnn_chainm_kernel, grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
