// The wide frame layers of the f32x precision mode in its "f32m" form (round 6; ASV_FLAG_X3_MX8): kernels_tdnn_x3.hip's 128-row SHARED
// kernel - F.conv1d + ReLU + eval BN of components.py:107-149, 410-431 with f32 activations in HBM - with the two correction products
// of the operand split on gfx950's block-scaled 8-bit matrix instruction:
//
//     w x  ~  w_hi x_hi  (v_mfma_f32_32x32x16_f16, exact in the f32 accumulator)
//           + [e4m3(w_hi) . e5m2(x_lo) | e4m3(w_lo) . e5m2(x_hi)]  (ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 channels and accumulator)
//
// Numerics, operand layout and scales: kernels_tdnn_chainm.hip (the same split, the same 8-bit fragments: pack_tdnn_weight_mx8).
// Structure: 128 frames x 256 channels per workgroup, 4 waves (128 x 64 = 4 x 2 accumulators), two workgroups per CU; the f32 window
// of a 32-channel chunk comes by LDS-DMA into one of two stages and is converted ONCE per workgroup into an image row
// [hi halves 64 B | x_lo8 32 B | x_hi8 32 B]; per (chunk, tap) a wave runs 8 + 8 half instructions and 8 scaled ones, every operand set
// (the accumulators take 128 of the 256 registers; the weight fragments of a phase are re-fetched from L2 right behind it, the LDS operands
// rotate through two sets of four registers one phase ahead):
//     phase 1  k-group 0 (wh0, xa)                 xb <- k-group 1
//     phase 2  k-group 1 (wh1, xb)                 wh0 <- next pair; xa <- 8-bit K blocks of frame fragments 0, 1
//     phase 3a corrections of fragments 0, 1       wh1 <- next pair; xb <- K blocks of fragments 2, 3     [a new chunk is entered behind it]
//     phase 3b corrections of fragments 2, 3       xa <- k-group 0 of the next pair; we <- next pair
// Plain and generic epilogues of kernels_tdnn_x3.hip; the fused-pooling and 64-row forms stay on the three-product kernel.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int QBN = 256;            // channels per workgroup
constexpr int QBK = 32;             // channels per chunk
constexpr int QROWB = 128;
constexpr int QSPITCH = 68;         // floats per epilogue scratch row
constexpr int QBM = 128;            // frames per workgroup
constexpr int QWIN = QBM + 2 * kHalo;          // 136 window rows
constexpr int QGROUPS = QWIN / 8;              // 17 eight-row DMA pieces
constexpr int QPIECES = (QGROUPS + 3) / 4;     // 5 per wave
constexpr int QSTAGE = QWIN * QROWB;           // 17408 B
constexpr int QRING = 4 * QSTAGE;              // two f32 stages + two images
static_assert(4 * 32 * QSPITCH * 4 <= QRING, "epilogue scratch must fit in the ring");
static_assert(QBN == kBigTileN, "weight padding must match the N tile");

constexpr int kQScaleWhi = 127 + 6, kQScaleWlo = 127 - 6, kQScaleXlo = 127 - 11, kQScaleXhi = 127;     // kernels_tdnn_chainm.hip

typedef __attribute__((address_space(3))) unsigned char x3m_lds_byte;
typedef int q_v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int qswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void x3m_glds16(const void *gsrc, uint32_t lds_dst) {
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

__device__ __forceinline__ uint32_t q_range_bits(uint32_t packed_hi) { return (packed_hi & 0x7fff7fffu) + 0x05000500u; }      // |half| >= 57344

template <bool SEL>
__device__ __forceinline__ void q_split(float v0, float v1, uint32_t &hi16, int &hi8, int &lo8, uint32_t &range) {
  hi16 = pack_h16x2<ET_F16>(v0, v1);
  range |= q_range_bits(hi16);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi16), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi16), "v"(v1));
  hi8 = __builtin_amdgcn_cvt_pk_bf8_f32(v0, v1, hi8, SEL);
  lo8 = __builtin_amdgcn_cvt_pk_bf8_f32(r0 * 2048.0f, r1 * 2048.0f, lo8, SEL);
}

// XIMG: the input rows are IMAGES already (p.x_image: written by this kernel's image epilogue, p.y_image, in the producing layer - see the
// epilogue): the window of a chunk goes by LDS-DMA straight into one of FOUR image slots (ring), no conversion pass, no f32 stage.
template <bool GENERIC, bool XIMG>
__global__ __launch_bounds__(256, 2) void tdnn_gemm_x3m_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[QRING + 3 * 256 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int scale_w = lh ? kQScaleWlo : kQScaleWhi, scale_x = lh ? kQScaleXhi : kQScaleXlo;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * QBM;
  const int n0 = (tile % n_tiles) * QBN;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *zero = reinterpret_cast<const unsigned char *>(p.zero16);
  const size_t x_pitch = (size_t)p.ldx * 4;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(x3m_lds_byte *)lds);
  const int g_row = lane >> 3, g_slot = lane & 7;

  const int nchunks = (p.cin_pad + QBK - 1) / QBK;
  const int n_taps = p.n_taps;
  const int nkg = ((p.cin_pad + 63) / 64) * 4;                  // 16-channel k-groups per tap in the half fragment array

  float *lds_par = reinterpret_cast<float *>(lds + QRING);
  if (tid < 192) {
    const int which = tid >> 6, idx = (tid & 63) * 4;
    float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *src = (which == 0) ? p.bias : (which == 1 ? p.scale : p.shift);
    if (src != nullptr) v = *reinterpret_cast<const float4 *>(src + n0 + idx);
    *reinterpret_cast<float4 *>(lds_par + which * 256 + idx) = v;
  }

  // feature window of chunk c -> f32 stage c & 1 (clamped rows are zero gap rows; the channels of a last, partial chunk beyond cin_pad come
  // from the zero page)
  size_t a_off[QPIECES];
#pragma unroll
  for (int i = 0; i < QPIECES; ++i) {
    const int grp = min(wn + i * 4, QGROUPS - 1);
    const int w = grp * 8 + g_row;
    const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
    a_off[i] = (size_t)row * x_pitch + (size_t)qswz(w, g_slot) * 16u;
  }
  auto issue_A = [&](int c) {
    const unsigned char *base = xg + (size_t)c * QROWB;
    const bool tail = (c + 1) * QBK > p.cin_pad;
#pragma unroll
    for (int i = 0; i < QPIECES; ++i) {
      const int grp = min(wn + i * 4, QGROUPS - 1);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (XIMG ? (c & 3) : (c & 1)) * QSTAGE + grp * 1024);
      const int w = grp * 8 + g_row;
      const bool ok = !tail || (c * QBK + qswz(w, g_slot) * 4 < p.cin_pad);
      x3m_glds16(ok ? base + a_off[i] : zero, dst);
    }
  };
  uint32_t range = 0u;
  // f32 stage c & 1 -> image c & 1 (at 2 * QSTAGE): row = [hi halves of 32 channels (slots 0-3) | x_lo8 (slots 4, 5) | x_hi8 (slots 6, 7)]; item
  // (w, q) = channels 8 q .. 8 q + 7 of row w
  auto convert = [&](int c) {
    const unsigned char *src = lds + (c & 1) * QSTAGE;
    unsigned char *dst = lds + (2 + (c & 1)) * QSTAGE;
#pragma unroll
    for (int it = 0; it < (QWIN * 4 + 255) / 256; ++it) {
      const int item = it * 256 + tid;
      if (item < QWIN * 4) {
        const int w = item >> 2, q = item & 3;
        const uint4 a = *reinterpret_cast<const uint4 *>(src + w * QROWB + qswz(w, 2 * q) * 16);
        const uint4 b = *reinterpret_cast<const uint4 *>(src + w * QROWB + qswz(w, 2 * q + 1) * 16);
        uint4 hi;
        int h8a = 0, l8a = 0, h8b = 0, l8b = 0;
        q_split<false>(__uint_as_float(a.x), __uint_as_float(a.y), hi.x, h8a, l8a, range);
        q_split<true>(__uint_as_float(a.z), __uint_as_float(a.w), hi.y, h8a, l8a, range);
        q_split<false>(__uint_as_float(b.x), __uint_as_float(b.y), hi.z, h8b, l8b, range);
        q_split<true>(__uint_as_float(b.z), __uint_as_float(b.w), hi.w, h8b, l8b, range);
        *reinterpret_cast<uint4 *>(dst + w * QROWB + qswz(w, q) * 16) = hi;
        // byte b of an 8-bit slot of lane half lh = channel (b < 8 ? 8 lh + b : 16 + 8 lh + b - 8): the order of the lane's two half fragments
        *reinterpret_cast<uint2 *>(dst + w * QROWB + qswz(w, 4 + (q & 1)) * 16 + (q >> 1) * 8) = make_uint2((uint32_t)l8a, (uint32_t)l8b);
        *reinterpret_cast<uint2 *>(dst + w * QROWB + qswz(w, 6 + (q & 1)) * 16 + (q >> 1) * 8) = make_uint2((uint32_t)h8a, (uint32_t)h8b);
      }
    }
  };

  const size_t frag_stride = (size_t)n_taps * nkg * 1024;                     // half fragments: bytes per 32-channel output fragment
  const size_t frag8_stride = (size_t)n_taps * nchunks * 1024;                // 8-bit fragments (w_lo8): [tap][32-channel group][lane][16]
  const unsigned char *wh = reinterpret_cast<const unsigned char *>(p.wfrag) + (size_t)((n0 + wn * 64) / 32) * frag_stride + (size_t)lane * 16;
  const unsigned char *w8 = reinterpret_cast<const unsigned char *>(p.w8) + (size_t)((n0 + wn * 64) / 32) * frag8_stride + (size_t)lane * 16;

  // Operand registers (the accumulators take 128 of the 256): the weight fragments of the three phases each have their own registers and are
  // re-fetched from L2 right after their phase (two phases ahead of their next use); the LDS operands rotate through two sets of four,
  // fetched one phase ahead.
  uint4 wh0[2], wh1[2], we[2];               // this wave's two 32-channel fragments: k-group 0, k-group 1 (halves), the pair's 8-bit K block 1 (w_lo8)
  uint4 wq[2];                               // K block 0 (w_hi8): made in registers from wh0 (bytes 0-7) and wh1 (bytes 8-15), no memory traffic
  uint4 xa[4], xb[4];
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // instruction q (0..7) of a main phase: accumulator (i, j) = (q & 3, q >> 2); an accumulator recurs every 8th instruction
  auto mma_main = [&](const uint4 (&w)[2], const uint4 (&x)[4], int q) {
    const int i = q & 3, j = q >> 2;
    acc[i][j] = mfma16<ET_F16>(w[j], x[i], acc[i][j]);
  };
  // the 8 half values of a weight fragment -> e4m3(w_hi 2^-6) in two registers (RNE: the bytes a host packer would write)
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
  // corrections of accumulator (i, j): x0 / x1 = the frame fragment's K blocks 0 (x_lo8) / 1 (x_hi8)
  auto mma_mx = [&](int i, int j, const uint4 &x0, const uint4 &x1) {
    const q_v8i a = {(int)wq[j].x, (int)wq[j].y, (int)wq[j].z, (int)wq[j].w, (int)we[j].x, (int)we[j].y, (int)we[j].z, (int)we[j].w};
    const q_v8i b = {(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 1, 0, scale_w, 0, scale_x);      // e4m3 weights, e5m2 activations
  };

  const int v_taps = p.taps[lane < 9 ? lane : 0];
  // LDS byte address of this lane's first row of image c & 1 for tap t, and its swizzle term (blind to + 32 rows)
  auto x_row = [&](int c, int t, uint32_t &base, int &sw) {
    const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t);
    sw = (wrow >> 1) & 7;
    base = (uint32_t)((XIMG ? (c & 3) : 2 + (c & 1)) * QSTAGE + wrow * QROWB);
  };

  // ---- prologue: windows 0 and 1 in flight; window 0 -> image 0; its stage takes window 2; window 1 -> image 1
  issue_A(0);
  if (nchunks > 1) issue_A(1);
  if constexpr (XIMG) {
    // images 0, 1, 2 in flight; image 0 has landed when all but the youngest 2 x QPIECES operations have
    if (nchunks > 2) issue_A(2);
    if (nchunks > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QPIECES) : "memory");
    else if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  } else {
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    convert(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nchunks > 2) issue_A(2);
    if (nchunks > 1) convert(1);
  }
  {
    wh0[0] = *reinterpret_cast<const uint4 *>(wh); wh0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride);
    wh1[0] = *reinterpret_cast<const uint4 *>(wh + 1024); wh1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + 1024);
    we[0] = *reinterpret_cast<const uint4 *>(w8); we[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride);
    uint32_t base; int sw;
    x_row(0, 0, base, sw);
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const uint4 *>(lds + base + (uint32_t)((lh ^ sw) << 4) + i * 32 * QROWB);
  }

  const int P = nchunks * n_taps;
  int c = 0, t = 0;
#pragma unroll 1
  for (int n = 0; n < P; ++n) {
    int c2 = c, t2 = t + 1;
    if (t2 == n_taps) { t2 = 0; c2 = c + 1; }
    const bool more = n + 1 < P;
    if (!more) { c2 = c; t2 = t; }                                 // the last pair re-fetches itself (valid memory, never used)
    const bool enter = more && c2 != c;
    uint32_t base; int sw;
    x_row(c, t, base, sw);
    const size_t offh2 = ((size_t)t2 * nkg + (size_t)c2 * 2) * 1024;
    const size_t off82 = ((size_t)t2 * nchunks + c2) * 1024;
    // phase 1: k-group 0 from (wh0, xa); the pair's 8-bit weights were requested one step ago - the NEXT pair's are requested behind phase 3
    {
      const uint32_t a = base + (uint32_t)(((2 + lh) ^ sw) << 4);                     // k-group 1 of this pair -> xb
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < 4) xb[q] = *reinterpret_cast<const uint4 *>(lds + a + q * 32 * QROWB);
        if (q == 5) { whi8(wh0[0], wq[0].x, wq[0].y); whi8(wh0[1], wq[1].x, wq[1].y); }
        mma_main(wh0, xa, q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // phase 2: k-group 1 from (wh1, xb); wh0 <- k-group 0 of the next pair; xa <- the 8-bit K blocks of frame fragments 0, 1
    {
      const uint32_t a0 = base + (uint32_t)(((4 + lh) ^ sw) << 4), a1 = base + (uint32_t)(((6 + lh) ^ sw) << 4);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q == 0) { wh0[0] = *reinterpret_cast<const uint4 *>(wh + offh2); wh0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh2); }
        if (q == 1) { xa[0] = *reinterpret_cast<const uint4 *>(lds + a0); xa[1] = *reinterpret_cast<const uint4 *>(lds + a1); }
        if (q == 2) { xa[2] = *reinterpret_cast<const uint4 *>(lds + a0 + 32 * QROWB); xa[3] = *reinterpret_cast<const uint4 *>(lds + a1 + 32 * QROWB); }
        if (q == 5) { whi8(wh1[0], wq[0].z, wq[0].w); whi8(wh1[1], wq[1].z, wq[1].w); }
        mma_main(wh1, xb, q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // phase 3, first half: corrections of frame fragments 0, 1 from (we, xa); wh1 <- k-group 1 of the next pair; xb <- the K blocks of fragments 2, 3
    {
      const uint32_t a0 = base + (uint32_t)(((4 + lh) ^ sw) << 4) + 64 * QROWB, a1 = base + (uint32_t)(((6 + lh) ^ sw) << 4) + 64 * QROWB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q == 0) { wh1[0] = *reinterpret_cast<const uint4 *>(wh + offh2 + 1024); wh1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh2 + 1024); }
        if (q == 0) { xb[0] = *reinterpret_cast<const uint4 *>(lds + a0); xb[1] = *reinterpret_cast<const uint4 *>(lds + a1); }
        if (q == 1) { xb[2] = *reinterpret_cast<const uint4 *>(lds + a0 + 32 * QROWB); xb[3] = *reinterpret_cast<const uint4 *>(lds + a1 + 32 * QROWB); }
        mma_mx(q & 1, q >> 1, xa[2 * (q & 1)], xa[2 * (q & 1) + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (enter) {
      // Entering chunk c + 1 (the protocol of kernels_tdnn_x3.hip's SHARED form, in the middle of phase 3): every read of image c has been
      // issued - the last ones, the K blocks of fragments 2 and 3, just above.  Window c + 2 has landed: it is older than the youngest 4
      // vector-memory operations (wh0 and wh1 of the next pair).  Behind the barrier nobody reads image c any more: window c + 2 becomes
      // image c & 1, and its stage takes window c + 3.
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // (XIMG: window c + 1 - issued three chunks ago - has landed for the same reason; window c + 3 goes into the slot of image c - 1)
      if constexpr (!XIMG) { if (c + 2 < nchunks) convert(c + 2); }
      if (c + 3 < nchunks) issue_A(c + 3);
    }
    // phase 3, second half: fragments 2, 3 from (we, xb); xa <- k-group 0 of the next pair; then we <- the next pair's 8-bit weights
    {
      uint32_t base2; int sw2;
      x_row(c2, t2, base2, sw2);
      const uint32_t a = base2 + (uint32_t)((lh ^ sw2) << 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xa[q] = *reinterpret_cast<const uint4 *>(lds + a + q * 32 * QROWB);
        mma_mx(2 + (q & 1), q >> 1, xb[2 * (q & 1)], xb[2 * (q & 1) + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      we[0] = *reinterpret_cast<const uint4 *>(w8 + off82); we[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off82);
    }
    c = c2; t = t2;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // the ring becomes epilogue scratch
  asm volatile("" ::: "memory");

  // ---- epilogue (kernels_tdnn_x3.hip): acc[i][j][r]: frame = m0 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  float *scr = reinterpret_cast<float *>(lds) + wn * (32 * QSPITCH);
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;
  float *yg = reinterpret_cast<float *>(p.y);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool valid = (p.row_valid[(m0 + i * 32) >> 5] >> lr) & 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;
        const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
        const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
        const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 512 + chl);
        const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        float y[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          if constexpr (GENERIC) {
            float z = fmaf(acc[i][j][q * 4 + e2], unscale, b[e2]);
            z = p.affine_first ? apply_act(z * sc[e2] + sh[e2], p.act1) : apply_act(z, p.act1) * sc[e2] + sh[e2];
            z = apply_act(z, p.act2);
            y[e2] = valid ? z : 0.0f;
          } else {
            const float z = fmaxf(fmaf(acc[i][j][q * 4 + e2], unscale, b[e2]), act_lo) * sc[e2] + sh[e2];
            y[e2] = valid ? z : 0.0f;
          }
        }
        if (p.y_image) {
          // image output (the reader is an f32m kernel that takes images: the split it would do per workgroup is done here, once): the 128
          // bytes of (row, 32-channel group) = [hi halves 64 B | x_lo8 32 B | x_hi8 32 B], bytes as q_split / the readers' conversion
          // pass would write them - the reader's results are bit-identical to those from the f32 row
          uint2 hi;
          int h8 = 0, l8 = 0;
          q_split<false>(y[0], y[1], hi.x, h8, l8, range);
          q_split<true>(y[2], y[3], hi.y, h8, l8, range);
          unsigned char *rowb = reinterpret_cast<unsigned char *>(scr + lr * QSPITCH) + j * 128;
          *reinterpret_cast<uint2 *>(rowb + q * 16 + lh * 8) = hi;
          *reinterpret_cast<uint32_t *>(rowb + 64 + (q & 1) * 16 + (q >> 1) * 8 + lh * 4) = (uint32_t)l8;
          *reinterpret_cast<uint32_t *>(rowb + 96 + (q & 1) * 16 + (q >> 1) * 8 + lh * 4) = (uint32_t)h8;
        } else {
          *reinterpret_cast<float4 *>(scr + lr * QSPITCH + j * 32 + 8 * q + 4 * lh) = make_float4(y[0], y[1], y[2], y[3]);
        }
      }
    // the scratch tile is wave-private: LDS operations of one wave complete in order
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int frow = it * 4 + (lane >> 4), slot = lane & 15;
      const float4 v = *reinterpret_cast<const float4 *>(scr + frow * QSPITCH + slot * 4);
      const int ch = n0 + wn * 64 + slot * 4;
      const int row = m0 + i * 32 + frow;
      if (ch < p.cout_store) *reinterpret_cast<float4 *>(yg + (size_t)row * p.ldy + ch) = v;
    }
  }
  x3_publish_range(range, p.status);       // the window conversions' watch and, with image output, the epilogue's
}

}  // namespace

// the layers this form takes: what the 128-row three-product kernel takes with the plain / generic epilogue (no fused pooling), 8-bit fragments present
bool tdnn_x3m_supported(const TdnnKernelParams &p) {
  return tdnn_x3_supported(p) && p.w8 != nullptr && p.x3_et == ET_F16 && (p.x3_terms & 7) == 7 && p.rows % 128 == 0 && p.pool_partial == nullptr && p.w_unscale > 0.0f;
}
// image rows in / out: whole 32-channel groups on both sides
bool tdnn_x3m_image_in_supported(const TdnnKernelParams &p) { return p.cin_pad % 32 == 0 && p.x2 == nullptr; }
bool tdnn_x3m_image_out_supported(const TdnnKernelParams &p) { return p.cout_store % 32 == 0 && p.res == nullptr; }

int launch_tdnn_x3m(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(tdnn_x3m_supported(p), "tdnn(x3m): layer not supported");
  for (int t = 0; t < p.n_taps; ++t)
    ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(x3m): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const int m_tiles = p.rows / QBM, n_tiles = round_up(p.cout_store, QBN) / QBN;
  const dim3 grid(m_tiles * n_tiles), block(256);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  ASV_REQUIRE(!p.x_image || tdnn_x3m_image_in_supported(p), "tdnn(x3m): image input needs whole 32-channel groups");
  ASV_REQUIRE(!p.y_image || tdnn_x3m_image_out_supported(p), "tdnn(x3m): image output needs whole 32-channel groups");
  if (p.x_image) {
    if (fast) hipLaunchKernelGGL((tdnn_gemm_x3m_kernel<false, true>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_x3m_kernel<true, true>), grid, block, 0, s, p, m_tiles, n_tiles);
  } else {
    if (fast) hipLaunchKernelGGL((tdnn_gemm_x3m_kernel<false, false>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_x3m_kernel<true, false>), grid, block, 0, s, p, m_tiles, n_tiles);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
