// kernels_tdnn_chainm.hip with 96 frames per workgroup instead of 64: the f32m layer chain tdnn -> [1-tap 512 -> 512]* -> 1-tap + fused
// statistics pooling (model/xvector.py:77-98; components.py:107-149, 410-431; pooling.py:58-67), same products, same split
//     w x  ~  w_hi x_hi (half instruction) + [e4m3(w_hi) . e5m2(x_lo) | e4m3(w_lo) . e5m2(x_hi)] (one block-scaled 8-bit instruction per 32 channels).
//
// MEASURED AND NOT THE DEFAULT (ASV_AMD_CHAINM_ROWS=96 selects it): the results are the 64-frame kernel's (tests/test_gpu_f32m.py passes with
// either), the time is 13 % WORSE - 507.6 us against 450.1 us for the 257 k-frame chain on one box, twice each (profiles/r6r_chainm96_ab.txt);
// a K step of layer A takes 2250 - 2870 cycles per wave where the 64-frame kernel takes 1250 - 1900, i.e. MORE than the 1.5 x of its matrix
// work.  The hypothesis it was built on was therefore wrong in the form stated below: the step does not cost "the weight fetch, whatever the
// rows" - it grows with the work of the step, the x_hi8 conversions (24 per step here) included.
//
// The hypothesis: the 64-frame kernel's K loops run at the pace of its WEIGHT STREAM, not of the matrix pipe (every CU streams the chain's
// whole weight set per tile; profiles/r6i, r6o, r6p) - and the bytes per tile do not depend on the tile's rows.  96 frames per workgroup =
// 1.5 x the matrix work per streamed byte.  What makes it fit:
//   * LDS: the resident tile is its half image Yh (96 x 1 KiB) + ONLY the 8-bit residual image x_lo8 (96 x 512 B); the other 8-bit row block,
//     e5m2(x) = x_hi8, is made in registers from the half rows the main product has loaded anyway (v_cvt_scalef32_pk_bf8_f16: a second rounding
//     of an 11-bit value to 3 bits - the correction it feeds is 2^-11 of a product) exactly as e4m3(w_hi) is made from the half weight fragments:
//     144 KiB + 8 KiB of constants.
//   * registers: 96 accumulators (3 x 2 fragments of 32 x 32 per wave) + ONE set of weight fragments, each re-fetched from L2 right behind the
//     phase that used it (a phase is 1.5 x longer than in the 64-frame kernel: the fetch distance is what its two register sets gave), the
//     8-bit weight block alone double-buffered.
// Tiles of 96 rows: the partial moments go to pool_finish_kernel as uniform 96-row blocks (its tail_rows form); the last tile may overhang the
// matrix (rows beyond it: clamped window rows = zero gap rows, no utterance).  Everything not said here: kernels_tdnn_chainm.hip.
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int NM = 96;                     // frames per workgroup
constexpr int NF = 3;                      // 32-frame fragments per wave
constexpr int NN = kChainWidth;            // channels of the resident tile (512)
constexpr int NROW = 128;                  // window row: 32 f32; image row: [hi16 64 B | x_lo8 32 B | unused 32 B]
constexpr int NWINR = NM + 2 * kHalo;      // 104 window rows
constexpr int NGRP = NWINR / 8;            // 13 eight-row DMA pieces
constexpr int NSTG = NWINR * NROW;         // 13312 B per window buffer (f32 stage, then image, in place)
constexpr int NYROW = NN * 2;              // 1024 B per row of Yh
constexpr int NY8ROW = NN;                 // 512 B per row of Y8 (x_lo8 only): 32-channel group g = slots 2 g (lane half 0), 2 g + 1
constexpr int NYIMG = NM * NYROW;          // 98304 B: Yh at 0
constexpr int NY8 = NM * NY8ROW;           // 49152 B: Y8 at NYIMG
constexpr int NPAR = NYIMG + NY8;          // bias | scale | shift of the layer in flight (6 KiB)
constexpr int CHAINM96_LDS = NPAR + 8192;
static_assert(NWINR % 8 == 0 && 4 * NSTG <= NYIMG, "layer A's window buffers live inside the Yh region");
static_assert(CHAINM96_LDS <= 163840, "160 KiB of LDS per CU");

constexpr int kNScaleWhi = 127 + 6, kNScaleWlo = 127 - 6, kNScaleXlo = 127 - 11, kNScaleXhi = 127;      // kernels_tdnn_chainm.hip

typedef __attribute__((address_space(3))) unsigned char chainm96_lds_byte;
typedef int n_v8i __attribute__((ext_vector_type(8)));
typedef short n_s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 n_h16x2 __attribute__((ext_vector_type(2)));
struct NTrNo { static constexpr bool value = false; };
struct NTrYes { static constexpr bool value = true; };

__device__ __forceinline__ int nswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

__device__ __forceinline__ void chainm96_glds16(const void *gsrc, uint32_t lds_dst) {
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

__device__ __forceinline__ uint32_t n_range_bits(uint32_t packed_hi) { return (packed_hi & 0x7fff7fffu) + 0x05000500u; }      // |half| >= 57344

// two f32 -> the packed pair of hi halves and the pair e5m2((x - hi) 2^11) (low 16 bits of lo8 when SEL = false, high 16 bits otherwise)
template <bool SEL>
__device__ __forceinline__ void n_split(float v0, float v1, uint32_t &hi16, int &lo8, uint32_t &range) {
  hi16 = pack_h16x2<ET_F16>(v0, v1);
  range |= n_range_bits(hi16);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi16), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi16), "v"(v1));
  lo8 = __builtin_amdgcn_cvt_pk_bf8_f32(r0 * 2048.0f, r1 * 2048.0f, lo8, SEL);
}

__global__ __launch_bounds__(512, 2) void tdnn_chainm96_kernel(const TdnnChainParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[CHAINM96_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * NM;
  uint32_t range = 0u;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(chainm96_lds_byte *)lds);
  float *par = reinterpret_cast<float *>(lds + NPAR);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const int scale_w = lh ? kNScaleWlo : kNScaleWhi, scale_x = lh ? kNScaleXhi : kNScaleXlo;

  // developer aid (ASV_AMD_CHAIN_DBG=1): [workgroup][wave][32] s_memtime stamps at the phase boundaries; 14 / 15: s_memrealtime at start / end
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
      *reinterpret_cast<float4 *>(par + which * NN + idx) = v;
    }
  };

  // Operand registers of a 32-channel pair: this wave's two 32-channel weight fragments as halves for the two k-groups (wh0, wh1: one set, each
  // re-fetched right behind its phase) and as K block 1 of the scaled instruction (w_lo8: two sets, a step ahead); the three frame fragments as
  // half rows per k-group (h0x, h1x) and as K block 0 (xl = x_lo8, from LDS); wq / xq = e4m3(w_hi) / e5m2(x), made here from wh* / h*x.
  uint4 wh0[2], wh1[2], wq[2];
  uint4 h0x[NF], h1x[NF], xl[NF], xq[NF];
  f32x16_t acc[NF][2];
  auto init_acc = [&](const float *bias64, float w_scale, auto tr) {
    if constexpr (decltype(tr)::value) {
      const float b0 = bias64[lr] * w_scale, b1 = bias64[32 + lr] * w_scale;
#pragma unroll
      for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = b0; acc[i][1][r] = b1; }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4 *>(bias64 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            acc[i][j][q * 4 + 0] = b4.x * w_scale; acc[i][j][q * 4 + 1] = b4.y * w_scale;
            acc[i][j][q * 4 + 2] = b4.z * w_scale; acc[i][j][q * 4 + 3] = b4.w * w_scale;
          }
        }
    }
  };
  // instruction q (0..5) of a phase: accumulator (i, j) = (q % 3, q / 3); an accumulator recurs every 6th instruction
  auto mma_main = [&](const uint4 (&w)[2], const uint4 (&x)[NF], int q, auto tr) {
    const int i = q % 3, j = q / 3;
    if constexpr (decltype(tr)::value) acc[i][j] = mfma16<ET_F16>(x[i], w[j], acc[i][j]);
    else acc[i][j] = mfma16<ET_F16>(w[j], x[i], acc[i][j]);
  };
  // 8 half values -> 8 bytes (two registers): e4m3(w 2^-6) of a weight fragment / e5m2(x) of a row fragment (RNE)
  auto whi8 = [&](const uint4 &f, uint32_t &d0, uint32_t &d1) {
    n_s16x2 a = {0, 0}, b = {0, 0};
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, __builtin_bit_cast(n_h16x2, f.x), 64.0f, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a, __builtin_bit_cast(n_h16x2, f.y), 64.0f, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, __builtin_bit_cast(n_h16x2, f.z), 64.0f, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b, __builtin_bit_cast(n_h16x2, f.w), 64.0f, true);
    d0 = __builtin_bit_cast(uint32_t, a); d1 = __builtin_bit_cast(uint32_t, b);
  };
  auto xhi8 = [&](const uint4 &f, uint32_t &d0, uint32_t &d1) {
    n_s16x2 a = {0, 0}, b = {0, 0};
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(a, __builtin_bit_cast(n_h16x2, f.x), 1.0f, false);
    a = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(a, __builtin_bit_cast(n_h16x2, f.y), 1.0f, true);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(b, __builtin_bit_cast(n_h16x2, f.z), 1.0f, false);
    b = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(b, __builtin_bit_cast(n_h16x2, f.w), 1.0f, true);
    d0 = __builtin_bit_cast(uint32_t, a); d1 = __builtin_bit_cast(uint32_t, b);
  };
  auto mma_mx = [&](const uint4 (&we)[2], int q, auto tr) {
    const int i = q % 3, j = q / 3;
    const n_v8i a = {(int)wq[j].x, (int)wq[j].y, (int)wq[j].z, (int)wq[j].w, (int)we[j].x, (int)we[j].y, (int)we[j].z, (int)we[j].w};
    const n_v8i b = {(int)xl[i].x, (int)xl[i].y, (int)xl[i].z, (int)xl[i].w, (int)xq[i].x, (int)xq[i].y, (int)xq[i].z, (int)xq[i].w};
    // operand formats: 0 = e4m3 (weights), 1 = e5m2 (activations)
    if constexpr (decltype(tr)::value) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc[i][j], 1, 0, 0, scale_x, 0, scale_w);
    else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i][j], 0, 1, 0, scale_w, 0, scale_x);
  };
  auto prio = [&](int n) {
    if ((p.abl & 8) == 0) { if (((n ^ (wave >> 2)) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
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
    // window of chunk c -> buffer c % 4 (four buffers of 104 rows x 128 B inside the not yet written Yh region): 13 eight-row pieces, piece w by wave w, pieces 8 .. 12 by waves 0 .. 4 (rows beyond the matrix ends are
    // clamped onto zero gap rows); converted in place between the matrix instructions of the step behind the chunk barrier
    auto piece_off = [&](int grp) -> size_t {
      const int w = grp * 8 + g_row;
      const int row = min(max(m0 - kHalo + w, 0), p.rows - 1);
      return (size_t)row * x_pitch + (size_t)nswz(w, g_slot) * 16u;
    };
    const size_t off_a = piece_off(wave), off_b = piece_off(min(8 + wave, NGRP - 1));
    auto issue_A = [&](int c, int buf) {
      const unsigned char *base = xg + (size_t)c * NROW;
      chainm96_glds16(base + off_a, __builtin_amdgcn_readfirstlane(lds_base + buf * NSTG + wave * 1024));
      if (wave < NGRP - 8) chainm96_glds16(base + off_b, __builtin_amdgcn_readfirstlane(lds_base + buf * NSTG + (8 + wave) * 1024));
    };
    // f32 rows -> image rows [hi halves of 32 channels (slots 0-3) | x_lo8 (slots 4, 5 = lane halves 0, 1)], in place; thread (w, q) converts
    // channels 8 q .. 8 q + 7 of row w (the four threads of a row sit in one wave: their reads retire before any of them writes)
    uint4 cva = make_uint4(0, 0, 0, 0), cvb = make_uint4(0, 0, 0, 0);
    auto cv_load = [&](int buf) {
      if (tid < NWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        const unsigned char *src = lds + buf * NSTG + w * NROW;
        cva = *reinterpret_cast<const uint4 *>(src + nswz(w, 2 * q) * 16);
        cvb = *reinterpret_cast<const uint4 *>(src + nswz(w, 2 * q + 1) * 16);
      }
    };
    auto cv_store = [&](int buf) {
      if (tid < NWINR * 4) {
        const int w = tid >> 2, q = tid & 3;
        uint4 hi;
        int l8a = 0, l8b = 0;
        n_split<false>(__uint_as_float(cva.x), __uint_as_float(cva.y), hi.x, l8a, range);
        n_split<true>(__uint_as_float(cva.z), __uint_as_float(cva.w), hi.y, l8a, range);
        n_split<false>(__uint_as_float(cvb.x), __uint_as_float(cvb.y), hi.z, l8b, range);
        n_split<true>(__uint_as_float(cvb.z), __uint_as_float(cvb.w), hi.w, l8b, range);
        unsigned char *dst = lds + buf * NSTG + w * NROW;
        *reinterpret_cast<uint4 *>(dst + nswz(w, q) * 16) = hi;
        // byte b of the 8-bit slot of lane half lh = channel (b < 8 ? 8 lh + b : 16 + 8 lh + b - 8): the order of the lane's two half fragments
        *reinterpret_cast<uint2 *>(dst + nswz(w, 4 + (q & 1)) * 16 + (q >> 1) * 8) = make_uint2((uint32_t)l8a, (uint32_t)l8b);
      }
    };
    auto off_h = [&](int c, int t) -> size_t { return ((size_t)t * nkg + (size_t)c * 2) * 1024; };
    auto off_8 = [&](int c, int t) -> size_t { return ((size_t)t * nchunks + c) * 1024; };
    const size_t frag_stride = (size_t)n_taps * nkg * 1024;                    // half fragments: bytes per 32-channel output fragment
    const size_t frag8_stride = (size_t)n_taps * nchunks * 1024;               // 8-bit fragments (w_lo8): [tap][32-channel group][lane][16]
    const unsigned char *wh = reinterpret_cast<const unsigned char *>(p.first.wfrag) + (size_t)(wave * 2) * frag_stride + lane16;
    const unsigned char *w8 = reinterpret_cast<const unsigned char *>(p.first.w8) + (size_t)(wave * 2) * frag8_stride + lane16;
    const int v_taps = p.taps[lane < 9 ? lane : 0];
    // LDS byte address of this lane's row of the image in buffer `buf` for tap t, and its swizzle term (blind to + 32 rows)
    auto x_row = [&](int buf, int t, uint32_t &base, int &sw) {
      const int wrow = lr + kHalo + __builtin_amdgcn_readlane(v_taps, t);
      sw = (wrow >> 1) & 7;
      base = (uint32_t)(buf * NSTG + wrow * NROW);
    };
    // prologue: windows 0, 1, 2 in flight; 0 and 1 converted at once
    issue_A(0, 0);
    if (nchunks > 1) issue_A(1, 1);
    if (nchunks > 2) issue_A(2, 2);
    uint4 we0[2], we1[2];
    wh0[0] = *reinterpret_cast<const uint4 *>(wh); wh0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride);
    wh1[0] = *reinterpret_cast<const uint4 *>(wh + 1024); wh1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + 1024);
    we0[0] = *reinterpret_cast<const uint4 *>(w8); we0[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cv_load(0); cv_store(0);
    if (nchunks > 1) { cv_load(1); cv_store(1); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    init_acc(p.first.bias + wave * 64, p.first.w_scale, NTrNo{});
    {
      uint32_t base; int sw;
      x_row(0, 0, base, sw);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        h0x[i] = *reinterpret_cast<const uint4 *>(lds + base + (uint32_t)((lh ^ sw) << 4) + i * 32 * NROW);
        h1x[i] = *reinterpret_cast<const uint4 *>(lds + base + (uint32_t)(((2 + lh) ^ sw) << 4) + i * 32 * NROW);
      }
    }
    const int P = nchunks * n_taps;
    int c = 0, t = 0, cb = 0;                                     // cb = c % 4: the buffer of chunk c's image
    stamp();                                                     // 1: layer A's prologue
    // pair n = (chunk c, tap t): 6 + 6 half instructions and 6 scaled ones
    auto step = [&](const uint4 (&wec)[2], uint4 (&wen)[2], int n) {
      if (p.dbg != nullptr && p.dbg_fine && lane == 0 && n < 16) p.dbg[((size_t)blockIdx.x * 8 + wave) * 32 + 16 + n] = __builtin_amdgcn_s_memtime();
      prio(n);
      int c2 = c, t2 = t + 1;
      if (t2 == n_taps) { t2 = 0; c2 = c + 1; }
      const bool more = n + 1 < P;
      if (!more) { c2 = c; t2 = t; }                               // the last pair re-fetches itself (valid memory, never used)
      const bool enter = more && c2 != c;
      const int cb1 = (cb + 1) & 3, cb2 = (cb + 2) & 3, cb3 = (cb + 3) & 3;
      const bool cv = enter && c + 2 < nchunks;
      if (enter) {
        // Entering chunk c + 1 at the start of the LAST step of chunk c (kernels_tdnn_chainm.hip): image c + 1 is complete, window c + 2 has
        // landed and is converted in place between this step's matrix instructions.  Unlike the 64-frame kernel this step still READS image c
        // (the x_lo8 rows of its own pair, phase 1) - so the next window, c + 3, does not go into image c's buffer but into a FOURTH one,
        // image c - 1's, whose last reads (the same phase-1 reads, one chunk ago) every wave has retired before this barrier.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      const size_t off8 = off_8(c2, t2);
      const size_t offh = off_h(c2, t2);
      uint32_t base, basec; int sw, swc;
      x_row(c2 != c ? cb1 : cb, t2, base, sw);
      x_row(cb, t, basec, swc);
      const uint32_t axl = basec + (uint32_t)(((4 + lh) ^ swc) << 4);
      const uint32_t ah0 = base + (uint32_t)(((lh) ^ sw) << 4), ah1 = base + (uint32_t)(((2 + lh) ^ sw) << 4);
      // phase 1: k-group 0; THIS pair's x_lo8 rows, the NEXT pair's 8-bit weights; e4m3 / e5m2 of k-group 0's operands
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q == 0) { wen[0] = *reinterpret_cast<const uint4 *>(w8 + off8); wen[1] = *reinterpret_cast<const uint4 *>(w8 + frag8_stride + off8); }
        if (q < 3) xl[q] = *reinterpret_cast<const uint4 *>(lds + axl + q * 32 * NROW);
        if (q == 1 && enter && c + 3 < nchunks) issue_A(c + 3, cb3);
        if (q == 1 && cv) cv_load(cb2);
        if (q == 3) { whi8(wh0[0], wq[0].x, wq[0].y); whi8(wh0[1], wq[1].x, wq[1].y); }
        if (q == 4) { xhi8(h0x[0], xq[0].x, xq[0].y); xhi8(h0x[1], xq[1].x, xq[1].y); xhi8(h0x[2], xq[2].x, xq[2].y); }
        mma_main(wh0, h0x, q, NTrNo{});
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 2: k-group 1; k-group 0 of the next pair into the registers phase 1 has just read
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        mma_main(wh1, h1x, q, NTrNo{});
        if (q == 0) { wh0[0] = *reinterpret_cast<const uint4 *>(wh + offh); wh0[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh); }
        if (q >= 1 && q < 4) h0x[q - 1] = *reinterpret_cast<const uint4 *>(lds + ah0 + (q - 1) * 32 * NROW);
        if (q == 2) { whi8(wh1[0], wq[0].z, wq[0].w); whi8(wh1[1], wq[1].z, wq[1].w); }
        if (q == 3) { xhi8(h1x[0], xq[0].z, xq[0].w); xhi8(h1x[1], xq[1].z, xq[1].w); xhi8(h1x[2], xq[2].z, xq[2].w); }
        if (q == 4 && cv) cv_store(cb2);
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase 3: the corrections; k-group 1 of the next pair
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        mma_mx(wec, q, NTrNo{});
        if (q == 0) { wh1[0] = *reinterpret_cast<const uint4 *>(wh + offh + 1024); wh1[1] = *reinterpret_cast<const uint4 *>(wh + frag_stride + offh + 1024); }
        if (q >= 1 && q < 4) h1x[q - 1] = *reinterpret_cast<const uint4 *>(lds + ah1 + (q - 1) * 32 * NROW);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (c2 != c) cb = cb1;
      c = c2; t = t2;
    };
    for (int n = 0; n < P; n += 2) {
      step(we0, we1, n);
      if (n + 1 < P) step(we1, we0, n + 1);
    }
  }

  // epilogue of a 512-wide layer: acc / w_scale -> [ReLU] -> [folded BN unless it sits in the next layer's weights] -> hi halves into Yh, the
  // 8-bit residuals into Y8 (16-byte slots XOR-swizzled by row & 15)
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
          const float4 sc4 = *reinterpret_cast<const float4 *>(par + NN + chl);
          const float4 sh4 = *reinterpret_cast<const float4 *>(par + 2 * NN + chl);
          sc[0] = sc4.x; sc[1] = sc4.y; sc[2] = sc4.z; sc[3] = sc4.w;
          sh[0] = sh4.x; sh[1] = sh4.y; sh[2] = sh4.z; sh[3] = sh4.w;
        }
        // Yh: 4 consecutive channels = 8 bytes inside the 16-byte slot (wave * 8 + j * 4 + q), half lh
        const uint32_t slot_off = (uint32_t)((((wave * 8 + j * 4 + q) ^ rx) << 4) + lh * 8);
        // Y8: the 32-channel group (wave * 2 + j) owns slots 2 g (lane half 0) and 2 g + 1; channels 8 q + 4 lh + e of this lane belong to
        // operand lane half lhK = q & 1, bytes 8 (q >> 1) + 4 lh + e
        const int slot8 = (wave * 2 + j) * 2 + (q & 1);
        const uint32_t lo_off = (uint32_t)(((slot8 ^ rx) << 4) + (q >> 1) * 8 + lh * 4);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          float y[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = max_lo(acc[i][j][q * 4 + e] * unscale, act_lo);
            y[e] = affine ? fmaf(v, sc[e], sh[e]) : v;
          }
          uint2 hi;
          int l8 = 0;
          n_split<false>(y[0], y[1], hi.x, l8, range);
          n_split<true>(y[2], y[3], hi.y, l8, range);
          *reinterpret_cast<uint2 *>(lds + (i * 32 + lr) * NYROW + slot_off) = hi;
          *reinterpret_cast<uint32_t *>(lds + NYIMG + (i * 32 + lr) * NY8ROW + lo_off) = (uint32_t)l8;
        }
      }
  };

  // main loop of a layer whose input is Y: K = 512 = 16 pairs of 32 channels, no barrier.  wbh / wb8: wave-uniform bases of the half / 8-bit
  // fragment arrays of this wave's (or unit's) first 32-channel output fragment; the second follows at + 32 KiB / + 16 KiB.
  auto yloop = [&](const unsigned char *wbh, const unsigned char *wb8, const float *bias64, float w_scale, auto tr) {
    constexpr size_t fs = (size_t)(NN / 16) * 1024, fs8 = (size_t)(NN / 32) * 1024;
    const uint32_t yb = (uint32_t)(lr * NYROW), y8b = (uint32_t)(NYIMG + lr * NY8ROW);
    const uint32_t sx = (uint32_t)(lh ^ (lr & 15));
    uint4 we0[2], we1[2];
    wh0[0] = *reinterpret_cast<const uint4 *>(wbh + lane16); wh0[1] = *reinterpret_cast<const uint4 *>(wbh + fs + lane16);
    wh1[0] = *reinterpret_cast<const uint4 *>(wbh + 1024 + lane16); wh1[1] = *reinterpret_cast<const uint4 *>(wbh + fs + 1024 + lane16);
    we0[0] = *reinterpret_cast<const uint4 *>(wb8 + lane16); we0[1] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + lane16);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      h0x[i] = *reinterpret_cast<const uint4 *>(lds + yb + ((0u ^ sx) << 4) + i * 32 * NYROW);          // slots (2 kg + lh) ^ (lr & 15), kg = 0, 1
      h1x[i] = *reinterpret_cast<const uint4 *>(lds + yb + ((2u ^ sx) << 4) + i * 32 * NYROW);
    }
    init_acc(bias64, w_scale, tr);
    auto step = [&](const uint4 (&wec)[2], uint4 (&wen)[2], int n, int nn) {      // computes pair n; fetches pair nn
      prio(nn);
      const size_t off8 = (size_t)nn * 1024 + lane16;
      const size_t offh = (size_t)(nn * 2) * 1024 + lane16;
      const uint32_t axl = y8b + ((((uint32_t)(n * 2)) ^ sx) << 4);                 // Y8 slot (2 n + lh) ^ (lr & 15)
      const uint32_t ah0 = yb + ((((uint32_t)(nn * 4)) ^ sx) << 4), ah1 = yb + ((((uint32_t)(nn * 4 + 2)) ^ sx) << 4);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q == 0) { wen[0] = *reinterpret_cast<const uint4 *>(wb8 + off8); wen[1] = *reinterpret_cast<const uint4 *>(wb8 + fs8 + off8); }
        if (q < 3) xl[q] = *reinterpret_cast<const uint4 *>(lds + axl + q * 32 * NY8ROW);
        if (q == 3) { whi8(wh0[0], wq[0].x, wq[0].y); whi8(wh0[1], wq[1].x, wq[1].y); }
        if (q == 4) { xhi8(h0x[0], xq[0].x, xq[0].y); xhi8(h0x[1], xq[1].x, xq[1].y); xhi8(h0x[2], xq[2].x, xq[2].y); }
        mma_main(wh0, h0x, q, tr);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        mma_main(wh1, h1x, q, tr);
        if (q == 0) { wh0[0] = *reinterpret_cast<const uint4 *>(wbh + offh); wh0[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh); }
        if (q >= 1 && q < 4) h0x[q - 1] = *reinterpret_cast<const uint4 *>(lds + ah0 + (q - 1) * 32 * NYROW);
        if (q == 2) { whi8(wh1[0], wq[0].z, wq[0].w); whi8(wh1[1], wq[1].z, wq[1].w); }
        if (q == 3) { xhi8(h1x[0], xq[0].z, xq[0].w); xhi8(h1x[1], xq[1].z, xq[1].w); xhi8(h1x[2], xq[2].z, xq[2].w); }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        mma_mx(wec, q, tr);
        if (q == 0) { wh1[0] = *reinterpret_cast<const uint4 *>(wbh + offh + 1024); wh1[1] = *reinterpret_cast<const uint4 *>(wbh + fs + offh + 1024); }
        if (q >= 1 && q < 4) h1x[q - 1] = *reinterpret_cast<const uint4 *>(lds + ah1 + (q - 1) * 32 * NYROW);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int n = 0; n < NN / 32; n += 2) {
      step(we0, we1, n, n + 1);
      step(we1, we0, n + 1, min(n + 2, NN / 32 - 1));              // the last pair re-fetches itself (never used)
    }
  };

  __builtin_amdgcn_s_setprio(0);
  stamp();                                                       // 2: layer A's K loop
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done with layer A's buffers: Y may be written
  asm volatile("" ::: "memory");
  stamp();                                                       // 3: the wait for the slowest wave
  store_Y(p.first.relu, p.first.scale != nullptr, 1.0f / p.first.w_scale);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // Y complete; the constants of layer A are dead
  asm volatile("" ::: "memory");
  stamp();                                                       // 4: layer A's epilogue + barrier

  // ================================ middle layers: Y -> Y ================================
#pragma unroll 1
  for (int m = 0; m < p.n_mid; ++m) {
    const TdnnChainLayer &L = p.mid[m];
    stage_params(L);
    const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(wave * 2) * ((size_t)(NN / 16) * 1024);
    const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(wave * 2) * ((size_t)(NN / 32) * 1024);
    yloop(wbh, wb8, L.bias + wave * 64, L.w_scale, NTrNo{});
    __builtin_amdgcn_s_setprio(0);
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
    const int tile = blockIdx.x;
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && m0 + k < p.rows) first_seg = p.row_seg[m0 + k];
    // the tile's 96 rows: rows 0 .. 63 = lane l of rs_a, rows 64 .. 95 = lanes 0 .. 31 of rs_b (rows beyond the matrix: no utterance)
    const int rs_a = (m0 + lane < p.rows) ? p.row_seg[m0 + lane] : -1;
    const int rs_b = (m0 + 64 + lr < p.rows) ? p.row_seg[m0 + 64 + lr] : -1;
#pragma unroll 1
    for (int cbase = wave * 64; cbase < L.cout_pad; cbase += 512) {
      const unsigned char *wbh = reinterpret_cast<const unsigned char *>(L.wfrag) + (size_t)(cbase / 32) * ((size_t)(NN / 16) * 1024);
      const unsigned char *wb8 = reinterpret_cast<const unsigned char *>(L.w8) + (size_t)(cbase / 32) * ((size_t)(NN / 32) * 1024);
      yloop(wbh, wb8, L.bias + cbase, L.w_scale, NTrYes{});
      __builtin_amdgcn_s_setprio(0);
      stamp();                                                   // 7, 9, 11: a unit's K loop
      // Pooling epilogue, registers only (kernels_tdnn_chainx.hip, the same arithmetic): acc[i][j][r] = channel cbase + j*32 + lr, frame
      // i*32 + 8 (r >> 2) + 4 lh + (r & 3); a lane sums its own frames per utterance about the pivot of its FIRST frame of that utterance,
      // the two lane halves publish P[tile of 96 rows][segment slot][lh][3 = sum (u - pv), sum (u - pv)^2, pv][channel] with the BN scale
      // applied at publication; pool_finish_kernel merges the parts and adds the BN shift.
      const float sc[2] = {L.scale != nullptr ? L.scale[cbase + lr] : 1.0f, L.scale != nullptr ? L.scale[cbase + 32 + lr] : 1.0f};
      float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f}, pv[2] = {0.f, 0.f};
      int cur_seg = -1;                      // uniform: all lanes walk the utterances of the tile together
      bool have = false;                     // per lane: pv is a frame of cur_seg (the lane has had a frame of it in this tile)
      auto publish = [&]() {
        const int slot = cur_seg - first_seg;
        if (cur_seg >= 0 && slot >= 0 && slot < p.pool_slots) {
          float *dst = p.pool_partial + ((size_t)((tile * p.pool_slots + slot) * 2 + lh) * 3) * p.ld_partial + cbase + lr;
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (cbase + j * 32 + lr < p.ld_partial) {
              dst[j * 32] = ps[j] * sc[j];
              dst[j * 32 + p.ld_partial] = pq[j] * sc[j] * sc[j];
              dst[j * 32 + 2 * p.ld_partial] = pv[j] * sc[j];
            }
        }
      };
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int rsv = (i == 2) ? rs_b : rs_a;                  // the register that holds this fragment's rows, at lane offset `shift`
        const int shift = (i == 1) ? 32 : 0;
        uint32_t rem = (uint32_t)(__builtin_amdgcn_ballot_w64(rsv >= 0) >> shift);       // rows of the fragment that belong to an utterance
        if (rem == 0) continue;                                                              // gap rows only
        float u[2][16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) u[j][r] = max_lo(acc[i][j][r] * unscale, act_lo);
        while (rem != 0) {                                       // one run per utterance present, in row order
          const int sg = __builtin_amdgcn_readlane(rsv, shift + __builtin_ctz(rem));
          const uint32_t bits = (uint32_t)(__builtin_amdgcn_ballot_w64(rsv == sg) >> shift) & rem;
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

int chainm96_tiles(int rows) { return (rows + NM - 1) / NM; }

int launch_tdnn_chainm96(const TdnnChainParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows >= NM && p.rows % 32 == 0, "tdnn(chainm96): %d rows", p.rows);
  ASV_REQUIRE(p.cin_pad % 64 == 0 && p.cin_pad >= 64 && p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS, "tdnn(chainm96): first layer with %d channels / %d taps", p.cin_pad, p.n_taps);
  ASV_REQUIRE(p.first.wfrag && p.first.w8 && p.last.wfrag && p.last.w8 && p.last.bias && p.n_mid >= 0 && p.n_mid <= 2 && p.last.cout_pad % 64 == 0,
              "tdnn(chainm96): incomplete layer description");
  for (int m = 0; m < p.n_mid; ++m) ASV_REQUIRE(p.mid[m].wfrag && p.mid[m].w8, "tdnn(chainm96): middle layer %d without 8-bit weights", m);
  ASV_REQUIRE(p.first.w_scale > 0.0f && p.last.w_scale > 0.0f && p.et == ET_F16, "tdnn(chainm96): the half split with scaled weights only");
  ASV_REQUIRE(p.pool_partial && p.row_seg && p.pool_slots >= 1, "tdnn(chainm96): the last layer feeds the fused pooling (partials / row map missing)");
  for (int t = 0; t < p.n_taps; ++t) ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(chainm96): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const dim3 grid(chainm96_tiles(p.rows)), block(512);
  hipLaunchKernelGGL(tdnn_chainm96_kernel, grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
