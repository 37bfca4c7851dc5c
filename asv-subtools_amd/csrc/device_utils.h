// Small device-side helpers shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "asv_internal.h"

namespace asv {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

__host__ __device__ __forceinline__ int round_up_dev(int x, int m) { return (x + m - 1) / m * m; }

// Element type of frames-domain storage and of the MFMA operands.  As a template argument it takes the place of the former
// `bool BF16` (false = 0 = f32, true = 1 = bf16); 2 = IEEE half: the same matrix rate and footprint as bf16 with an 11-bit
// significand (8x less operand rounding), range +-65504 - activations behind eval BatchNorm and CMN'd features are O(1..100);
// a value beyond the range becomes inf and the embedding NaN (loud, never silently saturated).
// (the enum lives in asv_internal.h: the host runtime passes it to the launchers)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 asv_f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// f32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per
// instruction).  The shift/add formulation it replaces cost ~10 VALU operations per value - 1280 per lane in a
// GEMM epilogue, ~600 per k-step in the split-precision pooled GEMM.
typedef float asv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 asv_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const asv_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, asv_bf16x2));
}

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) { return pack_bf16x2(f, 0.0f) & 0xffffu; }

// two f32 -> one packed pair of 16-bit elements (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE) and back
template <int ET>
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
  static_assert(ET == ET_BF16 || ET == ET_F16, "16-bit element types only");
  const asv_f32x2 v = {lo, hi};
  if constexpr (ET == ET_BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, asv_bf16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, asv_f16x2));
}
template <int ET>
__device__ __forceinline__ void unpack_h16x2(uint32_t w, float &lo, float &hi) {
  static_assert(ET == ET_BF16 || ET == ET_F16, "16-bit element types only");
  if constexpr (ET == ET_BF16) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
  } else {
    const asv_f32x2 f = __builtin_convertvector(__builtin_bit_cast(asv_f16x2, w), asv_f32x2);
    lo = f.x; hi = f.y;
  }
}
template <int ET>
__device__ __forceinline__ float h16_bits_to_f32(uint32_t h) {       // the low 16 bits of h
  float lo, hi;
  unpack_h16x2<ET>(h, lo, hi);
  return lo;
}
template <int ET>
__device__ __forceinline__ uint32_t f32_to_h16_bits(float f) { return pack_h16x2<ET>(f, 0.0f) & 0xffffu; }

// 8 elements (one 16-byte piece) <-> 8 f32
template <int ET>
__device__ __forceinline__ void unpack_h16x8(const uint4 u, float *v) {
  unpack_h16x2<ET>(u.x, v[0], v[1]); unpack_h16x2<ET>(u.y, v[2], v[3]);
  unpack_h16x2<ET>(u.z, v[4], v[5]); unpack_h16x2<ET>(u.w, v[6], v[7]);
}
template <int ET>
__device__ __forceinline__ uint4 pack_h16x8(const float *v) {
  return make_uint4(pack_h16x2<ET>(v[0], v[1]), pack_h16x2<ET>(v[2], v[3]), pack_h16x2<ET>(v[4], v[5]), pack_h16x2<ET>(v[6], v[7]));
}

// the matrix instruction of a 16-bit element type: D = A (32 x 16) . B (16 x 32) + C, f32 accumulate
template <int ET>
__device__ __forceinline__ f32x16_t mfma16(const uint4 a, const uint4 b, const f32x16_t c) {
  static_assert(ET == ET_BF16 || ET == ET_F16, "16-bit element types only");
  if constexpr (ET == ET_BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// ReLU of a packed pair of 16-bit floats as ONE integer instruction (v_pk_max_i16 x, 0): a negative float of either format has
// its sign bit set = a negative int16, non-negative ones are unchanged (-0 -> +0; a NaN with the sign bit becomes 0)
__device__ __forceinline__ uint32_t relu_h16x2(uint32_t w) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}

// elementwise add of two 16-byte pieces holding 8 16-bit elements (f32 add, RNE back)
template <int ET>
__device__ __forceinline__ uint4 add_h16x8(uint4 a, uint4 b) {
  float va[8], vb[8];
  unpack_h16x8<ET>(a, va);
  unpack_h16x8<ET>(b, vb);
#pragma unroll
  for (int i = 0; i < 8; ++i) va[i] += vb[i];
  return pack_h16x8<ET>(va);
}
__device__ __forceinline__ uint4 add_bf16x8(uint4 a, uint4 b) { return add_h16x8<ET_BF16>(a, b); }

__device__ __forceinline__ uint4 add_f32x4(uint4 a, uint4 b) {
  uint4 r;
  r.x = __float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x));
  r.y = __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y));
  r.z = __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z));
  r.w = __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w));
  return r;
}

// ---- the operand split of the f32x precision mode (kernels_tdnn_x3.hip, kernels_tdnn_chainx.hip, kernels_conv2d_x3.hip)
struct X3Frag { uint4 hi, lo; };     // 8 k values of one lane: the two 16-bit halves

// Range watch of the IEEE-half split: a packed pair of hi halves -> a word whose bits 15 / 31 are set iff the low / high half is
// inf or NaN (0x7c00 + 0x0400 carries into bit 15; the largest finite half, 0x7bff, does not).  Callers OR these words together
// (3 VALU operations per pair) and publish once per wave with x3_publish_range: an activation beyond +-65504 would otherwise turn
// into a NaN product that the next ReLU (fmaxf) silently maps to 0 - wrong embeddings without a trace.
__device__ __forceinline__ uint32_t h16_range_bits(uint32_t packed_hi) { return (packed_hi & 0x7fff7fffu) + 0x04000400u; }
__device__ __forceinline__ void x3_publish_range(uint32_t acc_bits, uint32_t *status) {
  if (status != nullptr && __builtin_amdgcn_ballot_w64((acc_bits & 0x80008000u) != 0u) != 0ull && (threadIdx.x & 63) == 0) atomicOr(status, ASV_STATUS_HALF_RANGE);
}

// 8 f32 -> hi + lo in the 16-bit type ET (x - hi is exact in f32: hi keeps the leading 8 / 11 significand bits of x).
// ET_BF16: 16 significant bits together, the whole f32 exponent range.  ET_F16: 22 bits together for |x| >= 2^-2 (lo is a
// normal half there), an absolute error <= 2^-25 below (lo a subnormal half; the matrix cores keep subnormal inputs - checked
// on the device by tests/test_gpu_kernels.py), |x| <= 65504.  WITH_LO = false: one rounding, no second half (the measured
// two-instruction variants).
template <int ET, bool WITH_LO>
__device__ __forceinline__ X3Frag x3_split(const uint4 a, const uint4 b) {
  const float v[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                      __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
  uint32_t h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = pack_h16x2<ET>(v[2 * k], v[2 * k + 1]);
    l[k] = 0u;
    if constexpr (WITH_LO) {
      float r0, r1;
      if constexpr (ET == ET_F16) {
        // x - hi in ONE mixed-precision fma per value, straight from the packed halves (v_fma_mix_f32: hi is read as half, -1 and x
        // as f32): the unpacking conversions (two v_cvt_f32_f16 per pair) were a third of this loop's VALU work, and the VALU work -
        // ~4 operations per matrix instruction - is what this kernel's K loop is bound by
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h[k]), "v"(v[2 * k]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h[k]), "v"(v[2 * k + 1]));
      } else {
        float h0, h1;
        unpack_h16x2<ET>(h[k], h0, h1);
        r0 = v[2 * k] - h0; r1 = v[2 * k + 1] - h1;
      }
      l[k] = pack_h16x2<ET>(r0, r1);
    }
  }
  X3Frag f;
  f.hi = make_uint4(h[0], h[1], h[2], h[3]);
  f.lo = make_uint4(l[0], l[1], l[2], l[3]);
  return f;
}
// the same, watching the range of the half split: `range` collects h16_range_bits of the hi halves (ET_F16 only; bf16 halves
// share the f32 exponent range)
template <int ET, bool WITH_LO>
__device__ __forceinline__ X3Frag x3_split(const uint4 a, const uint4 b, uint32_t &range) {
  const X3Frag f = x3_split<ET, WITH_LO>(a, b);
  if constexpr (ET == ET_F16) range |= h16_range_bits(f.hi.x) | h16_range_bits(f.hi.y) | h16_range_bits(f.hi.z) | h16_range_bits(f.hi.w);
  return f;
}

// max(v, lo) as ONE instruction.  fmaxf() costs three on this target: IEEE mode makes the compiler quiet both operands
// (v_max_f32 x, x, x) in front of the real v_max_f32 (and it folds __builtin_amdgcn_fmed3f(v, lo, inf) back into the same) - 256
// extra VALU operations in a 128-register epilogue, next to a partner wave whose MFMA stream leaves a VALU operation ~14 cycles.
// lo = 0 (ReLU) or -inf (no activation); the accumulators are finite.
__device__ __forceinline__ float max_lo(float v, float lo) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(lo));
  return r;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ASV_ACT_RELU: return fmaxf(v, 0.0f);
    case ASV_ACT_TANH: return tanhf(v);
    case ASV_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <int ET>
__device__ __forceinline__ float load_elem(const void *base, size_t idx) {
  if constexpr (ET != ET_F32) return h16_bits_to_f32<ET>(reinterpret_cast<const uint16_t *>(base)[idx]);
  else return reinterpret_cast<const float *>(base)[idx];
}

template <int ET>
__device__ __forceinline__ void store_elem(void *base, size_t idx, float v) {
  if constexpr (ET != ET_F32) reinterpret_cast<uint16_t *>(base)[idx] = (uint16_t)f32_to_h16_bits<ET>(v);
  else reinterpret_cast<float *>(base)[idx] = v;
}

// The shared epilogue of one output element of a TDNN layer (asv_amd.h asv_tdnn_desc_t):
//   z = acc + bias (+ seg_bias);  z = affine_first ? act1(z*s+t) : act1(z)*s+t;
//   y = act2(z) (* seg_scale) (+ residual);  gap rows produce 0.
template <int RES_ET>
__device__ __forceinline__ float tdnn_epilogue(const TdnnKernelParams &p, float acc, int row, int ch,
                                               float bias, float scale, float shift, bool valid) {
  if (!valid) return 0.0f;
  float z = acc + bias;
  int seg = 0;
  if (p.seg_bias != nullptr || p.seg_scale != nullptr) seg = p.row_seg[row];
  if (p.seg_bias != nullptr) z += p.seg_bias[(size_t)seg * p.ld_segbias + ch];
  if (p.affine_first) {
    z = apply_act(z * scale + shift, p.act1);
  } else {
    z = apply_act(z, p.act1) * scale + shift;
  }
  z = apply_act(z, p.act2);
  if (p.seg_scale != nullptr) z *= p.seg_scale[(size_t)seg * p.ld_segscale + ch];
  if (p.res != nullptr) z += load_elem<RES_ET>(p.res, (size_t)row * p.ldres + ch);
  return z;
}

// Fast epilogue for the common layer shape (affine -> [ReLU] -> folded BN, no per-segment terms):
// branch-free, a handful of VALU ops per element.  `lo` = 0 for ReLU, -inf for no activation.
// The general form above (tanh / sigmoid / "bn-relu" order / per-segment bias and scale / residual)
// lives in separately instantiated kernels so its large code never sits in the hot kernels.
__device__ __forceinline__ float tdnn_epilogue_fast(float acc, float bias, float lo, float scale, float shift, bool valid) {
  const float z = fmaxf(acc + bias, lo) * scale + shift;
  return valid ? z : 0.0f;
}

// XCD-aware, bijective remap of a 1-D grid (cdna_hip_programming.md T1): hardware places
// block b on XCD b % 8; give each XCD a contiguous run of logical tiles so the N tiles that
// share an A panel hit the same L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
  const int xcd = bid & 7, q = nblocks >> 3, r = nblocks & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

}  // namespace asv
