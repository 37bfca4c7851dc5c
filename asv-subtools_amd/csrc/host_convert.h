// Host-side conversions f32 <-> bf16 / IEEE half used when weights are packed (plain C++: also compiled by
// tests/test_capi_and_plan.py with g++ and checked against numpy's float16 / a bit-level bf16 reference).
#pragma once
#include <stdint.h>
#include <string.h>

namespace asv {

inline uint16_t f32_to_bf16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

inline float bf16_to_f32_host(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// f32 -> IEEE half, round-to-nearest-even, subnormals and overflow to infinity as the hardware conversion does
// (v_cvt_pk_f16_f32); and back
inline uint16_t f32_to_f16_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                     // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                    // >= 65520 rounds to infinity
  if (a < 0x33000001u) return (uint16_t)sign;                                 // <= 2^-25: rounds to zero (2^-25 itself ties to even = 0)
  const int e = (int)(a >> 23) - 127;                                         // unbiased exponent
  uint32_t man = (a & 0x7fffffu) | 0x800000u;                                 // 24-bit significand
  int shift = (e < -14) ? (13 + (-14 - e)) : 13;                              // bits dropped (subnormal results drop more)
  uint32_t half = man >> shift;
  const uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (half & 1u))) ++half;
  // half now holds the significand with the implicit bit at position 10 (normal) or below it (subnormal); adding the
  // exponent field lets a carry out of the significand bump the exponent (and reach infinity) by itself
  const uint32_t ebits = (e < -14) ? 0u : (uint32_t)(e + 15 - 1) << 10;
  return (uint16_t)(sign | (ebits + half));
}

inline float f16_to_f32_host(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const int e = (h >> 10) & 0x1f;
  const uint32_t m = h & 0x3ffu;
  float f;
  if (e == 0) {
    f = (float)m * (1.0f / 16777216.0f);                                      // m * 2^-24
    uint32_t u;
    memcpy(&u, &f, 4);
    u |= sign;
    memcpy(&f, &u, 4);
    return f;
  }
  const uint32_t u = sign | (e == 31 ? (0x7f800000u | (m << 13)) : ((uint32_t)(e - 15 + 127) << 23 | (m << 13)));
  memcpy(&f, &u, 4);
  return f;
}

// f32 -> OCP e4m3 (fn: no infinities, 0x7f / 0xff = NaN), round-to-nearest-even, saturating at +-448 (what the weights of the f32m
// form are packed with: pack_tdnn_weight_mx8); and back
inline uint8_t f32_to_e4m3_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint8_t)(sign | 0x7fu);                        // NaN
  float m;
  memcpy(&m, &a, 4);                                                           // |f|
  if (m >= 448.0f) return (uint8_t)(sign | 0x7eu);                            // saturate (incl. infinity)
  if (m < 0.0009765625f) return sign;                                         // < 2^-10: half of the smallest subnormal 2^-9 ties to even = 0
  const int e = (int)(a >> 23) - 127;                                         // unbiased exponent
  const uint32_t man = (a & 0x7fffffu) | 0x800000u;                           // 24-bit significand
  const int shift = (e < -6) ? (20 + (-6 - e)) : 20;                          // keep 3 fraction bits (subnormals: fewer)
  uint32_t q = man >> shift;
  const uint32_t rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
  if (rem > mid || (rem == mid && (q & 1u))) ++q;
  const uint32_t ebits = (e < -6) ? 0u : (uint32_t)(e + 7 - 1) << 3;           // the implicit bit of q bumps the exponent field by one
  const uint32_t r = ebits + q;
  return (uint8_t)(sign | (r > 0x7eu ? 0x7eu : r));
}

inline float e4m3_to_f32_host(uint8_t b) {
  const int e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) { const uint32_t n = 0x7fc00000u; memcpy(&v, &n, 4); return v; }
  if (e == 0) v = (float)m * 0.001953125f;                                    // m * 2^-9
  else {
    const uint32_t u = ((uint32_t)(e - 7 + 127) << 23) | ((uint32_t)m << 20);
    memcpy(&v, &u, 4);
  }
  return (b & 0x80u) ? -v : v;
}

}  // namespace asv
