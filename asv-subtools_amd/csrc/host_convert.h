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

}  // namespace asv
