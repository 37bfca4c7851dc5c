// Developer tool: do VALU operations of one wave run in the shadow of another wave's MFMA stream on the same SIMD?
// One workgroup of 8 waves per CU = two waves per SIMD (waves w and w + 4 share SIMD w on this launch order).  Modes:
//   0: waves 0-3 issue N independent v_mfma_f32_32x32x16_bf16 (8 accumulators), waves 4-7 idle
//   1: waves 4-7 issue M independent v_fma_f32 (16 chains), waves 0-3 idle
//   2: both at once
//   3: waves 0-3: MFMA stream; waves 4-7: MFMA stream too (two MFMA waves per SIMD)
//   4: like 2 with packed VALU operations (v_pk_fma_f32)
// Prints shader cycles (s_memtime) per MFMA and per VALU operation for each mode: mode 2 = max(mode 0, mode 1) means
// free co-issue, mode 2 = sum means the two streams serialise.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void probe(int mode, int n_mfma_groups, int n_valu_groups, unsigned long long *out, float *sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  //   5: like 2, the VALU waves at s_setprio 3;   6: like 2 with the roles swapped (waves 0-3 VALU, waves 4-7 MFMA)
  //   7: like 2, the MFMA waves leave a gap (s_nop 7 x 2) behind every MFMA;   8: like 5 + 6 (older VALU waves with priority)
  const bool swap = (mode == 6 || mode == 8);
  const bool lo = wave < 4, m2 = (mode == 2 || mode == 4 || mode == 5 || mode == 7);
  const bool mfma_wave = swap ? !lo : (lo ? (mode == 0 || m2 || mode == 3) : (mode == 3));
  const bool valu_wave = swap ? lo : (!lo && (mode == 1 || m2));
  if (valu_wave && (mode == 5 || mode == 8)) __builtin_amdgcn_s_setprio(3);
  const bool gap = mode == 7;
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 a = make_uint4(lane, 1, 2, 3), b = make_uint4(3, lane, 1, 0);
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * (lane + i);
  f32x2 pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = f32x2{v[2 * i], v[2 * i + 1]};
  const float m = 1.000001f, c = 1e-7f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mfma_wave) {
    if (!gap) {
      for (int g = 0; g < n_mfma_groups; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
      }
    } else {
      for (int g = 0; g < n_mfma_groups; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
          asm volatile("s_nop 7\n\ts_nop 7");
        }
      }
    }
  }
  if (valu_wave) {
    if (mode == 4) {
      const f32x2 m2 = {m, m}, c2 = {c, c};
      for (int g = 0; g < n_valu_groups; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = __builtin_elementwise_fma(pv[i], m2, c2);
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = __builtin_elementwise_fma(pv[i], m2, c2);
      }
    } else {
      for (int g = 0; g < n_valu_groups; ++g) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], m, c);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += pv[i].x + pv[i].y;
  if (s == 12345.678f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = ((t1 - t0) << 8) | (__builtin_amdgcn_s_getreg(2308) & 3);    // low byte: SIMD id (HW_ID[5:4])
}

int main() {
  const int blocks = 256, n_mfma_groups = 2000, n_valu_groups = 2000;       // 16 000 MFMAs | 32 000 VALU operations per wave
  unsigned long long *out; float *sink;
  CK(hipMalloc(&out, blocks * 8 * 8)); CK(hipMalloc(&sink, 4));
  std::vector<unsigned long long> h(blocks * 8);
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 9; ++mode) {
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), 0, 0, mode, n_mfma_groups, n_valu_groups, out, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      double mf = 0, va = 0;
      const bool sw = (mode == 6 || mode == 8);        // roles swapped: waves 4-7 are the MFMA waves
      for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) (sw ? va : mf) += (double)(h[b * 8 + w] >> 8); for (int w = 4; w < 8; ++w) (sw ? mf : va) += (double)(h[b * 8 + w] >> 8); }
      if (rep == 1 && mode == 0) { printf("SIMD of waves 0..7 (workgroup 0):"); for (int w = 0; w < 8; ++w) printf(" %d", (int)(h[w] & 3)); printf("\n"); }
      mf /= blocks * 4; va /= blocks * 4;
      if (rep == 1)
        printf("mode %d: MFMA waves %.0f cycles (%.2f per MFMA), VALU waves %.0f cycles (%.2f per VALU op%s)\n", mode, mf, mf / (n_mfma_groups * 8.0),
               va, va / (n_valu_groups * 16.0), mode == 3 ? " - MFMA here: per MFMA x2" : (mode == 4 ? ", packed: 2 fma each" : ""));
    }
  return 0;
}
