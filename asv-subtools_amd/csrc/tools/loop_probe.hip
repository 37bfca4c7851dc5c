// Issue-rate probe for the GEMM inner loop (developer tool): one workgroup per CU, each wave runs
// ITER groups of {8 MFMA 32x32x16 bf16 + NDS ds_read_b128 (+ address VALU) + NVM global_load_dwordx4}
// with the operands prefetched one group ahead, and reports shader cycles per group.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NDS, int NVM, bool VALU_ADDR, int WAVES, int IL>
__global__ __launch_bounds__(WAVES * 64) void probe(const uint4 *g, unsigned long long *out, float *sinkp, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 65536 / 16; i += WAVES * 64) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(i, i, i, i);
  __syncthreads();
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 xa[4], xb[4], wa[2], wb[2];
  for (int i = 0; i < 4; ++i) xa[i] = xb[i] = make_uint4(lane, i, 1, 2);
  for (int i = 0; i < 2; ++i) wa[i] = wb[i] = make_uint4(lane, i, 3, 4);
  const int lr = lane & 31, lh = lane >> 5;
  const uint4 *gp = g + (size_t)(blockIdx.x * WAVES + wave) * 64 * 64 + lane;
  int base[4];
  for (int i = 0; i < 4; ++i) { const int w = i * 32 + lr; base[i] = w * 128 + ((lh ^ ((w >> 1) & 7)) * 16); }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // "group": reads for the next group, then 8 MFMAs on the current operands
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 *xn = half ? xa : xb;
      uint4 *xc = half ? xb : xa;
      uint4 *wn = half ? wa : wb;
      uint4 *wc = half ? wb : wa;
#pragma unroll
      for (int i = 0; i < NDS; ++i) {
        int a;
        if (VALU_ADDR) { const int w = i * 32 + lr + (it & 3); a = w * 128 + (((half * 2 + lh) ^ ((w >> 1) & 7)) * 16); }
        else a = base[i] + half * 32;
        xn[i] = *reinterpret_cast<const uint4 *>(lds + a);
      }
      if (IL == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wc[j]), __builtin_bit_cast(bf16x8_t, xc[i]), acc[i * 2 + j], 0, 0, 0);
      if (IL == 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NVM; ++j) wn[j] = gp[(size_t)((it * 2 + half) & 31) * 64 + j * 2048];
      if (IL == 1) {
        // one LDS read (+ its address VALU) behind every second MFMA, the fragment fetches behind the last ones
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (VALU_ADDR) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          if (NDS > 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (NVM > 0 && k >= 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sinkp[0] = s;
  if (lane == 0) out[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int NDS, int NVM, bool VALU_ADDR, int WAVES, int IL = 0>
int run(const char *name, const uint4 *g, unsigned long long *out, float *sinkp) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((probe<NDS, NVM, VALU_ADDR, WAVES, IL>), dim3(blocks), dim3(WAVES * 64), 0, 0, g, out, sinkp, iters);
  hipLaunchKernelGGL((probe<NDS, NVM, VALU_ADDR, WAVES, IL>), dim3(blocks), dim3(WAVES * 64), 0, 0, g, out, sinkp, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * WAVES);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  double m = 0; for (auto v : h) m += (double)v; m /= h.size();
  printf("  %-46s %7.1f cycles per 8-MFMA group (256 = matrix pipe bound for one wave per SIMD)\n", name, m / (iters * 2.0));
  return 0;
}

int main() {
  uint4 *g; unsigned long long *out; float *sinkp;
  CK(hipMalloc(&g, (size_t)256 * 8 * 64 * 64 * 16 + (1 << 20))); CK(hipMemset(g, 1, (size_t)256 * 8 * 64 * 64 * 16 + (1 << 20)));
  CK(hipMalloc(&out, 256 * 8 * 8)); CK(hipMalloc(&sinkp, 4));
  printf("4 waves per CU (one per SIMD):\n");
  run<0, 0, false, 4>("8 MFMA", g, out, sinkp);
  run<4, 0, false, 4>("8 MFMA + 4 ds_read_b128 (fixed addresses)", g, out, sinkp);
  run<4, 0, true, 4>("8 MFMA + 4 ds_read_b128 + address VALU", g, out, sinkp);
  run<0, 2, false, 4>("8 MFMA + 2 global_load_dwordx4 (L2)", g, out, sinkp);
  run<4, 2, true, 4>("8 MFMA + 4 ds_read + VALU + 2 global_load", g, out, sinkp);
  run<4, 0, false, 4, 1>("interleaved: 8 MFMA + 4 ds_read (fixed addr)", g, out, sinkp);
  run<4, 0, true, 4, 1>("interleaved: 8 MFMA + 4 ds_read + VALU", g, out, sinkp);
  run<4, 2, true, 4, 1>("interleaved: 8 MFMA + 4 ds_read + VALU + 2 gload", g, out, sinkp);
  printf("8 waves per CU (two per SIMD), per wave:\n");
  run<0, 0, false, 8>("8 MFMA", g, out, sinkp);
  run<4, 0, true, 8>("8 MFMA + 4 ds_read_b128 + address VALU", g, out, sinkp);
  run<4, 2, true, 8>("8 MFMA + 4 ds_read + VALU + 2 global_load", g, out, sinkp);
  run<4, 0, true, 8, 1>("interleaved: 8 MFMA + 4 ds_read + VALU", g, out, sinkp);
  run<4, 2, true, 8, 1>("interleaved: 8 MFMA + 4 ds_read + VALU + 2 gload", g, out, sinkp);
  return 0;
}
