// Does a wave alone on its SIMD issue v_mfma_f32_32x32x16_bf16 every 32 cycles when its accumulators live in AGPRs?
// NACC accumulator tiles (16 registers each) per wave, 4 waves per workgroup (one per SIMD), one workgroup per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
template <int NACC>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *out, float *sinkp, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16_t acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 a = make_uint4(lane, 1, 2, 3), b = make_uint4(3, lane, 1, 0);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sinkp[0] = s;
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NACC> void run(unsigned long long *out, float *sinkp) {
  const int iters = 1000;
  hipLaunchKernelGGL((probe<NACC>), dim3(256), dim3(256), 0, 0, out, sinkp, iters);
  hipLaunchKernelGGL((probe<NACC>), dim3(256), dim3(256), 0, 0, out, sinkp, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), out, 8192, hipMemcpyDeviceToHost);
  double m = 0; for (auto v : h) m += (double)v; m /= h.size();
  printf("  %2d accumulator tiles (%3d registers): %.1f cycles per MFMA\n", NACC, NACC * 16, m / (iters * (double)NACC));
}
int main() {
  unsigned long long *out; float *sinkp;
  hipMalloc(&out, 8192); hipMalloc(&sinkp, 4);
  run<4>(out, sinkp); run<8>(out, sinkp); run<12>(out, sinkp); run<16>(out, sinkp);
  return 0;
}
