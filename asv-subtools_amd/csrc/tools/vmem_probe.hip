// Developer probe (not part of libasv_amd.so): what does a stream of 16-byte-per-lane global loads cost a CU that is running matrix
// instructions?  (Round 6: the chain kernels' K loops run at the matrix rate once HALF of their weight-fragment loads are taken away,
// whatever cache level serves them - profiles/r6z_chain_fetch_experiments.txt.  This probe takes the kernels out of the question.)
//
// One workgroup of 8 waves per CU (2 per SIMD, 128 accumulator registers per wave like tdnn_chain_kernel).  Per iteration a wave issues
// 16 x v_mfma_f32_32x32x16_bf16 on 8 accumulators and L global_load_dwordx4 (1 KiB per wave and load), the loaded registers feeding the
// NEXT iteration's matrix instructions as operands (so the loads are real dependencies, a step ahead, as in the kernels).  Source:
//   mode 0  a 64 KiB region per workgroup walked linearly (L1 / L2 hits after the first pass: the latency is short, the path is what is measured)
//   mode 1  a 48 MiB region shared by all workgroups walked linearly (the weight stream of the chain kernels: L2 hits + misses into the MALL)
// Output per L (loads per 16 matrix instructions of a wave), spacing S (a load behind every S-th instruction) and fetch distance D: time per iteration from HIP events, the share of the bf16
// matrix peak the chip reached, the load bandwidth over the chip, and the s_memtime count per iteration (NOT core cycles: a pure matrix
// loop at 97 % of the 2.5 PF peak counts 24.4 per instruction where the pipe needs 32 core cycles - the counter runs at ~0.76 of the core clock there).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools_vmem_probe tools/vmem_probe.hip && ./tools_vmem_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8b __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// S = spacing of the loads in matrix-instruction slots (1: behind instructions 0 .. L - 1; 16 / L: spread evenly over the 16)
// R = ds_read_b128 per 16 matrix instructions (one behind each of the first R; the chain kernels read their frame operands from LDS: 16 per 16)
template <int L, int D, int S, int R>
__global__ __launch_bounds__(512, 2) void probe(const unsigned char *src, size_t region, size_t wg_stride, int iters, float *out, unsigned long long *cyc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (R > 0) { for (int i = threadIdx.x; i < 65536 / 16; i += 512) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u); __syncthreads(); }
  uint4 xr[2] = {make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u)};
  const unsigned char *base = src + (size_t)blockIdx.x * wg_stride + (size_t)lane * 16;
  v16f acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  // D = fetch distance in half-iterations (16 matrix instructions of this wave each): the loads of half h feed the matrix instructions of half h + D
  uint4 w[D + 1][8];
#pragma unroll
  for (int d = 0; d <= D; ++d)
#pragma unroll
    for (int k = 0; k < 8; ++k) w[d][k] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  size_t off = (size_t)wave * 8 * 1024;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; it += D + 1) {
#pragma unroll
    for (int half = 0; half <= D; ++half) {
      uint4 (&cur)[8] = w[half];
      uint4 (&nxt)[8] = w[(half + D) % (D + 1)];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (q % S == 0 && q / S < L) {
          nxt[(q / S) & 7] = *reinterpret_cast<const uint4 *>(base + off + (size_t)(q / S) * 1024);
        }
        if (q < R) xr[q & 1] = *reinterpret_cast<const uint4 *>(lds + ((wave * 8192 + q * 1024 + lane * 16 + (int)(off >> 6)) & 65535 & ~15));
        const v8b a = __builtin_bit_cast(v8b, cur[q & 7]), b = __builtin_bit_cast(v8b, R > 0 ? xr[(q + 1) & 1] : cur[(q + 3) & 7]);
        acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 7], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      off += 64 * 1024;                                  // 8 waves x 8 KiB
      if (off + 64 * 1024 > region) off = (size_t)wave * 8 * 1024;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

// The same question for ONE wave per SIMD with 256 accumulator registers (a 128 x 128 wave tile: 4 + 4 operand fetches per 16 matrix
// instructions = 0.5 per instruction instead of 0.75): 4 waves per CU, 32 matrix instructions per wave and iteration on 16 accumulators,
// L loads (spread, two iterations ahead) and R LDS reads beside them.
template <int L, int R>
__global__ __launch_bounds__(256, 1) void probe4(const unsigned char *src, size_t region, int iters, float *out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  __syncthreads();
  const unsigned char *base = src + (size_t)lane * 16;
  v16f acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  uint4 w[3][8], xr[4];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int k = 0; k < 8; ++k) w[d][k] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
  for (int k = 0; k < 4; ++k) xr[k] = w[0][0];
  size_t off = (size_t)wave * 8 * 1024;
  constexpr int S = L > 0 ? 32 / L : 32;
#pragma unroll 1
  for (int it = 0; it < iters; it += 3) {
#pragma unroll
    for (int half = 0; half < 3; ++half) {
      uint4 (&cur)[8] = w[half];
      uint4 (&nxt)[8] = w[(half + 2) % 3];
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        if (L > 0 && q % S == 0 && q / S < L) nxt[(q / S) & 7] = *reinterpret_cast<const uint4 *>(base + off + (size_t)(q / S) * 1024);
        if (R > 0 && q % (32 / (R > 0 ? R : 1)) == 0) xr[(q >> 2) & 3] = *reinterpret_cast<const uint4 *>(lds + ((wave * 16384 + q * 512 + lane * 16 + (int)(off >> 6)) & 65535 & ~15));
        const v8b a = __builtin_bit_cast(v8b, cur[q & 7]), b = __builtin_bit_cast(v8b, R > 0 ? xr[((q >> 2) + 1) & 3] : cur[(q + 3) & 7]);
        acc[q & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      off += 32 * 1024;
      if (off + 32 * 1024 > region) off = (size_t)wave * 8 * 1024;
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int L, int R>
static void run4(const unsigned char *src, size_t region, int wgs, int iters, float *out) {
  hipLaunchKernelGGL((probe4<L, R>), dim3(wgs), dim3(256), 0, 0, src, region, 66, out);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe4<L, R>), dim3(wgs), dim3(256), 0, 0, src, region, iters, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_it = (double)ms * 1e6 / iters;
  printf("one wave per SIMD, 256 accumulators: %d loads + %d LDS reads per 32 matrix instructions (%.2f fetches per instruction): %6.1f ns per iteration = %5.1f %% of the bf16 matrix peak\n",
         L, R, (L + R) / 32.0, ns_it, 100.0 * (double)wgs * 4 * 32 * 32768.0 / (ns_it * 1e-9) / 2.5e15);
}

// Calibration of s_memtime: wave 0 of every workgroup sleeps (s_sleep 16 = 16 x 64 sequencer clocks, 64 times) between two stamps while the
// other 7 waves either idle (LOAD = 0) or stream matrix instructions (LOAD = 1); s_memrealtime (100 MHz) brackets the same region.
template <int LOAD>
__global__ __launch_bounds__(512, 2) void calib(unsigned long long *res, float *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < 64; ++i) __builtin_amdgcn_s_sleep(16);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { res[blockIdx.x * 2] = t1 - t0; res[blockIdx.x * 2 + 1] = r1 - r0; }
  } else if (LOAD) {
    v16f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const uint4 c = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const v8b a = __builtin_bit_cast(v8b, c);
#pragma unroll 1
    for (int it = 0; it < 3000; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc[q & 3], 0, 0, 0);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
  }
}

static int cus_g = 256;
template <int L, int D = 1, int S = 1, int R = 0>
static void run(const unsigned char *src, size_t region, size_t wg_stride, int wgs, int iters, float *out, unsigned long long *cyc, const char *what) {
  CK(hipMemset(cyc, 0, (size_t)wgs * 8 * 8));
  hipLaunchKernelGGL((probe<L, D, S, R>), dim3(wgs), dim3(512), 0, 0, src, region, wg_stride, 66, out, cyc);        // warm
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<L, D, S, R>), dim3(wgs), dim3(512), 0, 0, src, region, wg_stride, iters, out, cyc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)wgs * 8);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (auto v : h) sum += (double)v;
  const double per_it = sum / (double)h.size() / iters;
  // per CU and iteration (= one step of both waves of every SIMD): 8 waves x L KiB
  const double ns_it = (double)ms * 1e6 / iters;
  printf("%-28s L = %d loads + %d LDS reads per 16 matrix instructions, a load every %d, fetched %d x 16 instructions ahead: %6.1f ns per iteration = %5.1f %% of the bf16 matrix peak (2.5 PF), %5.2f TB/s over the chip; s_memtime %7.1f per iteration\n",
         what, L, R, S, D, ns_it, 100.0 * (double)cus_g * 4 * 32 * 32768.0 / (ns_it * 1e-9) / 2.5e15, (double)cus_g * 8.0 * L * 1024.0 / (ns_it * 1e-9) / 1e12, per_it);
}

int main() {
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  cus_g = cus;
  const size_t shared_region = 48u << 20, priv = 64u << 10;
  unsigned char *src = nullptr;
  float *out = nullptr;
  unsigned long long *cyc = nullptr;
  const size_t bytes = std::max(shared_region, (size_t)cus * priv) + (64u << 10);
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 0x3f, bytes));
  CK(hipMalloc(&out, (size_t)cus * 512 * 4));
  CK(hipMalloc(&cyc, (size_t)cus * 8 * 8));
  for (int load = 0; load < 2; ++load) {
    unsigned long long *res = nullptr;
    CK(hipMalloc(&res, (size_t)cus * 16));
    for (int rep = 0; rep < 2; ++rep) {
      if (load) hipLaunchKernelGGL(calib<1>, dim3(cus), dim3(512), 0, 0, res, out); else hipLaunchKernelGGL(calib<0>, dim3(cus), dim3(512), 0, 0, res, out);
      CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)cus * 2);
    CK(hipMemcpy(h.data(), res, h.size() * 8, hipMemcpyDeviceToHost));
    double t = 0, r = 0;
    for (int i = 0; i < cus; ++i) { t += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    printf("calibration, %s: 64 x s_sleep 16 (= 65536 sequencer clocks + loop overhead) = %.0f s_memtime ticks = %.2f us of s_memrealtime -> s_memtime runs at %.0f MHz, the sequencer at >= %.0f MHz\n",
           load ? "7 waves of matrix instructions beside" : "idle CU", t / cus, r / cus / 100.0, 100.0 * t / r, 100.0 * 65536.0 * cus / r);
    CK(hipFree(res));
  }
  for (int round = 0; round < 2; ++round) {
    run4<0, 0>(src, shared_region, cus, 2049, out);
    run4<8, 0>(src, shared_region, cus, 2049, out);
    run4<0, 8>(src, shared_region, cus, 2049, out);
    run4<8, 8>(src, shared_region, cus, 2049, out);
    run4<4, 16>(src, shared_region, cus, 2049, out);
    run4<8, 16>(src, shared_region, cus, 2049, out);
  }
  const int iters = 4098;                                // a multiple of 2 and of 3
  for (int round = 0; round < 2; ++round) {
    run<0>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<1>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<2>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<4>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<6>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<8>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<2, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 2>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<8, 2>(src, priv, priv, cus, iters, out, cyc, "private 64 KiB (cache hits)");
    run<2, 1, 8>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<2, 1, 4>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<2, 1, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 1, 4>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 1, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 2, 4>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8, 1, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8, 2, 2>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<0, 1, 1, 8>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<0, 1, 1, 16>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 2, 4, 8>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<4, 2, 4, 16>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8, 2, 2, 16>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
    run<8, 1, 1, 16>(src, shared_region, 0, cus, iters, out, cyc, "shared 48 MiB stream");
  }
  return 0;
}
