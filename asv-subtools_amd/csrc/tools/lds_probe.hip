// LDS read-pattern probe: cycles per ds_read_b128 wave-instruction for several address patterns
// (developer tool).  8 waves per workgroup, one workgroup per CU, like the GEMM kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int PAT>
__device__ __forceinline__ int addr_of(int lane, int it) {
  const int lr = lane & 31, lh = lane >> 5;
  const int slot = (it & 3) * 2 + lh;
  const int row = lr + (it >> 2) * 32 + 4;
  switch (PAT) {
    case 0: return lane * 16 + it * 1024;                                   // lane-linear
    case 1: return row * 128 + ((slot ^ ((row >> 1) & 7)) * 16);            // GEMM swizzle (row>>1)
    case 2: return row * 128 + slot * 16;                                   // no swizzle
    case 3: return row * 128 + ((slot ^ (row & 7)) * 16);                   // swizzle (row&7)
    case 4: return row * 144 + slot * 16;                                   // padded rows (+16 B)
    case 5: return row * 128 + ((slot ^ ((row >> 1) & 7)) * 16) + 0;        // same as 1 (control)
    default: return 0;
  }
}

template <int PAT>
__global__ __launch_bounds__(512) void probe(uint32_t *out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[160 * 1024];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i;
  __syncthreads();
  uint4 acc = make_uint4(0, 0, 0, 0);
  const long long t0 = clock64();
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const uint4 v = *reinterpret_cast<const uint4 *>(lds + (addr_of<PAT>(lane, it) & (128 * 1024 - 16)));
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
  }
  const long long t1 = clock64();
  if (acc.x == 0x12345678u) out[1000] = acc.y ^ acc.z ^ acc.w;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[PAT] = (uint32_t)(t1 - t0);
}

int main() {
  uint32_t *d; CK(hipMalloc(&d, 4096 * 4)); CK(hipMemset(d, 0, 4096 * 4));
  const int iters = 200;
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, d, iters);
  hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, d, iters);
  hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, d, iters);
  hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, d, iters);
  hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, d, iters);
  CK(hipDeviceSynchronize());
  uint32_t h[8]; CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
  const char *names[] = {"lane-linear", "swizzle slot^((row>>1)&7)", "row*128 no swizzle", "swizzle slot^(row&7)", "row pitch 144 B"};
  for (int p = 0; p < 5; ++p)
    printf("%-28s %8.1f wave-clocks per ds_read_b128 (8 waves/CU issuing; divide by 8 for CU throughput)\n", names[p], (double)h[p] / (iters * 16));
  return 0;
}
