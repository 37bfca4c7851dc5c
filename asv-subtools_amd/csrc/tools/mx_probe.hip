// Developer probe (not part of libasv_amd.so): can the two correction products of the f32x mode run on the block-scaled fp8 matrix
// instruction of gfx950?
//
//   f32x today, per 16 channels and fragment pair:  w_hi x_hi + w_hi x_lo + w_lo x_hi          = 3 x v_mfma_f32_32x32x16_f16
//   candidate, per 32 channels and fragment pair:   2 x v_mfma_f32_32x32x16_f16 (w_hi x_hi)
//                                                 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4 with K = [w_hi8 . x_lo8 | w_lo8 . x_hi8]
//
// Part 1 pins the operand layout of the scaled instruction (which lane / register / byte holds A[i][k], B[k][j], which lanes' scale
// bytes apply to which 32-deep K block) against a host evaluation - several hypotheses, the matching one is printed.
// Part 2 measures the issue rate of the three instruction mixes on every CU (8 waves per CU, 2 x 2 accumulators per wave like
// tdnn_chainx_kernel, random operands), interleaved rounds in one process.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools_mx_probe tools/mx_probe.hip && ./tools_mx_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float e4m3_decode(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 0) v = std::ldexp((float)m, -9);               // subnormal: m * 2^-3 * 2^-6
  else v = std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -v : v;
}

__global__ void one_mfma(const v8i *a, const v8i *b, const int *sa, const int *sb, v16f *d) {
  v16f acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  d[threadIdx.x] = acc;
}

// VAR 0: 24 f16 instructions per 32 channels (f32x today); 1: 8 f16 + 4 scaled fp8; 2: 8 f16 only (the f16 mode's matrix work);
//     3: 8 f16 + 4 scaled fp6 (format 2: e2m3, 6 operand registers)
template <int VAR>
__global__ __launch_bounds__(512, 2) void mix_kernel(const v8h *h, const v8i *q, float *out, int iters) {
  const int t = threadIdx.x;
  v8h wh[2], wl[2], xh[2], xl[2];
  v8i w8[2], x8[2];
  for (int i = 0; i < 2; ++i) {
    wh[i] = h[(t * 8 + i) & 4095]; wl[i] = h[(t * 8 + 2 + i) & 4095]; xh[i] = h[(t * 8 + 4 + i) & 4095]; xl[i] = h[(t * 8 + 6 + i) & 4095];
    w8[i] = q[(t * 4 + i) & 4095]; x8[i] = q[(t * 4 + 2 + i) & 4095];
  }
  v16f acc[2][2] = {};
  const int sc = 0x7f7f7f7f;                                      // E8M0 127 = 1.0
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
          if constexpr (VAR == 0) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
          }
        }
    }
    if constexpr (VAR == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[j], x8[i], acc[i][j], 0, 0, 0, sc, 0, sc);
    }
    if constexpr (VAR == 3) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v8i a6 = w8[j], b6 = x8[i];
          a6[6] = 0; a6[7] = 0; b6[6] = 0; b6[7] = 0;
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a6, b6, acc[i][j], 2, 2, 0, sc, 0, sc);
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + t] = s;
}

int main() {
  // ---- part 1: layout ----------------------------------------------------------------------------------------------------------
  std::vector<uint8_t> A(32 * 64), B(64 * 32);                    // A[i][k], B[k][j] as e4m3 bytes
  std::vector<uint8_t> SA(32 * 2), SB(32 * 2);                    // scale bytes per (row | column, K block of 32)
  srand(7);
  auto rnd8 = [] { uint8_t b; do { b = (uint8_t)(rand() & 0xff); } while ((b & 0x7f) == 0x7f || ((b >> 3) & 15) > 9 || ((b >> 3) & 15) < 5); return b; };
  for (auto &v : A) v = rnd8();
  for (auto &v : B) v = rnd8();
  for (auto &v : SA) v = (uint8_t)(124 + rand() % 7);
  for (auto &v : SB) v = (uint8_t)(124 + rand() % 7);
  std::vector<double> want(32 * 32, 0.0);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j)
      for (int k = 0; k < 64; ++k)
        want[i * 32 + j] += (double)e4m3_decode(A[i * 64 + k]) * std::ldexp(1.0, SA[i * 2 + k / 32] - 127) * (double)e4m3_decode(B[k * 32 + j]) * std::ldexp(1.0, SB[j * 2 + k / 32] - 127);
  // hypothesis: lane l holds row / column l & 31 and the K block l >> 5: register v, byte b = k 32 (l >> 5) + 4 v + b; its scale byte 0
  // applies to its own (row | column, block)
  std::vector<int> ha(64 * 8), hb(64 * 8), hsa(64), hsb(64);
  for (int l = 0; l < 64; ++l) {
    for (int v = 0; v < 8; ++v) {
      uint32_t wa = 0, wb = 0;
      for (int b = 0; b < 4; ++b) {
        const int k = 32 * (l >> 5) + 4 * v + b;
        wa |= (uint32_t)A[(l & 31) * 64 + k] << (8 * b);
        wb |= (uint32_t)B[k * 32 + (l & 31)] << (8 * b);
      }
      ha[l * 8 + v] = (int)wa; hb[l * 8 + v] = (int)wb;
    }
    hsa[l] = SA[(l & 31) * 2 + (l >> 5)]; hsb[l] = SB[(l & 31) * 2 + (l >> 5)];
  }
  int *da, *db, *dsa, *dsb; float *dd;
  CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 64));
  CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
  one_mfma<<<1, 64>>>((const v8i *)da, (const v8i *)db, dsa, dsb, (v16f *)dd);
  CK(hipDeviceSynchronize());
  std::vector<float> got(64 * 16);
  CK(hipMemcpy(got.data(), dd, 64 * 64, hipMemcpyDeviceToHost));
  double worst = 0.0, worst_t = 0.0, scale = 0.0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      worst = std::fmax(worst, std::fabs(got[l * 16 + r] - want[row * 32 + col]));
      worst_t = std::fmax(worst_t, std::fabs(got[l * 16 + r] - want[col * 32 + row]));
      scale = std::fmax(scale, std::fabs(want[row * 32 + col]));
    }
  printf("layout: max |D - host| = %.3g (transposed reading: %.3g), max |host| = %.3g  -> %s\n", worst, worst_t, scale,
         worst < 1e-5 * scale ? "A = rows, B = columns, D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31], k = 32 (l >> 5) + 4 v + b, scale byte per lane: CONFIRMED"
                              : "HYPOTHESIS WRONG");
  // ---- part 2: rates -----------------------------------------------------------------------------------------------------------
  int dev = 0, cus = 256;
  CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  std::vector<_Float16> hh(4096 * 8);
  for (auto &v : hh) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
  std::vector<uint8_t> hq(4096 * 32);
  for (auto &v : hq) v = rnd8();
  void *dh, *dq; float *dout;
  CK(hipMalloc(&dh, hh.size() * 2)); CK(hipMalloc(&dq, hq.size())); CK(hipMalloc(&dout, (size_t)cus * 512 * 4));
  CK(hipMemcpy(dh, hh.data(), hh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, hq.data(), hq.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 40000;
  const char *names[4] = {"f32x today: 24 f16 instructions / 32 channels", "8 f16 + 4 scaled fp8 (K = 64)", "8 f16 only (f16 mode)", "8 f16 + 4 scaled fp6 e2m3"};
  for (int round = 0; round < 4; ++round)
    for (int var = 0; var < 4; ++var) {
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 3; ++rep) {
        if (var == 0) mix_kernel<0><<<cus, 512>>>((const v8h *)dh, (const v8i *)dq, dout, iters);
        if (var == 1) mix_kernel<1><<<cus, 512>>>((const v8h *)dh, (const v8i *)dq, dout, iters);
        if (var == 2) mix_kernel<2><<<cus, 512>>>((const v8h *)dh, (const v8i *)dq, dout, iters);
        if (var == 3) mix_kernel<3><<<cus, 512>>>((const v8h *)dh, (const v8i *)dq, dout, iters);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
      // useful work of one iteration of one wave: 2 x 2 fragment pairs x 32 channels = 4 x 32 x 32 x 32 MACs
      const double macs = 3.0 * (double)cus * 8 * iters * 4.0 * 32 * 32 * 32;
      printf("round %d  %-48s %8.2f ms   %7.1f useful TFLOP/s (f32-grade products)\n", round, names[var], ms, 2.0 * macs / (ms * 1e-3) / 1e12);
    }
  return 0;
}
