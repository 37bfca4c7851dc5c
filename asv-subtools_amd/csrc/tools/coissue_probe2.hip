// Developer tool, second part of coissue_probe.hip: WHICH instruction streams share a SIMD with a saturating MFMA stream?
// coissue_probe.hip showed (profiles/r2o_coissue.txt) that the VALU operations of one wave do not issue at all while the other
// wave of the SIMD streams independent v_mfma_f32_32x32x16_bf16.  This probe asks the follow-up questions a kernel design needs:
//   * VALU operations interleaved INSIDE the MFMA wave (k per MFMA): free, or 4 cycles each on top of the 32?
//   * the 16x16x32 shape instead of 32x32x16: same behaviour?
//   * LDS reads / LDS float atomics of the partner wave: do they issue beside the MFMA stream?
//   * an MFMA stream that carries one ds_read_b128 per MFMA (a real main loop): does the partner's VALU get slots then?
// One workgroup of 8 waves per CU; waves w and w + 4 share a SIMD.  Waves 0-3 take role A, waves 4-7 role B.
// Everything timed is inline assembly, so the instruction order is the one written here.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { A_IDLE = 0, A_MFMA32, A_MFMA16, A_MFMA32_DS, A_MFMA32_V1, A_MFMA32_V2, A_MFMA32_V4, A_MFMA32_V6, A_MFMA32_PK2, A_MFMA32_NOP0 };
enum { B_IDLE = 0, B_VALU, B_DSREAD, B_DSADD, B_MFMA32, B_MFMA32_V4, B_VPK };

#define MFMA32(i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define MFMA16(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a), "v"(b))
#define VFMA(j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(m), "v"(c))
#define VPKFMA(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[j]) : "v"(m2), "v"(c2))

__global__ __launch_bounds__(512) void probe2(int role_a, int role_b, int n_groups, unsigned long long *out, float *sink) {
  extern __shared__ unsigned char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool lo = wave < 4;
  const int role = lo ? role_a : role_b;
  f32x16_t acc[8];
  f32x4_t acc4[8];
  for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
  u32x4_t a = {(unsigned)lane, 1u, 2u, 3u}, b = {3u, (unsigned)lane, 1u, 0u};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 1e-3f * (lane + i);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pv[8];
  for (int i = 0; i < 8; ++i) pv[i] = f32x2{v[2 * i], v[2 * i + 1]};
  const float m = 1.000001f, c = 1e-7f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  u32x4_t rd[8];
  for (int i = 0; i < 8; ++i) rd[i] = u32x4_t{0u, 0u, 0u, 0u};
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float *>(smem)[i] = 0.f;
  const unsigned lds_addr = (unsigned)(wave * 8192 + lane * 16);           // a private 8 KiB region per wave, conflict-free
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (lo) {
    switch (role) {
    case A_MFMA32:
      for (int g = 0; g < n_groups; ++g) { MFMA32(0); MFMA32(1); MFMA32(2); MFMA32(3); MFMA32(4); MFMA32(5); MFMA32(6); MFMA32(7); }
      break;
    case A_MFMA16:                                                          // twice as many, half the work each
      for (int g = 0; g < 2 * n_groups; ++g) { MFMA16(0); MFMA16(1); MFMA16(2); MFMA16(3); MFMA16(4); MFMA16(5); MFMA16(6); MFMA16(7); }
      break;
    case A_MFMA32_DS:
#define MD(i) MFMA32(i); asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[i]) : "v"(lds_addr), "n"((i) * 1024) : "memory")
      for (int g = 0; g < n_groups; ++g) { MD(0); MD(1); MD(2); MD(3); MD(4); MD(5); MD(6); MD(7); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      break;
    case A_MFMA32_V1:
#define M1(i) MFMA32(i); VFMA(i)
      for (int g = 0; g < n_groups; ++g) { M1(0); M1(1); M1(2); M1(3); M1(4); M1(5); M1(6); M1(7); }
      break;
    case A_MFMA32_V2:
#define M2(i) MFMA32(i); VFMA(2 * (i)); VFMA(2 * (i) + 1)
      for (int g = 0; g < n_groups; ++g) { M2(0); M2(1); M2(2); M2(3); M2(4); M2(5); M2(6); M2(7); }
      break;
    case A_MFMA32_V4:
#define M4(i) MFMA32(i); VFMA((4 * (i)) & 15); VFMA((4 * (i) + 1) & 15); VFMA((4 * (i) + 2) & 15); VFMA((4 * (i) + 3) & 15)
      for (int g = 0; g < n_groups; ++g) { M4(0); M4(1); M4(2); M4(3); M4(4); M4(5); M4(6); M4(7); }
      break;
    case A_MFMA32_V6:
#define M6(i) MFMA32(i); VFMA((6 * (i)) & 15); VFMA((6 * (i) + 1) & 15); VFMA((6 * (i) + 2) & 15); VFMA((6 * (i) + 3) & 15); VFMA((6 * (i) + 4) & 15); VFMA((6 * (i) + 5) & 15)
      for (int g = 0; g < n_groups; ++g) { M6(0); M6(1); M6(2); M6(3); M6(4); M6(5); M6(6); M6(7); }
      break;
    case A_MFMA32_PK2:
#define MP(i) MFMA32(i); VPKFMA((i)); VPKFMA(((i) + 4) & 7)
      for (int g = 0; g < n_groups; ++g) { MP(0); MP(1); MP(2); MP(3); MP(4); MP(5); MP(6); MP(7); }
      break;
    case A_MFMA32_NOP0:
#define MN(i) MFMA32(i); asm volatile("s_nop 0")
      for (int g = 0; g < n_groups; ++g) { MN(0); MN(1); MN(2); MN(3); MN(4); MN(5); MN(6); MN(7); }
      break;
    default: break;
    }
  } else {
    switch (role) {
    case B_VALU:
      for (int g = 0; g < n_groups; ++g) {
        VFMA(0); VFMA(1); VFMA(2); VFMA(3); VFMA(4); VFMA(5); VFMA(6); VFMA(7); VFMA(8); VFMA(9); VFMA(10); VFMA(11); VFMA(12); VFMA(13); VFMA(14); VFMA(15);
      }
      break;
    case B_DSREAD:
#define DR(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rd[(i) & 7]) : "v"(lds_addr), "n"(((i) & 7) * 1024) : "memory")
      for (int g = 0; g < n_groups; ++g) {
        DR(0); DR(1); DR(2); DR(3); DR(4); DR(5); DR(6); DR(7); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DR(8); DR(9); DR(10); DR(11); DR(12); DR(13); DR(14); DR(15); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      break;
    case B_DSADD:
#define DA(i) asm volatile("ds_add_f32 %0, %1 offset:%2" : : "v"(lds_addr), "v"(v[i]), "n"(((i) & 7) * 1024) : "memory")
      for (int g = 0; g < n_groups; ++g) {
        DA(0); DA(1); DA(2); DA(3); DA(4); DA(5); DA(6); DA(7); DA(8); DA(9); DA(10); DA(11); DA(12); DA(13); DA(14); DA(15);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      break;
    case B_MFMA32:
      for (int g = 0; g < n_groups; ++g) { MFMA32(0); MFMA32(1); MFMA32(2); MFMA32(3); MFMA32(4); MFMA32(5); MFMA32(6); MFMA32(7); }
      break;
    case B_VPK:
      for (int g = 0; g < n_groups; ++g) {
        VPKFMA(0); VPKFMA(1); VPKFMA(2); VPKFMA(3); VPKFMA(4); VPKFMA(5); VPKFMA(6); VPKFMA(7);
        VPKFMA(0); VPKFMA(1); VPKFMA(2); VPKFMA(3); VPKFMA(4); VPKFMA(5); VPKFMA(6); VPKFMA(7);
      }
      break;
    case B_MFMA32_V4:
      for (int g = 0; g < n_groups; ++g) { M4(0); M4(1); M4(2); M4(3); M4(4); M4(5); M4(6); M4(7); }
      break;
    default: break;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5] + acc4[i][1] + __uint_as_float(rd[i].x) + pv[i].x + pv[i].y;
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.678f) sink[0] = s + reinterpret_cast<float *>(smem)[lane];
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

struct Case { int a, b; const char *what; double ops_a, ops_b; };       // ops per group for the per-operation figures

int main() {
  const int blocks = 256, n_groups = 1000;                                // 8 000 MFMAs (32x32x16) | 16 000 partner operations per wave
  unsigned long long *out; float *sink;
  CK(hipMalloc(&out, blocks * 8 * 8)); CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe2), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<unsigned long long> h(blocks * 8);
  const Case cases[] = {
      {A_MFMA32, B_IDLE, "MFMA 32x32x16 alone", 8, 0},
      {A_IDLE, B_VALU, "partner VALU alone", 0, 16},
      {A_MFMA32, B_VALU, "MFMA 32x32x16 | partner VALU", 8, 16},
      {A_IDLE, B_VPK, "partner v_pk_fma_f32 alone", 0, 16},
      {A_MFMA32, B_VPK, "MFMA 32x32x16 | partner v_pk_fma_f32", 8, 16},
      {A_MFMA16, B_IDLE, "MFMA 16x16x32 alone (2 per unit of work)", 16, 0},
      {A_MFMA16, B_VALU, "MFMA 16x16x32 | partner VALU", 16, 16},
      {A_MFMA32_V1, B_IDLE, "MFMA + 1 VALU in the same wave", 8, 0},
      {A_MFMA32_V2, B_IDLE, "MFMA + 2 VALU in the same wave", 8, 0},
      {A_MFMA32_V4, B_IDLE, "MFMA + 4 VALU in the same wave", 8, 0},
      {A_MFMA32_V6, B_IDLE, "MFMA + 6 VALU in the same wave", 8, 0},
      {A_MFMA32_PK2, B_IDLE, "MFMA + 2 packed VALU in the same wave", 8, 0},
      {A_MFMA32_V4, B_MFMA32, "MFMA + 4 VALU in wave A | MFMA stream in wave B", 8, 8},
      {A_MFMA32_V4, B_VALU, "MFMA + 4 VALU in wave A | VALU in wave B", 8, 16},
      {A_MFMA32, B_MFMA32, "MFMA stream in both waves", 8, 8},
      {A_MFMA32_V4, B_MFMA32_V4, "MFMA + 4 VALU in both waves", 8, 8},
      {A_IDLE, B_DSREAD, "partner ds_read_b128 alone", 0, 16},
      {A_MFMA32, B_DSREAD, "MFMA | partner ds_read_b128", 8, 16},
      {A_IDLE, B_DSADD, "partner ds_add_f32 alone", 0, 16},
      {A_MFMA32, B_DSADD, "MFMA | partner ds_add_f32", 8, 16},
      {A_MFMA32_DS, B_IDLE, "MFMA + 1 ds_read_b128 in the same wave", 8, 0},
      {A_MFMA32_DS, B_VALU, "MFMA + 1 ds_read_b128 in wave A | partner VALU", 8, 16},
      {A_MFMA32_NOP0, B_IDLE, "MFMA + s_nop 0 alone", 8, 0},
      {A_MFMA32_NOP0, B_VALU, "MFMA + s_nop 0 | partner VALU", 8, 16},
  };
  for (int rep = 0; rep < 2; ++rep)
    for (const Case &cs : cases) {
      hipLaunchKernelGGL(probe2, dim3(blocks), dim3(512), 65536, 0, cs.a, cs.b, n_groups, out, sink);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      double ta = 0, tb = 0;
      for (int b = 0; b < blocks; ++b) { for (int w = 0; w < 4; ++w) ta += (double)h[b * 8 + w]; for (int w = 4; w < 8; ++w) tb += (double)h[b * 8 + w]; }
      ta /= blocks * 4; tb /= blocks * 4;
      if (rep == 1) {
        printf("%-52s A %9.0f cycles", cs.what, ta);
        if (cs.ops_a > 0) printf(" (%6.2f per MFMA)", ta / (n_groups * cs.ops_a));
        printf("   B %9.0f cycles", tb);
        if (cs.ops_b > 0) printf(" (%6.2f per %s)", tb / (n_groups * cs.ops_b), cs.ops_b == 8 ? "MFMA" : "operation");
        printf("\n");
      }
    }
  return 0;
}
