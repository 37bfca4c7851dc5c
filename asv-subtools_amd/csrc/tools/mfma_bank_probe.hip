// Developer tool: does the issue rate of a lone wave's v_mfma_f32_32x32x16_bf16 stream depend on WHICH registers hold the A / B
// operands and the accumulators?  (kernels_tdnn_chain4.hip: 38.5 cycles per instruction in a K loop with no loads at all,
// against 33.2 for tools/coissue_probe2.hip's stream, whose operands never change.)  Four waves per workgroup (one per SIMD),
// one workgroup per CU; everything timed is inline assembly with explicit registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 16 matrix instructions: tile t = a[16 t ..], A operand v[A0 + 4 * (t / 4) ..] (the same for four in a row), B operand v[B0 + 4 * (t % 4) ..]
#define M(T, A, B) "v_mfma_f32_32x32x16_bf16 a[" #T "*16:" #T "*16+15], v[" #A ":" #A "+3], v[" #B ":" #B "+3], a[" #T "*16:" #T "*16+15]\n\t"
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *out, int iters) {
  __shared__ float sink[4096];
  asm volatile("" ::: "a0", "a255");
  asm volatile("v_mov_b32 v50, %0" ::"v"((unsigned)(uintptr_t)sink % 65536u + (threadIdx.x & 255) * 16u) : "v50");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)        // the chain kernel's order: tile (i, j) = i * 4 + j, four in a row share A (= j), B = i   [A: v8.., B: v24.. both = 0 mod 4]
      asm volatile(M(0, 8, 24) M(4, 8, 28) M(8, 8, 32) M(12, 8, 36) M(1, 12, 24) M(5, 12, 28) M(9, 12, 32) M(13, 12, 36)
                   M(2, 16, 24) M(6, 16, 28) M(10, 16, 32) M(14, 16, 36) M(3, 20, 24) M(7, 20, 28) M(11, 20, 32) M(15, 20, 36) ::: CLOB);
    else if (MODE == 1)   // B operands shifted by two registers (tuples must be 64-bit aligned)
      asm volatile(M(0, 8, 26) M(4, 8, 30) M(8, 8, 34) M(12, 8, 38) M(1, 12, 26) M(5, 12, 30) M(9, 12, 34) M(13, 12, 38)
                   M(2, 16, 26) M(6, 16, 30) M(10, 16, 34) M(14, 16, 38) M(3, 20, 26) M(7, 20, 30) M(11, 20, 34) M(15, 20, 38) ::: CLOB);
    else if (MODE == 2)   // A and B shifted by two
      asm volatile(M(0, 10, 26) M(4, 10, 30) M(8, 10, 34) M(12, 10, 38) M(1, 14, 26) M(5, 14, 30) M(9, 14, 34) M(13, 14, 38)
                   M(2, 18, 26) M(6, 18, 30) M(10, 18, 34) M(14, 18, 38) M(3, 22, 26) M(7, 22, 30) M(11, 22, 34) M(15, 22, 38) ::: CLOB);
    else if (MODE == 3)   // consecutive tiles (0, 1, 2, ...) instead of stride 4, same operands as MODE 0
      asm volatile(M(0, 8, 24) M(1, 8, 28) M(2, 8, 32) M(3, 8, 36) M(4, 12, 24) M(5, 12, 28) M(6, 12, 32) M(7, 12, 36)
                   M(8, 16, 24) M(9, 16, 28) M(10, 16, 32) M(11, 16, 36) M(12, 20, 24) M(13, 20, 28) M(14, 20, 32) M(15, 20, 36) ::: CLOB);
    else if (MODE == 4)   // one operand pair for all (the coissue probe's stream)
      asm volatile(M(0, 8, 24) M(1, 8, 24) M(2, 8, 24) M(3, 8, 24) M(4, 8, 24) M(5, 8, 24) M(6, 8, 24) M(7, 8, 24)
                   M(8, 8, 24) M(9, 8, 24) M(10, 8, 24) M(11, 8, 24) M(12, 8, 24) M(13, 8, 24) M(14, 8, 24) M(15, 8, 24) ::: CLOB);
    else if (MODE == 5)   // only 8 tiles, each used twice per 16 (dependent distance 8)
      asm volatile(M(0, 8, 24) M(1, 8, 28) M(2, 8, 32) M(3, 8, 36) M(4, 12, 24) M(5, 12, 28) M(6, 12, 32) M(7, 12, 36)
                   M(0, 16, 24) M(1, 16, 28) M(2, 16, 32) M(3, 16, 36) M(4, 20, 24) M(5, 20, 28) M(6, 20, 32) M(7, 20, 36) ::: CLOB);
    else if (MODE == 6)   // A and B swapped roles: four in a row share B
      asm volatile(M(0, 24, 8) M(4, 28, 8) M(8, 32, 8) M(12, 36, 8) M(1, 24, 12) M(5, 28, 12) M(9, 32, 12) M(13, 36, 12)
                   M(2, 24, 16) M(6, 28, 16) M(10, 32, 16) M(14, 36, 16) M(3, 24, 20) M(7, 28, 20) M(11, 32, 20) M(15, 36, 20) ::: CLOB);
    else if (MODE == 7)   // MODE 0 with an s_nop 0 behind every instruction
      asm volatile(M(0, 8, 24) "s_nop 0\n\t" M(4, 8, 28) "s_nop 0\n\t" M(8, 8, 32) "s_nop 0\n\t" M(12, 8, 36) "s_nop 0\n\t" M(1, 12, 24) "s_nop 0\n\t" M(5, 12, 28) "s_nop 0\n\t" M(9, 12, 32) "s_nop 0\n\t" M(13, 12, 36) "s_nop 0\n\t"
                   M(2, 16, 24) "s_nop 0\n\t" M(6, 16, 28) "s_nop 0\n\t" M(10, 16, 32) "s_nop 0\n\t" M(14, 16, 36) "s_nop 0\n\t" M(3, 20, 24) "s_nop 0\n\t" M(7, 20, 28) "s_nop 0\n\t" M(11, 20, 32) "s_nop 0\n\t" M(15, 20, 36) "s_nop 0\n\t" ::: CLOB);
    else if (MODE == 8)   // 8 tiles; one v_accvgpr_read of an untouched AGPR per two matrix instructions
      asm volatile(M(0, 8, 24) M(1, 8, 28) "v_accvgpr_read_b32 v40, a[200]\n\t" M(2, 8, 32) M(3, 8, 36) "v_accvgpr_read_b32 v41, a[201]\n\t" M(4, 12, 24) M(5, 12, 28) "v_accvgpr_read_b32 v42, a[202]\n\t" M(6, 12, 32) M(7, 12, 36) "v_accvgpr_read_b32 v43, a[203]\n\t"
                   M(0, 16, 24) M(1, 16, 28) "v_accvgpr_read_b32 v44, a[204]\n\t" M(2, 16, 32) M(3, 16, 36) "v_accvgpr_read_b32 v45, a[205]\n\t" M(4, 20, 24) M(5, 20, 28) "v_accvgpr_read_b32 v46, a[206]\n\t" M(6, 20, 32) M(7, 20, 36) "v_accvgpr_read_b32 v47, a[207]\n\t" ::: CLOB);
    else if (MODE == 9)   // the same with a plain VALU operation instead
      asm volatile(M(0, 8, 24) M(1, 8, 28) "v_add_f32 v40, v48, v49\n\t" M(2, 8, 32) M(3, 8, 36) "v_add_f32 v41, v48, v49\n\t" M(4, 12, 24) M(5, 12, 28) "v_add_f32 v42, v48, v49\n\t" M(6, 12, 32) M(7, 12, 36) "v_add_f32 v43, v48, v49\n\t"
                   M(0, 16, 24) M(1, 16, 28) "v_add_f32 v44, v48, v49\n\t" M(2, 16, 32) M(3, 16, 36) "v_add_f32 v45, v48, v49\n\t" M(4, 20, 24) M(5, 20, 28) "v_add_f32 v46, v48, v49\n\t" M(6, 20, 32) M(7, 20, 36) "v_add_f32 v47, v48, v49\n\t" ::: CLOB);
    else if (MODE == 10)  // v_accvgpr_read + 4 independent plain VALU operations per two matrix instructions (the chain4 kernel's plain chunk)
#define V5(K) "v_accvgpr_read_b32 v40, a[20" #K "]\n\tv_add_f32 v41, v48, v49\n\tv_max_f32 v42, v48, v49\n\tv_add_f32 v43, v48, v49\n\tv_fmac_f32 v44, v48, v49\n\t"
      asm volatile(M(0, 8, 24) M(1, 8, 28) V5(0) M(2, 8, 32) M(3, 8, 36) V5(1) M(4, 12, 24) M(5, 12, 28) V5(2) M(6, 12, 32) M(7, 12, 36) V5(3)
                   M(0, 16, 24) M(1, 16, 28) V5(4) M(2, 16, 32) M(3, 16, 36) V5(5) M(4, 20, 24) M(5, 20, 28) V5(6) M(6, 20, 32) M(7, 20, 36) V5(7) ::: CLOB);
    else if (MODE == 11)  // 5 plain VALU operations per two matrix instructions, no AGPR read
#define V5P "v_add_f32 v40, v48, v49\n\tv_add_f32 v41, v48, v49\n\tv_max_f32 v42, v48, v49\n\tv_add_f32 v43, v48, v49\n\tv_fmac_f32 v44, v48, v49\n\t"
      asm volatile(M(0, 8, 24) M(1, 8, 28) V5P M(2, 8, 32) M(3, 8, 36) V5P M(4, 12, 24) M(5, 12, 28) V5P M(6, 12, 32) M(7, 12, 36) V5P
                   M(0, 16, 24) M(1, 16, 28) V5P M(2, 16, 32) M(3, 16, 36) V5P M(4, 20, 24) M(5, 20, 28) V5P M(6, 20, 32) M(7, 20, 36) V5P ::: CLOB);
    else if (MODE == 12)  // MODE 10 with the operations split 3 + 2 behind each of the two matrix instructions
#define V3(K) "v_accvgpr_read_b32 v40, a[20" #K "]\n\tv_add_f32 v41, v48, v49\n\tv_max_f32 v42, v48, v49\n\t"
#define V2 "v_add_f32 v43, v48, v49\n\tv_fmac_f32 v44, v48, v49\n\t"
      asm volatile(M(0, 8, 24) V3(0) M(1, 8, 28) V2 M(2, 8, 32) V3(1) M(3, 8, 36) V2 M(4, 12, 24) V3(2) M(5, 12, 28) V2 M(6, 12, 32) V3(3) M(7, 12, 36) V2
                   M(0, 16, 24) V3(4) M(1, 16, 28) V2 M(2, 16, 32) V3(5) M(3, 16, 36) V2 M(4, 20, 24) V3(6) M(5, 20, 28) V2 M(6, 20, 32) V3(7) M(7, 20, 36) V2 ::: CLOB);
    else if (MODE == 13)  // one ds_write_b128 of four AGPRs per four matrix instructions (the LDS path out of the accumulator file)
#define DW(K) "ds_write_b128 v50, a[" #K ":" #K "+3]\n\t"
      asm volatile(M(0, 8, 24) M(1, 8, 28) M(2, 8, 32) M(3, 8, 36) DW(200) M(4, 12, 24) M(5, 12, 28) M(6, 12, 32) M(7, 12, 36) DW(204)
                   M(0, 16, 24) M(1, 16, 28) M(2, 16, 32) M(3, 16, 36) DW(208) M(4, 20, 24) M(5, 20, 28) M(6, 20, 32) M(7, 20, 36) DW(212) "s_waitcnt lgkmcnt(0)\n\t" ::: CLOB, "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> int run(unsigned long long *out, const char *what) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(256), 0, 0, out, iters);
  hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(256), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(1024);
  CK(hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-90s %.2f cycles per MFMA\n", what, s / 1024 / iters / 16);
  return 0;
}

int main() {
  unsigned long long *out;
  CK(hipMalloc(&out, 1024 * 8));
  run<0>(out, "chain4 order: tiles 0,4,8,12,1,..; 4 in a row share A; A v8.., B v24.. (both 0 mod 4)");
  run<1>(out, "B operands at v26.. (2 mod 4)");
  run<2>(out, "A at v10.., B at v26.. (both 2 mod 4)");
  run<3>(out, "consecutive tiles 0,1,2,3,..");
  run<4>(out, "one A / B pair for all 16");
  run<5>(out, "8 tiles, each every 8th instruction");
  run<6>(out, "4 in a row share B instead of A");
  run<7>(out, "chain4 order + s_nop 0 behind every instruction");
  run<8>(out, "8 tiles + one v_accvgpr_read (untouched AGPR) per 2 MFMAs");
  run<9>(out, "8 tiles + one v_add_f32 per 2 MFMAs");
  run<10>(out, "8 tiles + v_accvgpr_read + 4 plain VALU per 2 MFMAs");
  run<11>(out, "8 tiles + 5 plain VALU per 2 MFMAs");
  run<12>(out, "8 tiles + (read, 2 VALU | 2 VALU) split behind the two MFMAs");
  run<13>(out, "8 tiles + one ds_write_b128 from AGPRs per 4 MFMAs");
  return 0;
}
