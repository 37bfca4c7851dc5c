// Ablation / micro-benchmark of the TDNN GEMM kernels on synthetic operands (developer tool,
// not part of libasv_amd.so's ABI).  Usage: gemm_ablate [rows cin cout ntaps iters]
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../asv_internal.h"

using namespace asv;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  int rows = argc > 1 ? atoi(argv[1]) : 52224, cin = argc > 2 ? atoi(argv[2]) : 512, cout = argc > 3 ? atoi(argv[3]) : 512;
  int ntaps = argc > 4 ? atoi(argv[4]) : 3, iters = argc > 5 ? atoi(argv[5]) : 10;
  const int xpad = argc > 6 ? atoi(argv[6]) : 0;   // extra elements of row pitch on the activation matrix (channel-camping experiment)
  rows = round_up(rows, 256);
  const int cout_pad = round_up(cout, 256);
  std::vector<uint16_t> hx((size_t)rows * (cin + xpad)), hw((size_t)cout_pad * ntaps * cin);
  srand(1);
  for (auto &v : hx) v = f32_to_bf16_host((rand() / (float)RAND_MAX) * 2 - 1);
  for (auto &v : hw) v = f32_to_bf16_host(((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f);
  void *x, *w, *y, *zero; float *bias, *scale, *shift; uint32_t *valid;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&y, (size_t)rows * cout_pad * 2));
  CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
  CK(hipMalloc(&bias, cout_pad * 4)); CK(hipMalloc(&scale, cout_pad * 4)); CK(hipMalloc(&shift, cout_pad * 4)); CK(hipMalloc(&valid, rows / 32 * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, cout_pad * 4)); CK(hipMemset(shift, 0, cout_pad * 4)); CK(hipMemset(valid, 0xff, rows / 32 * 4));
  std::vector<float> ones(cout_pad, 1.0f); CK(hipMemcpy(scale, ones.data(), cout_pad * 4, hipMemcpyHostToDevice));
  std::vector<float> wf32((size_t)cout * cin * ntaps);
  for (auto &v : wf32) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  const int tapsets0[5][5] = {{0}, {-1, 1}, {-2, 0, 2}, {-3, -1, 1, 3}, {-2, -1, 0, 1, 2}};
  std::vector<uint16_t> hfrag(tdnn_weight_frag_elems(cout_pad, cin, ntaps));
  {
    // dense kernel [cout][cin][tot] with left context = first tap
    const int left = tapsets0[ntaps - 1][0], tot = tapsets0[ntaps - 1][ntaps - 1] - left + 1;
    std::vector<float> dense((size_t)cout * cin * tot, 0.0f);
    for (int co = 0; co < cout; ++co) for (int ci = 0; ci < cin; ++ci) for (int t = 0; t < ntaps; ++t)
      dense[((size_t)co * cin + ci) * tot + (tapsets0[ntaps - 1][t] - left)] = wf32[((size_t)co * cin + ci) * ntaps + t];
    pack_tdnn_weight_frags(dense.data(), cout, cin, tot, left, tapsets0[ntaps - 1], ntaps, cout_pad, cin, hfrag.data());
  }
  void *wfrag; CK(hipMalloc(&wfrag, hfrag.size() * 2)); CK(hipMemcpy(wfrag, hfrag.data(), hfrag.size() * 2, hipMemcpyHostToDevice));
  TdnnKernelParams p; memset(&p, 0, sizeof(p));
  p.wfrag = wfrag;
  p.x = x; p.w = w; p.bias = bias; p.scale = scale; p.shift = shift; p.y = y; p.row_valid = valid; p.zero16 = zero;
  p.ldx = cin + xpad; p.ldy = cout_pad; p.rows = rows; p.cin_pad = cin; p.cout_store = round_up(cout, 16); p.n_taps = ntaps;
  const int tapsets[5][5] = {{0}, {-1, 1}, {-2, 0, 2}, {-3, -1, 1, 3}, {-2, -1, 0, 1, 2}};
  for (int t = 0; t < ntaps; ++t) p.taps[t] = tapsets[ntaps - 1][t];
  p.act1 = ASV_ACT_RELU;
  const double flops = 2.0 * rows * cin * cout * ntaps;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("rows=%d cin=%d cout=%d taps=%d xpad=%d  (%.1f GFLOP)\n", rows, cin, cout, ntaps, xpad, flops / 1e9);
  const char *names[] = {"big: full", "big: no LDS-DMA in loop", "big: MFMA + barrier only", "big: DMA + ds_read, no MFMA", "big: no epilogue stores", "big: DMA (cache-hot) + ds_read", "small 128x128 (v1)", "big: DMA + barrier only", "big: ds_read + barrier only", "big3 (2 WG/CU): full", "big3 (2 WG/CU): MFMA only", "big3 (2 WG/CU): no epilogue stores", "big3 (1 WG/CU): full", "big3 (1 WG/CU): MFMA only"};
  for (int v = 0; v <= 13; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < iters; ++i) {
        int rc = v == 6 ? launch_tdnn_mfma(p, true, false, 0) : v == 9 ? launch_tdnn_big3_variant(p, 0, 0) : v == 10 ? launch_tdnn_big3_variant(p, 2, 0) : v == 11 ? launch_tdnn_big3_variant(p, 4, 0) : v == 12 ? launch_tdnn_big3_variant(p, 100, 0) : v == 13 ? launch_tdnn_big3_variant(p, 102, 0) : launch_tdnn_big_variant(p, v, 0);
        if (rc) { printf("launch failed: %s\n", asv_last_error()); return 1; }
      }
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 1) printf("  %-30s %9.1f us  %8.1f TFLOP/s\n", names[v], 1e3 * ms / iters, flops * iters / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
