// Ablation / micro-benchmark of the TDNN GEMM kernels on synthetic operands (developer tool,
// not part of libasv_amd.so's ABI).  Usage: gemm_ablate [rows cin cout ntaps iters]
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../asv_internal.h"
#include "../host_convert.h"

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
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  if (getenv("ABLATE_GRID")) {
    // 3x3 convolution over a row-flattened (time, frequency) grid of pitch P: 9 row offsets dt * P + df; v1 kernel only
    const int P = atoi(getenv("ABLATE_GRID"));
    ntaps = 9; p.n_taps = 9;
    for (int dt = -1, t = 0; dt <= 1; ++dt) for (int df = -1; df <= 1; ++df) p.taps[t++] = dt * P + df;
    std::vector<uint16_t> hw9((size_t)cout_pad * 9 * cin);
    for (auto &v : hw9) v = f32_to_bf16_host(((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f);
    void *w9; CK(hipMalloc(&w9, hw9.size() * 2)); CK(hipMemcpy(w9, hw9.data(), hw9.size() * 2, hipMemcpyHostToDevice));
    p.w = w9;
    const double fl = 2.0 * rows * cin * cout * 9;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < iters; ++i) if (launch_tdnn_mfma(p, true, false, 0)) { printf("launch failed: %s\n", asv_last_error()); return 1; }
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 1) printf("  v1 128x128, 3x3 grid taps (pitch %d)  %9.1f us  %8.1f TFLOP/s\n", P, 1e3 * ms / iters, fl * iters / (ms * 1e-3) / 1e12);
    }
    return 0;
  }
  p.act1 = ASV_ACT_RELU;
  const bool pool_mode = getenv("ABLATE_POOL") != nullptr;
  if (pool_mode) {
    // utterances of 200 frames with 4 gap rows between them, like a C2 batch; fused statistics pooling epilogue
    std::vector<int32_t> rs(rows, -1);
    int seg = 0;
    for (int r = 4; r + 200 <= rows; r += 204, ++seg) for (int k = 0; k < 200; ++k) rs[r + k] = seg;
    std::vector<uint32_t> vb(rows / 32, 0);
    for (int r = 0; r < rows; ++r) if (rs[r] >= 0) vb[r >> 5] |= 1u << (r & 31);
    int32_t *drs; CK(hipMalloc(&drs, rows * 4)); CK(hipMemcpy(drs, rs.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(valid, vb.data(), rows / 8, hipMemcpyHostToDevice));
    p.row_seg = drs; p.pool_slots = 2; p.ld_partial = cout_pad;
    float *pp; CK(hipMalloc(&pp, (size_t)(rows / 128) * 2 * 3 * cout_pad * 4)); p.pool_partial = pp;
  }
  p.tune = getenv("ABLATE_TUNE") ? (int)strtol(getenv("ABLATE_TUNE"), nullptr, 0) : 0;
  const double flops = 2.0 * rows * cin * cout * ntaps;
  printf("rows=%d cin=%d cout=%d taps=%d xpad=%d  (%.1f GFLOP)\n", rows, cin, cout, ntaps, xpad, flops / 1e9);
  const char *names[] = {"big: full", "big: no LDS-DMA in loop", "big: MFMA + barrier only", "big: DMA + ds_read, no MFMA", "big: no epilogue stores", "big: DMA (cache-hot) + ds_read", "small 128x128 (v1)", "big: DMA + barrier only", "big: ds_read + barrier only", "big3 (2 WG/CU): full", "big3 (2 WG/CU): MFMA only", "big3 (2 WG/CU): no epilogue stores", "big3 (1 WG/CU): full", "big3 (1 WG/CU): MFMA only", "big3 64-row (3 WG/CU): full", "big3 64-row (3 WG/CU): MFMA only"};
  for (int v = 9; v <= (pool_mode ? 9 : 15); ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < iters; ++i) {
        int rc = v == 6 ? launch_tdnn_mfma(p, true, false, 0) : v == 9 ? launch_tdnn_big3_variant(p, 0, 0) : v == 10 ? launch_tdnn_big3_variant(p, 2, 0) : v == 11 ? launch_tdnn_big3_variant(p, 4, 0) : v == 12 ? launch_tdnn_big3_variant(p, 100, 0) : v == 13 ? launch_tdnn_big3_variant(p, 102, 0) : v == 14 ? launch_tdnn_big3_variant(p, 200, 0) : v == 15 ? launch_tdnn_big3_variant(p, 202, 0) : ASV_EINVAL;   // (variants 0-5, 7, 8 belonged to the removed 256x256 both-operands-through-LDS kernel)
        if (rc) { printf("launch failed: %s\n", asv_last_error()); return 1; }
      }
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 1) printf("  %-30s %9.1f us  %8.1f TFLOP/s\n", names[v], 1e3 * ms / iters, flops * iters / (ms * 1e-3) / 1e12);
    }
  }
  // ---- per-workgroup phase timeline of the default kernel (variants 5 / 6 write s_memrealtime stamps)
  const int tl_variants[] = {5, 6, 17, 18, 20, 19, 21, 22, 24, 29};
  const char *tl_names[] = {"full", "MFMA only", "no wf loads", "no ds_reads", "no DMA in loop", "no wf, no ds_reads", "no wf, no DMA", "no ds_reads, no DMA", "ds_reads to dummy regs", "ds_reads to dummy, no wf, no DMA"};
  for (int vi = 0; vi < (pool_mode ? 1 : 10); ++vi) {
    const int variant = tl_variants[vi];
    const int n_wg = (rows / 128) * (cout_pad / 256);
    unsigned long long *dbg; CK(hipMalloc(&dbg, (size_t)n_wg * (64 + 1024))); CK(hipMemset(dbg, 0, (size_t)n_wg * (64 + 1024)));
    p.partial = reinterpret_cast<float *>(dbg);
    if (getenv("ABLATE_HOT")) for (int i = 0; i < atoi(getenv("ABLATE_HOT")); ++i) launch_tdnn_big3_variant(p, 0, 0);   // sustained load first
    for (int i = 0; i < 3; ++i) if (launch_tdnn_big3_variant(p, variant, 0)) { printf("launch failed: %s\n", asv_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)n_wg * 8);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < n_wg; ++w) { if (h[w * 8] < t0) t0 = h[w * 8]; if (h[w * 8 + 3] > t1) t1 = h[w * 8 + 3]; }
    printf("timeline (%s): %d workgroups, first start -> last end %.2f us (100 MHz ticks)\n", tl_names[vi], n_wg, (t1 - t0) * 0.01);
    {
      double cyc = 0, us = 0;
      for (int w = 0; w < n_wg; ++w) { cyc += (double)(h[w * 8 + 6] - h[w * 8 + 5]); us += (h[w * 8 + 2] - h[w * 8 + 1]) * 0.01; }
      printf("  shader clock inside the main loops: %.2f GHz (s_memtime ticks / s_memrealtime)\n", cyc / us * 1e-3);
    }
    double sp = 0, sl = 0, se = 0, sl2 = 0; int late = 0;
    for (int w = 0; w < n_wg; ++w) {
      sp += (h[w * 8 + 1] - h[w * 8]) * 0.01; se += (h[w * 8 + 3] - h[w * 8 + 2]) * 0.01;
      if ((h[w * 8] - t0) * 0.01 > 5.0) { ++late; sl2 += (h[w * 8 + 2] - h[w * 8 + 1]) * 0.01; } else sl += (h[w * 8 + 2] - h[w * 8 + 1]) * 0.01;
    }
    printf("  mean per workgroup: prologue %.2f us, epilogue %.2f us; main loop %.2f us in the first wave of %d workgroups, %.2f us in the %d later ones\n", sp / n_wg, se / n_wg, sl / (n_wg - late), n_wg - late, late ? sl2 / late : 0.0, late);
    FILE *f = variant > 6 ? nullptr : fopen(variant == 5 ? "gpurun_out/timeline_full.csv" : "gpurun_out/timeline_mfma.csv", "w");
    if (f) {
      fprintf(f, "wg,start_us,prologue_end_us,loop_end_us,end_us,xcc_id,hw_id\n");
      for (int w = 0; w < n_wg; ++w)
        fprintf(f, "%d,%.2f,%.2f,%.2f,%.2f,%u,%u\n", w, (h[w * 8] - t0) * 0.01, (h[w * 8 + 1] - t0) * 0.01, (h[w * 8 + 2] - t0) * 0.01, (h[w * 8 + 3] - t0) * 0.01,
                (unsigned)(h[w * 8 + 4] >> 32), (unsigned)(h[w * 8 + 4] & 0xffffffffu));
      fclose(f);
    }
    if (variant == 5) {
      // shader-clock stamps of wave 0 after every k-group of the first 32 steps: mean cycles per group by position
      std::vector<unsigned long long> g((size_t)n_wg * 128);
      CK(hipMemcpy(g.data(), dbg + (size_t)n_wg * 8, g.size() * 8, hipMemcpyDeviceToHost));
      const int nst = std::min(32, ntaps * ((cin + 63) / 64));
      for (int pass = 0; pass < 2; ++pass) {
        printf("  cycles per k-group, wave 0, %s workgroups:", pass == 0 ? "first-wave" : "later");
        for (int k = 1; k < nst * 4; ++k) {
          double m = 0; int n = 0;
          for (int w = 0; w < n_wg; ++w) {
            const bool late = (h[w * 8] - t0) * 0.01 > 5.0;
            if (late != (pass == 1)) continue;
            m += (double)(g[(size_t)w * 128 + k] - g[(size_t)w * 128 + k - 1]); ++n;
          }
          if (k % 4 == 0) printf(" |");
          printf(" %.0f", n ? m / n : 0.0);
        }
        printf("\n");
      }
    }
    CK(hipFree(dbg));
  }
  return 0;
}
