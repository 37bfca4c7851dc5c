// A/B of kernels_tdnn_p8.hip (256 x 256, both operands through LDS-DMA, four phases per K-tile) against kernels_tdnn_v3.hip (128 x 256,
// window through LDS, weight fragments from L2) on the same synthetic operands: outputs compared bit for bit and against an f64 host
// reference on sampled entries; timing in interleaved rounds inside one process (cdna_hip_programming.md 5.4 rules 13 / 24 / 25:
// uniform random operands in [-1, 1), medians over rounds).  Developer tool, not part of libasv_amd.so's ABI.
//     p8_probe [rows cin cout ntaps iters rounds]
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include "../asv_internal.h"
#include "../host_convert.h"

using namespace asv;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
  int rows = argc > 1 ? atoi(argv[1]) : 52224, cin = argc > 2 ? atoi(argv[2]) : 512, cout = argc > 3 ? atoi(argv[3]) : 512;
  const int ntaps = argc > 4 ? atoi(argv[4]) : 3, iters = argc > 5 ? atoi(argv[5]) : 20, rounds = argc > 6 ? atoi(argv[6]) : 7;
  rows = round_up(rows, 256);
  const int cout_pad = round_up(cout, 256);
  const int tapsets[5][5] = {{0}, {-1, 1}, {-2, 0, 2}, {-3, -1, 1, 3}, {-2, -1, 0, 1, 2}};
  std::vector<uint16_t> hx((size_t)rows * cin);
  srand(1);
  for (auto &v : hx) v = f32_to_bf16_host((rand() / (float)RAND_MAX) * 2 - 1);
  // utterances of 200 frames with 4 gap rows between them (gap rows hold zeros and are invalid), like a configs[1] batch
  std::vector<uint32_t> vb(rows / 32, 0);
  for (int r = 0; r < rows; ++r) {
    const bool valid = r >= 4 && ((r - 4) % 204) < 200 && r < rows - 4;
    if (valid) vb[r >> 5] |= 1u << (r & 31);
    else for (int c = 0; c < cin; ++c) hx[(size_t)r * cin + c] = 0;
  }
  std::vector<float> wf32((size_t)cout * cin * ntaps);
  for (auto &v : wf32) v = ((rand() / (float)RAND_MAX) * 2 - 1) * 0.05f;
  const int left = tapsets[ntaps - 1][0], tot = tapsets[ntaps - 1][ntaps - 1] - left + 1;
  std::vector<float> dense((size_t)cout * cin * tot, 0.0f);
  for (int co = 0; co < cout; ++co) for (int ci = 0; ci < cin; ++ci) for (int t = 0; t < ntaps; ++t)
    dense[((size_t)co * cin + ci) * tot + (tapsets[ntaps - 1][t] - left)] = wf32[((size_t)co * cin + ci) * ntaps + t];
  std::vector<uint16_t> hw((size_t)cout_pad * ntaps * cin), hfrag(tdnn_weight_frag_elems(cout_pad, cin, ntaps));
  pack_tdnn_weight(dense.data(), cout, cin, tot, left, tapsets[ntaps - 1], ntaps, cout_pad, cin, ET_BF16, hw.data());
  pack_tdnn_weight_frags(dense.data(), cout, cin, tot, left, tapsets[ntaps - 1], ntaps, cout_pad, cin, hfrag.data());
  std::vector<float> hb(cout_pad), hs(cout_pad), hsh(cout_pad);
  for (int c = 0; c < cout_pad; ++c) { hb[c] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.2f; hs[c] = 0.5f + rand() / (float)RAND_MAX; hsh[c] = ((rand() / (float)RAND_MAX) - 0.5f) * 0.4f; }
  void *x, *w, *wfrag, *y0, *y1, *zero; float *bias, *scale, *shift; uint32_t *valid;
  CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&wfrag, hfrag.size() * 2));
  CK(hipMalloc(&y0, (size_t)rows * cout_pad * 2)); CK(hipMalloc(&y1, (size_t)rows * cout_pad * 2));
  CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
  CK(hipMalloc(&bias, cout_pad * 4)); CK(hipMalloc(&scale, cout_pad * 4)); CK(hipMalloc(&shift, cout_pad * 4)); CK(hipMalloc(&valid, rows / 32 * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(wfrag, hfrag.data(), hfrag.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), cout_pad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(scale, hs.data(), cout_pad * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(shift, hsh.data(), cout_pad * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(valid, vb.data(), rows / 8, hipMemcpyHostToDevice));
  TdnnKernelParams p; memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.wfrag = wfrag; p.bias = bias; p.scale = scale; p.shift = shift; p.row_valid = valid; p.zero16 = zero;
  p.ldx = cin; p.ldy = cout_pad; p.rows = rows; p.cin_pad = cin; p.cout_store = round_up(cout, 16); p.n_taps = ntaps; p.et = ET_BF16; p.act1 = ASV_ACT_RELU;
  for (int t = 0; t < ntaps; ++t) p.taps[t] = tapsets[ntaps - 1][t];
  const double flops = 2.0 * rows * cin * cout * ntaps;
  printf("rows=%d cin=%d cout=%d taps=%d  (%.1f GFLOP; %d tiles of 256 x 256 on 256 CUs = %.2f rounds)\n", rows, cin, cout, ntaps, flops / 1e9,
         (rows / 256) * (cout_pad / 256), (rows / 256) * (cout_pad / 256) / 256.0);
  if (!tdnn_p8_supported(p, ET_BF16, false)) { printf("shape not supported by the p8 kernel\n"); return 1; }
  // ---- correctness: p8 against big3, bit for bit, and both against f64 on sampled entries
  CK(hipMemset(y0, 0xff, (size_t)rows * cout_pad * 2)); CK(hipMemset(y1, 0xee, (size_t)rows * cout_pad * 2));
  p.y = y0; if (launch_tdnn_big3_variant(p, 0, 0)) { printf("big3 launch failed: %s\n", asv_last_error()); return 1; }
  p.y = y1; if (launch_tdnn_p8_variant(p, 0, 0)) { printf("p8 launch failed: %s\n", asv_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> h0((size_t)rows * cout_pad), h1((size_t)rows * cout_pad);
  CK(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
  size_t diff = 0, first = (size_t)-1;
  for (int r = 0; r < rows; ++r) for (int c = 0; c < p.cout_store; ++c) { const size_t i = (size_t)r * cout_pad + c; if (h0[i] != h1[i]) { if (!diff) first = i; ++diff; } }
  printf("p8 vs big3: %zu of %zu stored values differ", diff, (size_t)rows * p.cout_store);
  if (diff) printf(" (first at row %zu channel %zu: %04x vs %04x)", first / cout_pad, first % cout_pad, h0[first], h1[first]);
  printf("\n");
  double worst = 0;
  for (int s = 0; s < 4000; ++s) {
    const int r = s < 600 ? (s % 300) + (s < 300 ? 0 : rows - 300) : rand() % rows, c = rand() % p.cout_store;
    double a = 0;
    if (c < cout) for (int t = 0; t < ntaps; ++t) {
      const int rr = std::min(std::max(r + p.taps[t], 0), rows - 1);
      for (int k = 0; k < cin; ++k) a += (double)bf16_to_f32_host(hx[(size_t)rr * cin + k]) * (double)bf16_to_f32_host(hw[((size_t)c * ntaps + t) * cin + k]);
    }
    double z = std::max(a + hb[c], 0.0) * hs[c] + hsh[c];
    if (!((vb[r >> 5] >> (r & 31)) & 1u)) z = 0;
    const double got = bf16_to_f32_host(h1[(size_t)r * cout_pad + c]);
    worst = std::max(worst, std::fabs(got - z) / (std::fabs(z) + 1.0));
  }
  printf("p8 vs f64 reference on 4000 sampled outputs: worst |got - want| / (|want| + 1) = %.3g (bf16 output rounding = 3.9e-3)\n", worst);
  // repeat the bitwise check a few times (a race between DMA and reads shows as rare differing tiles)
  size_t race = 0;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipMemset(y1, 0x11 * rep, (size_t)rows * cout_pad * 2));
    p.y = y1; if (launch_tdnn_p8_variant(p, 0, 0)) return 1;
    CK(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
    for (int r = 0; r < rows; ++r) for (int c = 0; c < p.cout_store; ++c) { const size_t i = (size_t)r * cout_pad + c; race += h0[i] != h1[i]; }
  }
  printf("p8 vs big3 over 6 more launches: %zu differing values\n", race);
  {
    CK(hipMemset(y1, 0x55, (size_t)rows * cout_pad * 2));
    p.y = y1;
    if (launch_tdnn_p8_variant(p, 50, 0) == 0) {
      CK(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
      size_t d1 = 0;
      for (int r = 0; r < rows; ++r) for (int c = 0; c < p.cout_store; ++c) { const size_t i = (size_t)r * cout_pad + c; d1 += h0[i] != h1[i]; }
      printf("p8 one-tile form vs big3: %zu differing values\n", d1);
    }
  }
  {
    CK(hipMemset(y1, 0x33, (size_t)rows * cout_pad * 2));
    p.y = y1;
    if (tdnn_p8_supported(p, ET_BF16, false) && ((cin / 64) * ntaps) % 2 == 0 && launch_tdnn_p8_variant(p, 40, 0) == 0) {
      CK(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
      size_t d4 = 0;
      for (int r = 0; r < rows; ++r) for (int c = 0; c < p.cout_store; ++c) { const size_t i = (size_t)r * cout_pad + c; d4 += h0[i] != h1[i]; }
      printf("p8 four-phase form vs big3: %zu differing values\n", d4);
    }
  }
  // ---- phase stamps of the production kernel (variant 7): prologue | K loop | epilogue, cycles of wave 0, mean over the workgroups
  {
    const int n_wg = (rows / 256) * (cout_pad / 256);
    unsigned long long *dbg; CK(hipMalloc(&dbg, (size_t)n_wg * 64)); CK(hipMemset(dbg, 0, (size_t)n_wg * 64));
    p.partial = reinterpret_cast<float *>(dbg); p.y = y1;
    for (int i = 0; i < 3; ++i) if (launch_tdnn_p8_variant(p, 7, 0)) { printf("stamp launch failed: %s\n", asv_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)n_wg * 8);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < n_wg; ++w) { a += (double)(h[w * 8 + 1] - h[w * 8]); b += (double)(h[w * 8 + 2] - h[w * 8 + 1]); c += (double)(h[w * 8 + 3] - h[w * 8 + 2]); }
    printf("p8 phase stamps (s_memtime cycles of wave 0, mean over %d workgroups): prologue %.0f | K loop %.0f (%.0f per K-tile) | epilogue + store drain %.0f\n", n_wg, a / n_wg, b / n_wg,
           b / n_wg / ((cin / 64) * ntaps), c / n_wg);
    p.partial = nullptr;
    CK(hipFree(dbg));
  }
  // ---- timing: interleaved rounds
  struct Var { const char *name; int kind, variant; };
  const Var vars[] = {{"big3 128x256 (2 WG/CU)", 0, 0}, {"p8 persistent (production)", 1, 0}, {"p8 one tile per workgroup", 1, 50}, {"p8 four-phase, one tile", 1, 40},
                      {"p8 one tile, no stagger", 1, 1}, {"p8 one tile, without setprio", 1, 2}, {"p8 one tile, skeleton", 1, 4}};
  const int nv = sizeof(vars) / sizeof(vars[0]);
  std::vector<std::vector<float>> us(nv);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  p.y = y0;
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < nv; ++v) {
      if (vars[v].variant >= 40 && ((cin / 64) * ntaps) % 2 != 0) { if (r > 0) us[v].push_back(0.0f); continue; }
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < iters; ++i) {
        const int rc = vars[v].kind == 0 ? launch_tdnn_big3_variant(p, vars[v].variant, 0) : launch_tdnn_p8_variant(p, vars[v].variant, 0);
        if (rc) { printf("launch failed: %s\n", asv_last_error()); return 1; }
      }
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (r > 0) us[v].push_back(1e3f * ms / iters);
    }
  for (int v = 0; v < nv; ++v) {
    std::sort(us[v].begin(), us[v].end());
    const float med = us[v][us[v].size() / 2];
    printf("  %-34s median %8.1f us  %7.1f TFLOP/s   (min %.1f us, max %.1f us)\n", vars[v].name, med, flops / (med * 1e-6) / 1e12, us[v].front(), us[v].back());
  }
  return 0;
}
