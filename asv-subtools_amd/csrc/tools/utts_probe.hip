// Micro-benchmark of the pooled-domain GEMM (developer tool).  Usage: utts_probe [rows cin cout iters]
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../asv_internal.h"
using namespace asv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 256, cin = argc > 2 ? atoi(argv[2]) : 3008, cout = argc > 3 ? atoi(argv[3]) : 512, iters = argc > 4 ? atoi(argv[4]) : 50;
  const int rows_pad = round_up(rows, 256), cout_pad = round_up(cout, 256), ksteps = (cin + 31) / 32;
  float *x, *y, *w, *bias; uint16_t *hi, *lo; uint32_t *valid;
  CK(hipMalloc(&x, (size_t)rows_pad * cin * 4)); CK(hipMemset(x, 0, (size_t)rows_pad * cin * 4));
  CK(hipMalloc(&y, (size_t)rows_pad * cout_pad * 4)); CK(hipMalloc(&w, (size_t)cout_pad * cin * 4)); CK(hipMemset(w, 0, (size_t)cout_pad * cin * 4));
  CK(hipMalloc(&hi, (size_t)(cout_pad / 32) * ksteps * 2048)); CK(hipMalloc(&lo, (size_t)(cout_pad / 32) * ksteps * 2048));
  CK(hipMemset(hi, 0, (size_t)(cout_pad / 32) * ksteps * 2048)); CK(hipMemset(lo, 0, (size_t)(cout_pad / 32) * ksteps * 2048));
  CK(hipMalloc(&bias, cout_pad * 4)); CK(hipMemset(bias, 0, cout_pad * 4)); CK(hipMalloc(&valid, rows_pad / 8)); CK(hipMemset(valid, 0xff, rows_pad / 8));
  TdnnKernelParams p; memset(&p, 0, sizeof(p));
  p.x = x; p.ldx = cin; p.w = w; p.wfrag = hi; p.wlo = lo; p.bias = bias; p.y = y; p.ldy = cout_pad; p.rows = rows_pad; p.cin_pad = cin; p.cout_store = cout;
  p.n_taps = 1; p.row_valid = valid; p.act1 = ASV_ACT_RELU;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("utts gemm %d x %d -> %d\n", rows, cin, cout);
  const char *names[] = {"full", "no X refetch", "no W refetch", "no refetch", "hi*hi only", "no LDS turn", "no refetch, hi*hi only, no turn"};
  const int tunes[] = {0, 1, 2, 3, 4, 8, 15};
  for (int split = 1; split >= 0; --split)
    for (int v = 0; v < (split ? 7 : 1); ++v) {
      p.tune = tunes[v];
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < iters; ++i) if (launch_utts_gemm(p, rows, split, 0)) { printf("launch failed: %s\n", asv_last_error()); return 1; }
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("  %-8s %-34s %8.2f us\n", split ? "split" : "exact", names[v], 1e3 * ms / iters);
      }
    }
  return 0;
}
