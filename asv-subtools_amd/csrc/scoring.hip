// Scoring back-end: length normalisation, global-mean subtraction, the batched
// enroll x test dot-product GEMM, trial-list gathers, Kaldi-style PLDA transform + LLR and EER.
//
// Replaces the Kaldi binaries the reference shells out to (score/process.sh:156-203
// ivector-mean / ivector-subtract-global-mean / ivector-normalize-length --scaleup=false;
// score/score.sh:82-121 ivector-compute-dot-products / ivector-plda-scoring), the in-repo
// Python PLDA they mirror (score/pyplda/plda_base.py:93-136,165-172) and the sort-based EER
// of computeEER-like-Bosaris.py:50-91.
#include <hipcub/hipcub.hpp>

#include <string.h>
#include <vector>

#include "device_utils.h"

namespace asv {
namespace {

// Scratch memory of one call: stream-ordered (hipMallocAsync / hipFreeAsync on the caller's stream, served from the
// device's memory pool after the first use), so it belongs to the device the call runs on and to this stream only -
// back-to-back calls on different streams or devices never share it, and nothing outlives the call.
struct Workspace {
  void *ptr = nullptr; hipStream_t s = nullptr;
  ~Workspace() { if (ptr) (void)hipFreeAsync(ptr, s); }
  int get(size_t bytes, hipStream_t stream, void **out) {
    s = stream;
    ASV_HIP_CHECK(hipMallocAsync(&ptr, bytes + 256, stream));
    *out = ptr;
    return ASV_OK;
  }
};

// Every entry point runs on the device that owns its (first) device pointer, whatever the caller's current device is, and
// restores the caller's current device when it returns (asv_internal.h DeviceGuard).
#define ASV_ENTER(ptr) ASV_ON_OWNER(ptr, "scoring")

__global__ __launch_bounds__(256) void col_mean_kernel(const float *x, int n, int dim, float *mean) {
  // one block per 64 columns; lanes along columns, 4 waves stride the rows
  __shared__ float sm[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  float s = 0.0f;
  if (col < dim)
    for (int r = wave; r < n; r += 4) s += x[(size_t)r * dim + col];
  sm[wave][threadIdx.x & 63] = s;
  __syncthreads();
  if (wave == 0 && col < dim) mean[col] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x])) / (float)n;
}

// ivector-mean with a spk2utt map: thread = column, rows of the group added in list order (f32), then scaled by 1 / n
__global__ __launch_bounds__(256) void group_mean_kernel(const float *x, int n, int dim, const int32_t *order, const int32_t *offsets, float *means, int32_t *counts) {
  const int g = blockIdx.x, col = blockIdx.y * 256 + threadIdx.x;
  const int b = offsets[g], e = offsets[g + 1];
  if (col == 0) counts[g] = e - b;
  if (col >= dim) return;
  float s = 0.0f;
  for (int k = b; k < e; ++k) {
    const int r = order[k];
    if (r >= 0 && r < n) s += x[(size_t)r * dim + col];
  }
  means[(size_t)g * dim + col] = (e > b) ? s * (float)(1.0 / (double)(e - b)) : 0.0f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// one wave per vector: x <- (x - mean); x <- x / ||x||
__global__ __launch_bounds__(256) void length_norm_kernel(float *x, int n, int dim, const float *mean, int normalize) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float *v = x + (size_t)row * dim;
  float ss = 0.0f;
  for (int i = lane; i < dim; i += 64) {
    float t = v[i];
    if (mean) t -= mean[i];
    v[i] = t;
    ss += t * t;
  }
  if (!normalize) return;
  ss = wave_sum(ss);
  const float nrm = sqrtf(ss);
  if (nrm > 0.0f)
    for (int i = lane; i < dim; i += 64) v[i] = v[i] / nrm;
}

__global__ __launch_bounds__(256) void dot_trials_kernel(const float *enroll, const float *test, int dim, const int32_t *ei,
                                                         const int32_t *ti, int n_trials, float *scores) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_trials) return;
  const float *a = enroll + (size_t)ei[t] * dim, *b = test + (size_t)ti[t] * dim;
  float s = 0.0f;
  for (int i = lane; i < dim; i += 64) s = fmaf(a[i], b[i], s);
  s = wave_sum(s);
  if (lane == 0) scores[t] = s;
}

// rows [n][dim] f32 -> padded [rows_pad][ld] f32 (zero fill), optionally minus mean
__global__ __launch_bounds__(256) void pad_rows_kernel(const float *src, int n, int dim, const float *mean, float *dst, int rows_pad, int ld) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows_pad * ld) return;
  const int r = (int)(gid / ld), c = (int)(gid % ld);
  float v = 0.0f;
  if (r < n && c < dim) { v = src[(size_t)r * dim + c]; if (mean) v -= mean[c]; }
  dst[gid] = v;
}

__global__ __launch_bounds__(256) void set_valid_kernel(uint32_t *bits, int rows, int rows_pad) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= rows_pad / 32) return;
  uint32_t b = 0;
  for (int i = 0; i < 32; ++i) if (w * 32 + i < rows) b |= (1u << i);
  bits[w] = b;
}

__global__ __launch_bounds__(256) void copy_out_kernel(const float *src, int ld, float *dst, int n, int m) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)n * m) return;
  dst[gid] = src[(size_t)(gid / m) * ld + (gid % m)];
}

// C[M][N] = A[M][K] . B[N][K]^T through the f32 MFMA GEMM of kernels_tdnn.hip.
// A gets an optional per-column mean subtracted while being padded.
int gemm_nt_f32(const float *A, int M, const float *a_mean, const float *B, int N, int K, float *C, hipStream_t s) {
  const int m_pad = round_up(M, kRowTile), n_pad = round_up(N, 128), k_pad = round_up(K, kChanAlign), ldc = round_up(N, kChanAlign);
  const size_t bytes = ((size_t)m_pad * k_pad + (size_t)n_pad * k_pad + (size_t)m_pad * ldc + n_pad + m_pad / 32) * 4;
  Workspace g_ws;
  void *ws = nullptr;
  int rc = g_ws.get(bytes, s, &ws);
  if (rc) return rc;
  float *Ap = reinterpret_cast<float *>(ws), *Bp = Ap + (size_t)m_pad * k_pad, *Cp = Bp + (size_t)n_pad * k_pad, *bias = Cp + (size_t)m_pad * ldc;
  uint32_t *valid = reinterpret_cast<uint32_t *>(bias + n_pad);
  auto blocks = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };
  hipLaunchKernelGGL(pad_rows_kernel, blocks((long long)m_pad * k_pad), dim3(256), 0, s, A, M, K, a_mean, Ap, m_pad, k_pad);
  hipLaunchKernelGGL(pad_rows_kernel, blocks((long long)n_pad * k_pad), dim3(256), 0, s, B, N, K, (const float *)nullptr, Bp, n_pad, k_pad);
  ASV_HIP_CHECK(hipMemsetAsync(bias, 0, (size_t)n_pad * 4, s));
  hipLaunchKernelGGL(set_valid_kernel, blocks(m_pad / 32), dim3(256), 0, s, valid, M, m_pad);
  ASV_HIP_CHECK(hipGetLastError());
  TdnnKernelParams p;
  memset(&p, 0, sizeof(p));
  p.x = Ap; p.ldx = k_pad; p.w = Bp; p.bias = bias; p.y = Cp; p.ldy = ldc; p.row_valid = valid;
  p.rows = m_pad; p.cin_pad = k_pad; p.cout_store = ldc; p.n_taps = 1; p.taps[0] = 0;
  if ((rc = launch_tdnn_mfma(p, false, true, s))) return rc;
  hipLaunchKernelGGL(copy_out_kernel, blocks((long long)M * N), dim3(256), 0, s, Cp, ldc, C, M, N);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

// PLDA length normalisation (plda_base.py:96-105,165-172), one wave per vector
__global__ __launch_bounds__(256) void plda_norm_kernel(float *y, int n, int dim, const float *psi, const int32_t *num_examples, int mode) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  float *v = y + (size_t)row * dim;
  const double ne = num_examples ? (double)num_examples[row] : 1.0;
  double acc = 0.0;
  for (int i = lane; i < dim; i += 64) {
    const double t = (double)v[i];
    acc += (mode == ASV_PLDA_NORM_SIMPLE) ? t * t : t * t / ((double)psi[i] + 1.0 / ne);
  }
  acc = wave_sum_f64(acc);
  const double factor = (mode == ASV_PLDA_NORM_SIMPLE) ? sqrt((double)dim) / sqrt(acc) : sqrt((double)dim / acc);
  for (int i = lane; i < dim; i += 64) v[i] = (float)(factor * (double)v[i]);
}

// PLDA log-likelihood ratio (plda_base.py:109-136), float64 like the reference
__global__ __launch_bounds__(256) void plda_llr_kernel(const float *enroll, const float *test, int dim, const float *psi, const int32_t *enroll_n,
                                                       const int32_t *ei, const int32_t *ti, int n_trials, float *scores) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_trials) return;
  const int e = ei[t];
  const float *g = enroll + (size_t)e * dim, *q = test + (size_t)ti[t] * dim;
  const double n = enroll_n ? (double)enroll_n[e] : 1.0;
  double given = 0.0, without = 0.0;
  for (int i = lane; i < dim; i += 64) {
    const double ps = (double)psi[i];
    const double mean = n * ps / (n * ps + 1.0) * (double)g[i];
    const double var = 1.0 + ps / (n * ps + 1.0);
    const double d = (double)q[i] - mean;
    given += log(var) + d * d / var;
    const double var0 = ps + 1.0;
    without += log(var0) + (double)q[i] * (double)q[i] / var0;
  }
  given = wave_sum_f64(given);
  without = wave_sum_f64(without);
  // the M_LOG_2PI*dim terms cancel in the ratio
  if (lane == 0) scores[t] = (float)(-0.5 * given + 0.5 * without);
}

// ---- EER --------------------------------------------------------------------------------
// key = order-preserving uint32 image of the score in the high word, label in the low word:
// ascending sort == Python's sorted([[score, label], ...]) (computeEER-like-Bosaris.py:64).
__global__ __launch_bounds__(256) void eer_keys_kernel(const float *scores, const int32_t *labels, int n, unsigned long long *keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t u = __float_as_uint(scores[i]);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  keys[i] = ((unsigned long long)u << 32) | (unsigned)(labels[i] != 0);
}
__global__ __launch_bounds__(256) void eer_labels_kernel(const unsigned long long *keys, int n, int *lab) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) lab[i] = (int)(keys[i] & 1ull);
}
// first sorted index with far <= frr (computeEER-like-Bosaris.py:75-80)
__global__ __launch_bounds__(256) void eer_cross_kernel(const int *cum_tgt, int n, int num_p, int num_n, int *first) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int fr = cum_tgt[i];                       // targets at or below this score
  const int fa = num_n - ((i + 1) - fr);           // non-targets above this score
  // far <= frr  <=>  fa/num_n <= fr/num_p, compared exactly in integers
  if ((long long)fa * num_p <= (long long)fr * num_n) atomicMin(first, i);
}

float key_to_score(unsigned long long k) {
  uint32_t u = (uint32_t)(k >> 32);
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace
}  // namespace asv

using namespace asv;

// ---------------------------------------------------------------------------------------
// Score normalisation (S-norm / AS-norm, reference score/ScoreNormalization.py:70-179: a pandas sort + groupby +
// per-trial Python loop).  Here: one workgroup per enrol / test vector selects its top_n cohort scores with a
// 4-pass radix select over order-preserving keys (no sort), then mean and sample standard deviation (ddof = 1) in
// float64 like pandas; trials are one thread each, or one wave each for cross-select (statistics per trial).
namespace {

__device__ __forceinline__ uint32_t score_key(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // larger score <=> larger key
}
__device__ __forceinline__ float key_score(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ double block_sum256(double v, double *sh) {   // sh: 4 doubles; all 256 threads get the total
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void cohort_topn_kernel(const float *S, int C, int n_sel, double *mean, double *sdev, int32_t *top_idx) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_digit, sh_above, sh_cnt;
  __shared__ uint32_t wave_tot[4];
  __shared__ double sh[4];
  const int tid = threadIdx.x, row = blockIdx.x;
  const float *r = S + (size_t)row * C;
  uint32_t thr = 0, need_eq = 0;                                  // n_sel == C: every key is > 0 (no NaN scores)
  if (n_sel < C) {
    uint32_t prefix = 0, mask = 0, remaining = (uint32_t)n_sel;
    for (int shift = 24; shift >= 0; shift -= 8) {
      hist[tid] = 0;
      __syncthreads();
      for (int c = tid; c < C; c += 256) {
        const uint32_t k = score_key(r[c]);
        if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t above = 0;
        int d = 255;
        while (d > 0 && above + hist[d] < remaining) { above += hist[d]; --d; }
        sh_digit = (uint32_t)d; sh_above = above;
      }
      __syncthreads();
      prefix |= sh_digit << shift;
      mask |= 255u << shift;
      remaining -= sh_above;
      __syncthreads();
    }
    thr = prefix;                                                 // key of the n_sel-th largest score
    need_eq = remaining;                                          // how many scores equal to it are inside the top
  }
  const double thr_v = need_eq ? (double)key_score(thr) : 0.0;     // nothing taken at the threshold: the term is void
  double acc = 0.0;
  for (int c = tid; c < C; c += 256) { const float x = r[c]; if (score_key(x) > thr) acc += (double)x; }
  const double mu = (block_sum256(acc, sh) + (double)need_eq * thr_v) / (double)n_sel;
  acc = 0.0;
  for (int c = tid; c < C; c += 256) { const float x = r[c]; if (score_key(x) > thr) { const double d = (double)x - mu; acc += d * d; } }
  const double ss = block_sum256(acc, sh) + (double)need_eq * (thr_v - mu) * (thr_v - mu);
  if (tid == 0) {
    mean[row] = mu;
    sdev[row] = sqrt(ss / (double)(n_sel - 1));                  // one selected score: 0/0 = NaN, like pandas
  }
  if (top_idx != nullptr) {
    // cohort ids of the selection: the scores above the threshold in any order, then the first need_eq equal
    // ones in cohort order (a stable descending sort would pick the same set)
    int32_t *dst = top_idx + (size_t)row * n_sel;
    const uint32_t n_gt = (uint32_t)n_sel - need_eq;
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    uint32_t eq_base = 0;
    for (int c0 = 0; c0 < C; c0 += 256) {
      const int c = c0 + tid;
      const uint32_t k = c < C ? score_key(r[c]) : 0u;
      const bool gt = c < C && k > thr, eq = c < C && k == thr && n_sel < C;
      if (gt) dst[atomicAdd(&sh_cnt, 1u)] = c;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(eq);
      const uint32_t in_wave = (uint32_t)__builtin_popcountll(b & ((1ull << (tid & 63)) - 1ull));
      if ((tid & 63) == 0) wave_tot[tid >> 6] = (uint32_t)__builtin_popcountll(b);
      __syncthreads();
      uint32_t before = eq_base;
      for (int w = 0; w < (tid >> 6); ++w) before += wave_tot[w];
      const uint32_t rank = before + in_wave;
      if (eq && rank < need_eq) dst[n_gt + rank] = c;
      eq_base += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void score_norm_trials_kernel(const float *scores, const int32_t *ei, const int32_t *ti, int n, const double *mu_e,
                                                                const double *sd_e, const double *mu_t, const double *sd_t, float *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double s = (double)scores[i];
  const int e = ei[i], t = ti[i];
  out[i] = (float)(0.5 * ((s - mu_e[e]) / sd_e[e] + (s - mu_t[t]) / sd_t[t]));
}

// cross-select: the enrol-side statistics of trial (e, t) run over e's scores against the cohort vectors that are
// top_n for t, and vice versa (ScoreNormalization.py:139-148).  One wave per trial.
__global__ __launch_bounds__(256) void score_norm_cross_kernel(const float *ec, const float *tc, int C, int n_sel, const int32_t *top_e, const int32_t *top_t,
                                                               const float *scores, const int32_t *ei, const int32_t *ti, int n, float *out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const int e = ei[i], t = ti[i];
  const float *er = ec + (size_t)e * C, *tr = tc + (size_t)t * C;
  const int32_t *ie = top_e + (size_t)e * n_sel, *it = top_t + (size_t)t * n_sel;
  double se = 0.0, st = 0.0;
  for (int k = lane; k < n_sel; k += 64) { se += (double)er[it[k]]; st += (double)tr[ie[k]]; }
  for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); st += __shfl_xor(st, o); }
  const double me = se / n_sel, mt = st / n_sel;
  double qe = 0.0, qt = 0.0;
  for (int k = lane; k < n_sel; k += 64) {
    const double a = (double)er[it[k]] - me, b = (double)tr[ie[k]] - mt;
    qe += a * a; qt += b * b;
  }
  for (int o = 32; o > 0; o >>= 1) { qe += __shfl_xor(qe, o); qt += __shfl_xor(qt, o); }
  if (lane == 0) {
    const double s = (double)scores[i];
    out[i] = (float)(0.5 * ((s - me) / sqrt(qe / (double)(n_sel - 1)) + (s - mt) / sqrt(qt / (double)(n_sel - 1))));
  }
}

}  // namespace

extern "C" {

int asv_mean_vec(const float *x, int n, int dim, float *mean, void *stream) {
  ASV_REQUIRE(x && mean && n >= 1 && dim >= 1, "asv_mean_vec: bad argument");
  ASV_ENTER(x);
  hipLaunchKernelGGL(col_mean_kernel, dim3((dim + 63) / 64), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, dim, mean);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int asv_length_norm(float *x, int n, int dim, const float *mean, int normalize, void *stream) {
  ASV_REQUIRE(x && n >= 1 && dim >= 1, "asv_length_norm: bad argument");
  ASV_ENTER(x);
  hipLaunchKernelGGL(length_norm_kernel, dim3((n + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, dim, mean, normalize);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int asv_dot_score_matrix(const float *enroll, int n_enroll, const float *test, int n_test, int dim, float *scores, void *stream) {
  ASV_REQUIRE(enroll && test && scores && n_enroll >= 1 && n_test >= 1 && dim >= 1, "asv_dot_score_matrix: bad argument");
  ASV_ENTER(enroll);
  return gemm_nt_f32(enroll, n_enroll, nullptr, test, n_test, dim, scores, reinterpret_cast<hipStream_t>(stream));
}

int asv_dot_score_trials(const float *enroll, const float *test, int dim, const int32_t *ei, const int32_t *ti, int n_trials, float *scores, void *stream) {
  ASV_REQUIRE(enroll && test && ei && ti && scores && dim >= 1 && n_trials >= 0, "asv_dot_score_trials: bad argument");
  ASV_ENTER(enroll);
  if (n_trials == 0) return ASV_OK;
  hipLaunchKernelGGL(dot_trials_kernel, dim3((n_trials + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), enroll, test, dim, ei, ti, n_trials, scores);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int asv_plda_transform(const float *x, int n, int dim, const float *mean, const float *transform, const float *psi, const int32_t *num_examples,
                       int length_norm, float *y, void *stream) {
  ASV_REQUIRE(x && transform && y && n >= 1 && dim >= 1, "asv_plda_transform: bad argument");
  ASV_ENTER(x);
  ASV_REQUIRE(length_norm >= ASV_PLDA_NORM_NONE && length_norm <= ASV_PLDA_NORM_PSI, "asv_plda_transform: length_norm %d", length_norm);
  ASV_REQUIRE(length_norm != ASV_PLDA_NORM_PSI || psi != nullptr, "asv_plda_transform: psi needed for PLDA length normalisation");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // y[n][d] = sum_k T[d][k] (x[n][k] - mean[k])  (plda_base.py:93-97 with offset = -T mean, 158-163)
  int rc = gemm_nt_f32(x, n, mean, transform, dim, dim, y, s);
  if (rc) return rc;
  if (length_norm != ASV_PLDA_NORM_NONE) {
    hipLaunchKernelGGL(plda_norm_kernel, dim3((n + 3) / 4), dim3(256), 0, s, y, n, dim, psi, num_examples, length_norm);
    ASV_HIP_CHECK(hipGetLastError());
  }
  return ASV_OK;
}

int asv_plda_llr_trials(const float *enroll, const float *test, int dim, const float *psi, const int32_t *enroll_n, const int32_t *ei, const int32_t *ti,
                        int n_trials, float *scores, void *stream) {
  ASV_REQUIRE(enroll && test && psi && ei && ti && scores && dim >= 1 && n_trials >= 0, "asv_plda_llr_trials: bad argument");
  ASV_ENTER(enroll);
  if (n_trials == 0) return ASV_OK;
  hipLaunchKernelGGL(plda_llr_kernel, dim3((n_trials + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), enroll, test, dim, psi, enroll_n, ei, ti,
                     n_trials, scores);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int asv_group_mean(const float *x, int n, int dim, const int32_t *order, const int32_t *offsets, int n_groups, float *means, int32_t *counts, void *stream) {
  ASV_REQUIRE(x && order && offsets && means && counts && n >= 1 && dim >= 1 && n_groups >= 0, "asv_group_mean: bad argument");
  ASV_ENTER(x);
  if (n_groups == 0) return ASV_OK;
  hipLaunchKernelGGL(group_mean_kernel, dim3(n_groups, (dim + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, n, dim, order, offsets, means, counts);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int asv_eer(const float *scores, const int32_t *labels, int n, float *eer_percent, float *threshold, void *stream) {
  ASV_REQUIRE(scores && labels && eer_percent && threshold && n >= 2, "asv_eer: bad argument");
  ASV_ENTER(scores);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  size_t sort_tmp = 0, scan_tmp = 0;
  ASV_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, (unsigned long long *)nullptr, (unsigned long long *)nullptr, n, 0, 64, s));
  ASV_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(nullptr, scan_tmp, (int *)nullptr, (int *)nullptr, n, s));
  const size_t tmp = std::max(sort_tmp, scan_tmp);
  const size_t n8 = round_up64((int64_t)n * 8, 256), n4 = round_up64((int64_t)n * 4, 256);
  Workspace g_ws;
  void *ws = nullptr;
  int rc = g_ws.get(2 * n8 + 2 * n4 + 256 + tmp, s, &ws);
  if (rc) return rc;
  unsigned char *b = reinterpret_cast<unsigned char *>(ws);
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(b), *sorted = reinterpret_cast<unsigned long long *>(b + n8);
  int *lab = reinterpret_cast<int *>(b + 2 * n8), *cum = reinterpret_cast<int *>(b + 2 * n8 + n4);
  int *first = reinterpret_cast<int *>(b + 2 * n8 + 2 * n4);
  void *cub_tmp = b + 2 * n8 + 2 * n4 + 256;
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(eer_keys_kernel, grid, block, 0, s, scores, labels, n, keys);
  size_t t1 = tmp;
  ASV_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(cub_tmp, t1, keys, sorted, n, 0, 64, s));
  hipLaunchKernelGGL(eer_labels_kernel, grid, block, 0, s, sorted, n, lab);
  size_t t2 = tmp;
  ASV_HIP_CHECK(hipcub::DeviceScan::InclusiveSum(cub_tmp, t2, lab, cum, n, s));
  int num_p = 0;
  ASV_HIP_CHECK(hipMemcpyAsync(&num_p, cum + (n - 1), 4, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  const int num_n = n - num_p;
  ASV_REQUIRE(num_p > 0 && num_n > 0, "asv_eer: need both target and non-target trials (%d / %d)", num_p, num_n);
  const int init = n;
  ASV_HIP_CHECK(hipMemcpyAsync(first, &init, 4, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(eer_cross_kernel, grid, block, 0, s, cum, n, num_p, num_n, first);
  int idx = n;
  ASV_HIP_CHECK(hipMemcpyAsync(&idx, first, 4, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  ASV_REQUIRE(idx < n, "asv_eer: FAR never drops to FRR");
  // the crossing point and its predecessor ("memory"), computeEER-like-Bosaris.py:78-90
  int cums[2] = {0, 0};
  unsigned long long ks[2] = {0, 0};
  const int lo = idx > 0 ? idx - 1 : idx;
  ASV_HIP_CHECK(hipMemcpy(cums, cum + lo, 4 * (idx - lo + 1), hipMemcpyDeviceToHost));
  ASV_HIP_CHECK(hipMemcpy(ks, sorted + lo, 8 * (idx - lo + 1), hipMemcpyDeviceToHost));
  auto rates = [&](int i, int cum_t, double &far, double &frr) {
    frr = (double)cum_t / num_p;
    far = (double)(num_n - ((i + 1) - cum_t)) / num_n;
  };
  double far, frr, eer; float thr;
  rates(idx, cums[idx - lo], far, frr);
  eer = (far + frr) / 2; thr = key_to_score(ks[idx - lo]);
  if (idx > 0) {
    double pfar, pfrr;
    rates(idx - 1, cums[0], pfar, pfrr);
    const double lnow = far > frr ? far - frr : frr - far, lmem = pfar > pfrr ? pfar - pfrr : pfrr - pfar;
    if (!(lnow <= lmem)) { eer = (pfar + pfrr) / 2; thr = key_to_score(ks[0]); }
  }
  *eer_percent = (float)(eer * 100.0);
  *threshold = thr;
  return ASV_OK;
}

int asv_score_norm(const float *enroll_cohort, int n_enroll, const float *test_cohort, int n_test, int n_cohort, const int32_t *ei, const int32_t *ti,
                   const float *scores, int n_trials, int top_n, int cross_select, float *normed, void *stream) {
  ASV_REQUIRE(enroll_cohort && test_cohort && ei && ti && scores && normed && n_enroll >= 1 && n_test >= 1 && n_cohort >= 1 && n_trials >= 0,
              "asv_score_norm: bad argument");
  ASV_ENTER(scores);
  if (n_trials == 0) return ASV_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int n_sel = (top_n <= 0 || top_n > n_cohort) ? n_cohort : top_n;
  const int rows = n_enroll + n_test;
  const size_t stat_bytes = round_up64((int64_t)rows * 2 * 8, 256);
  const size_t idx_bytes = cross_select ? round_up64((int64_t)rows * n_sel * 4, 256) : 0;
  Workspace g_ws;
  void *ws = nullptr;
  int rc = g_ws.get(stat_bytes + idx_bytes, s, &ws);
  if (rc) return rc;
  double *mu_e = reinterpret_cast<double *>(ws), *sd_e = mu_e + n_enroll, *mu_t = sd_e + n_enroll, *sd_t = mu_t + n_test;
  int32_t *top_e = cross_select ? reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(ws) + stat_bytes) : nullptr;
  int32_t *top_t = cross_select ? top_e + (size_t)n_enroll * n_sel : nullptr;
  hipLaunchKernelGGL(cohort_topn_kernel, dim3(n_enroll), dim3(256), 0, s, enroll_cohort, n_cohort, n_sel, mu_e, sd_e, top_e);
  hipLaunchKernelGGL(cohort_topn_kernel, dim3(n_test), dim3(256), 0, s, test_cohort, n_cohort, n_sel, mu_t, sd_t, top_t);
  if (cross_select)
    hipLaunchKernelGGL(score_norm_cross_kernel, dim3((n_trials + 3) / 4), dim3(256), 0, s, enroll_cohort, test_cohort, n_cohort, n_sel, top_e, top_t, scores, ei, ti,
                       n_trials, normed);
  else
    hipLaunchKernelGGL(score_norm_trials_kernel, dim3((n_trials + 255) / 256), dim3(256), 0, s, scores, ei, ti, n_trials, mu_e, sd_e, mu_t, sd_t, normed);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // extern "C"
