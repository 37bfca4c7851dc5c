// PLDA training on the device: the statistics and EM of reference
// score/pyplda/plda_base.py:37-81 (PldaStats) and :227-300 (PldaEstimation), in float64 like the numpy original.
//
// The reference inverts one D x D matrix per speaker class and iteration (plda_base.py:283); the mixing covariance
// (between^-1 + n within^-1)^-1 only depends on the class size n, so the classes are grouped by size and everything
// becomes a handful of dense f64 products per iteration:
//   scatter  = X^T X - sum_k n_k c_k c_k^T                       (once: N x D^2, the only N-scale product)
//   mix_n    = (B^-1 + n W^-1)^-1,   G_n = n mix_n W^-1           (one batched SPD inverse + product per distinct n)
//   w_k      = G_n m_k  (rows of  M_n G_n^T),   between_stats = sum_n K_n mix_n + sum_k w_k w_k^T
//   within_stats = scatter + sum_n n K_n mix_n + sum_k n_k (m_k - w_k)(m_k - w_k)^T
// MI355X runs f64 FMAs at the same rate on the vector and the matrix pipes (78.6 TFLOP/s), so the products are a
// register-tiled VALU kernel (64 x 64 tile, 4 x 4 per thread, operands through LDS with generic strides so that one
// kernel serves A^T diag(s) A, A B^T and A B; split-K partials are summed in a fixed order).  The SPD inverses are
// in-place Gauss-Jordan sweeps, one workgroup per matrix, batched over the distinct class sizes.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "asv_internal.h"

namespace asv {
namespace {

struct Gemm64Params {
  const void *a, *b;
  double *c;
  long long sa_i, sa_k, sb_k, sb_j;      // element strides: A(i, k) = a[i sa_i + k sa_k], B(k, j) = b[k sb_k + j sb_j]
  long long batch_a, batch_b, batch_c;   // element strides between batches
  const int *kidx;                       // optional: the k-th term reads row kidx[k] of both operands
  const double *kscale;                  // optional: the k-th term is multiplied by kscale[k]
  int m, n, k, ldc;
  int ksplit;                            // > 1: c is [batch][ksplit][m][n] partials
  double alpha, beta;                    // ksplit == 1: c = alpha * product + beta * c
};

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm64_kernel(const Gemm64Params p) {
  constexpr int BK = 16;
  __shared__ double As[BK][65], Bs[BK][65];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int batch = blockIdx.z / p.ksplit, split = blockIdx.z % p.ksplit;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const TA *a = static_cast<const TA *>(p.a) + (size_t)batch * p.batch_a;
  const TB *b = static_cast<const TB *>(p.b) + (size_t)batch * p.batch_b;
  const int kchunk = ((p.k + p.ksplit - 1) / p.ksplit + BK - 1) / BK * BK;
  const int kb = split * kchunk, ke = min(p.k, kb + kchunk);
  double acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  const bool a_k_fast = p.sa_k == 1, b_k_fast = p.sb_k == 1;
  for (int k0 = kb; k0 < ke; k0 += BK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 256 * q;
      {
        const int kk = a_k_fast ? (e & 15) : (e >> 6), ii = a_k_fast ? (e >> 4) : (e & 63);
        const int k = k0 + kk, i = i0 + ii;
        double v = 0.0;
        if (k < ke && i < p.m) {
          const long long kr = p.kidx ? p.kidx[k] : k;
          v = (double)a[(size_t)i * p.sa_i + (size_t)kr * p.sa_k];
          if (p.kscale) v *= p.kscale[k];
        }
        As[kk][ii] = v;
      }
      {
        const int kk = b_k_fast ? (e & 15) : (e >> 6), jj = b_k_fast ? (e >> 4) : (e & 63);
        const int k = k0 + kk, j = j0 + jj;
        double v = 0.0;
        if (k < ke && j < p.n) {
          const long long kr = p.kidx ? p.kidx[k] : k;
          v = (double)b[(size_t)kr * p.sb_k + (size_t)j * p.sb_j];
        }
        Bs[kk][jj] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = As[kk][ty * 4 + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = Bs[kk][tx * 4 + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
    }
    __syncthreads();
  }
  if (p.ksplit > 1) {
    double *c = p.c + ((size_t)blockIdx.z * p.m) * p.n;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int i = i0 + ty * 4 + r, j = j0 + tx * 4 + cc;
        if (i < p.m && j < p.n) c[(size_t)i * p.n + j] = acc[r][cc];
      }
  } else {
    double *c = p.c + (size_t)batch * p.batch_c;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int i = i0 + ty * 4 + r, j = j0 + tx * 4 + cc;
        if (i < p.m && j < p.n) {
          double *dst = c + (size_t)i * p.ldc + j;
          *dst = p.alpha * acc[r][cc] + (p.beta != 0.0 ? p.beta * *dst : 0.0);
        }
      }
  }
}

// c = alpha * (sum of the ksplit partials, in order) + beta * c
__global__ __launch_bounds__(256) void splitk_sum_kernel(const double *part, int ksplit, int m, int n, double alpha, double beta, double *c, int ldc) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)m * n) return;
  double s = 0.0;
  for (int z = 0; z < ksplit; ++z) s += part[(size_t)z * m * n + e];
  double *dst = c + (size_t)(e / n) * ldc + (e % n);
  *dst = alpha * s + (beta != 0.0 ? beta * *dst : 0.0);
}

// In-place inverse of symmetric positive definite matrices by Gauss-Jordan sweeps without pivoting (the pivots of an SPD
// matrix are positive); one workgroup per matrix of the batch, the pivot row and column of a step staged in LDS.
__global__ __launch_bounds__(1024) void spd_inverse_kernel(double *mats, int d, int *status) {
  extern __shared__ double sh[];                      // rowp[d] | colp[d]
  double *rowp = sh, *colp = sh + d;
  double *a = mats + (size_t)blockIdx.x * d * d;
  for (int p = 0; p < d; ++p) {
    const double piv = a[(size_t)p * d + p];
    if (threadIdx.x == 0 && !(piv > 0.0)) atomicOr(status, 1);
    const double inv = 1.0 / piv;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
      rowp[j] = j == p ? inv : a[(size_t)p * d + j] * inv;
      colp[j] = a[(size_t)j * d + p];
    }
    __syncthreads();
    for (long long e = threadIdx.x; e < (long long)d * d; e += blockDim.x) {
      const int i = (int)(e / d), j = (int)(e % d);
      double v;
      if (i == p) v = rowp[j];
      else if (j == p) v = -colp[i] * inv;
      else v = fma(-colp[i], rowp[j], a[e]);
      a[e] = v;
    }
    __syncthreads();
  }
}

// class means (classes in the caller's order, rows through `order`), f64: one workgroup per (class, 64 columns)
__global__ __launch_bounds__(256) void class_mean_kernel(const float *x, int ldx, const int *order, const long long *class_off, int dim, double *cmeans) {
  __shared__ double red[4][64];
  const int k = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const long long b = class_off[k], e = class_off[k + 1];
  double s = 0.0;
  if (c < dim) for (long long i = b + r; i < e; i += 4) s += (double)x[(size_t)order[i] * ldx + c];
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < dim) cmeans[(size_t)k * dim + c] = (red[0][c & 63] + red[1][c & 63] + red[2][c & 63] + red[3][c & 63]) / (double)(e - b);
}

// column means of a [rows][dim] matrix (the global mean = mean of the class means, every class weighs 1)
__global__ __launch_bounds__(256) void col_mean_kernel(const double *m, int rows, int dim, double *out) {
  __shared__ double red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  double s = 0.0;
  if (c < dim) for (int i = r; i < rows; i += 4) s += m[(size_t)i * dim + c];
  red[r][threadIdx.x & 63] = s;
  __syncthreads();
  if (r == 0 && c < dim) out[c] = (red[0][c & 63] + red[1][c & 63] + red[2][c & 63] + red[3][c & 63]) / (double)rows;
}

// out[i][c] = a[i][c] - (b ? b[i][c] : v[c])
__global__ __launch_bounds__(256) void sub_rows_kernel(const double *a, const double *b, const double *v, long long rows, int dim, double *out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= rows * dim) return;
  out[e] = a[e] - (b ? b[e] : v[e % dim]);
}

// mats[u] = b_inv + n_u * w_inv
__global__ __launch_bounds__(256) void mix_arg_kernel(const double *b_inv, const double *w_inv, const double *sizes, long long dd, double *mats) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= dd) return;
  mats[(size_t)blockIdx.y * dd + e] = b_inv[e] + sizes[blockIdx.y] * w_inv[e];
}

// dst = base (or 0) + sum_u coef[u] * mats[u]; dst *= scale
__global__ __launch_bounds__(256) void weighted_sum_kernel(const double *base, const double *mats, const double *coef, int n_mats, long long dd, double *dst) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= dd) return;
  double s = base ? base[e] : 0.0;
  for (int u = 0; u < n_mats; ++u) s = fma(coef[u], mats[(size_t)u * dd + e], s);
  dst[e] = s;
}

__global__ __launch_bounds__(256) void scale_kernel(double *m, long long n, double f) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) m[e] *= f;
}

// one wave per trial: s = <EL[e], t> + <TL[t], e> + <EG[e], e> + <TG[t], t> + <e + t, c>   (gaussian-plda-scoring.py:23-29)
__global__ __launch_bounds__(256) void two_cov_trials_kernel(const float *enroll, const float *test, int dim, const double *el, const double *tl, const double *eg,
                                                             const double *tg, const double *c, const int32_t *ei, const int32_t *ti, int n_trials, double *scores) {
  const int tr = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tr >= n_trials) return;
  const size_t eo = (size_t)ei[tr] * dim, to = (size_t)ti[tr] * dim;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0;
  for (int d = lane; d < dim; d += 64) {
    const double e = (double)enroll[eo + d], t = (double)test[to + d];
    s1 += el[eo + d] * t;
    s2 += tl[to + d] * e;
    s3 += eg[eo + d] * e;
    s4 += tg[to + d] * t;
    s5 += (e + t) * c[d];
  }
  double s = (((s1 + s2) + s3) + s4) + s5;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) scores[tr] = s;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { ASV_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8)); return ASV_OK; }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

template <typename TA, typename TB>
int gemm64(Gemm64Params p, int batches, DevBuf &partials, size_t &partial_cap, hipStream_t s) {
  // split K when the tile grid alone cannot fill the device
  const int tiles = ((p.m + 63) / 64) * ((p.n + 63) / 64) * batches;
  int ksplit = 1;
  if (tiles < 512 && p.k > 2048) ksplit = std::min(64, std::max(1, std::min(1024 / tiles, p.k / 1024)));
  p.ksplit = ksplit;
  double *c = p.c;
  const double alpha = p.alpha, beta = p.beta;
  if (ksplit > 1) {
    ASV_REQUIRE(batches == 1, "gemm64: split-K is single-batch");
    const size_t need = (size_t)ksplit * p.m * p.n * 8;
    if (need > partial_cap) {
      if (partials.p) { ASV_HIP_CHECK(hipStreamSynchronize(s)); ASV_HIP_CHECK(hipFree(partials.p)); partials.p = nullptr; }
      int rc = partials.alloc(need);
      if (rc) return rc;
      partial_cap = need;
    }
    p.c = partials.as<double>();
  }
  const dim3 grid((p.n + 63) / 64, (p.m + 63) / 64, batches * ksplit);
  hipLaunchKernelGGL((gemm64_kernel<TA, TB>), grid, dim3(256), 0, s, p);
  if (ksplit > 1) {
    const long long n = (long long)p.m * p.n;
    hipLaunchKernelGGL(splitk_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partials.as<double>(), ksplit, p.m, p.n, alpha, beta, c, p.ldc);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace
}  // namespace asv

using namespace asv;

extern "C" int asv_plda_train(const float *x, int ldx, int n_rows, int dim, const int *order, const long long *class_offsets, int n_classes,
                              int num_iters, double *mean_out, double *within_out, double *between_out, void *stream) {
  ASV_REQUIRE(x && order && class_offsets && mean_out && within_out && between_out, "asv_plda_train: null argument");
  ASV_ON_OWNER(x, "asv_plda_train");
  ASV_REQUIRE(dim >= 1 && dim <= 4096 && ldx >= dim && n_classes >= 2 && num_iters >= 0, "asv_plda_train: dim %d / ld %d / classes %d / iterations %d", dim, ldx, n_classes, num_iters);
  ASV_REQUIRE(class_offsets[0] == 0 && class_offsets[n_classes] == n_rows, "asv_plda_train: class_offsets must run from 0 to n_rows");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int D = dim, K = n_classes;
  const long long DD = (long long)D * D;
  // classes grouped by size (plda_base.py:236-240 insists on stats sorted by size; any order gives the same sums)
  std::vector<double> sizes_k(K);
  std::map<long long, int> distinct;                          // size -> number of classes
  for (int k = 0; k < K; ++k) {
    const long long n = class_offsets[k + 1] - class_offsets[k];
    ASV_REQUIRE(n >= 1, "asv_plda_train: class %d is empty", k);
    ASV_REQUIRE(k == 0 || n >= class_offsets[k] - class_offsets[k - 1], "asv_plda_train: classes must be ordered by size (ascending)");
    sizes_k[k] = (double)n;
    distinct[n] += 1;
  }
  const int U = (int)distinct.size();
  std::vector<double> sizes_u, count_u, nk_u;
  std::vector<int> first_u;
  {
    int at = 0;
    for (auto &kv : distinct) {
      sizes_u.push_back((double)kv.first); count_u.push_back((double)kv.second); nk_u.push_back((double)kv.first * kv.second);
      first_u.push_back(at); at += kv.second;
    }
  }
  DevBuf d_order, d_off, d_cmeans, d_gmean, d_m, d_w, d_mw, d_scatter, d_within, d_between, d_inv2, d_mix, d_g, d_sizes_u, d_count_u, d_nk_u, d_sizes_k,
      d_status, partials;
  size_t partial_cap = 0;
  int rc;
#define TRY(e) do { if ((rc = (e))) return rc; } while (0)
  TRY(d_order.alloc((size_t)n_rows * 4)); TRY(d_off.alloc(((size_t)K + 1) * 8));
  TRY(d_cmeans.alloc((size_t)K * D * 8)); TRY(d_gmean.alloc((size_t)D * 8)); TRY(d_m.alloc((size_t)K * D * 8));
  TRY(d_w.alloc((size_t)K * D * 8)); TRY(d_mw.alloc((size_t)K * D * 8));
  TRY(d_scatter.alloc(DD * 8)); TRY(d_within.alloc(DD * 8)); TRY(d_between.alloc(DD * 8)); TRY(d_inv2.alloc(2 * DD * 8));
  TRY(d_mix.alloc((size_t)U * DD * 8)); TRY(d_g.alloc((size_t)U * DD * 8));
  TRY(d_sizes_u.alloc((size_t)U * 8)); TRY(d_count_u.alloc((size_t)U * 8)); TRY(d_nk_u.alloc((size_t)U * 8)); TRY(d_sizes_k.alloc((size_t)K * 8));
  TRY(d_status.alloc(4));
  ASV_HIP_CHECK(hipMemcpyAsync(d_order.p, order, (size_t)n_rows * 4, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_off.p, class_offsets, ((size_t)K + 1) * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_sizes_u.p, sizes_u.data(), (size_t)U * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_count_u.p, count_u.data(), (size_t)U * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_nk_u.p, nk_u.data(), (size_t)U * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_sizes_k.p, sizes_k.data(), (size_t)K * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemsetAsync(d_status.p, 0, 4, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));                     // the host vectors above are locals
  const unsigned dd_blocks = (unsigned)((DD + 255) / 256), kd_blocks = (unsigned)(((long long)K * D + 255) / 256);

  // ---- statistics (plda_base.py:37-81)
  hipLaunchKernelGGL(class_mean_kernel, dim3((unsigned)K, (unsigned)((D + 63) / 64)), dim3(256), 0, s, x, ldx, d_order.as<int>(), d_off.as<long long>(), D, d_cmeans.as<double>());
  hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)((D + 63) / 64)), dim3(256), 0, s, d_cmeans.as<double>(), K, D, d_gmean.as<double>());
  hipLaunchKernelGGL(sub_rows_kernel, dim3(kd_blocks), dim3(256), 0, s, d_cmeans.as<double>(), (const double *)nullptr, d_gmean.as<double>(), (long long)K, D, d_m.as<double>());
  {
    Gemm64Params g; memset(&g, 0, sizeof(g));                 // scatter = X^T X (rows through `order`: all of them, any order)
    g.a = x; g.b = x; g.c = d_scatter.as<double>(); g.sa_i = 1; g.sa_k = ldx; g.sb_k = ldx; g.sb_j = 1; g.kidx = d_order.as<int>();
    g.m = D; g.n = D; g.k = n_rows; g.ldc = D; g.alpha = 1.0; g.beta = 0.0;
    TRY((gemm64<float, float>(g, 1, partials, partial_cap, s)));
    g.a = d_cmeans.p; g.b = d_cmeans.p; g.sa_k = D; g.sb_k = D; g.kidx = nullptr; g.kscale = d_sizes_k.as<double>();   // - sum_k n_k c_k c_k^T
    g.k = K; g.alpha = -1.0; g.beta = 1.0;
    TRY((gemm64<double, double>(g, 1, partials, partial_cap, s)));
  }
  // ---- EM (plda_base.py:248-300), within = between = I to start
  {
    std::vector<double> eye((size_t)DD, 0.0);
    for (int i = 0; i < D; ++i) eye[(size_t)i * D + i] = 1.0;
    ASV_HIP_CHECK(hipMemcpyAsync(d_within.p, eye.data(), DD * 8, hipMemcpyHostToDevice, s));
    ASV_HIP_CHECK(hipMemcpyAsync(d_between.p, eye.data(), DD * 8, hipMemcpyHostToDevice, s));
    ASV_HIP_CHECK(hipStreamSynchronize(s));
  }
  const int inv_threads = 1024;
  const size_t inv_lds = (size_t)2 * D * 8;
  for (int it = 0; it < num_iters; ++it) {
    double *w_inv = d_inv2.as<double>(), *b_inv = d_inv2.as<double>() + DD;
    ASV_HIP_CHECK(hipMemcpyAsync(w_inv, d_within.p, DD * 8, hipMemcpyDeviceToDevice, s));
    ASV_HIP_CHECK(hipMemcpyAsync(b_inv, d_between.p, DD * 8, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(spd_inverse_kernel, dim3(2), dim3(inv_threads), inv_lds, s, d_inv2.as<double>(), D, d_status.as<int>());
    hipLaunchKernelGGL(mix_arg_kernel, dim3(dd_blocks, (unsigned)U), dim3(256), 0, s, b_inv, w_inv, d_sizes_u.as<double>(), DD, d_mix.as<double>());
    hipLaunchKernelGGL(spd_inverse_kernel, dim3((unsigned)U), dim3(inv_threads), inv_lds, s, d_mix.as<double>(), D, d_status.as<int>());
    {
      Gemm64Params g; memset(&g, 0, sizeof(g));               // G_u = mix_u W^-1 (the factor n_u rides on alpha below)
      g.a = d_mix.p; g.b = w_inv; g.c = d_g.as<double>(); g.sa_i = D; g.sa_k = 1; g.sb_k = D; g.sb_j = 1; g.batch_a = DD; g.batch_b = 0; g.batch_c = DD;
      g.m = D; g.n = D; g.k = D; g.ldc = D; g.alpha = 1.0; g.beta = 0.0;
      TRY((gemm64<double, double>(g, U, partials, partial_cap, s)));
    }
    for (int u = 0; u < U; ++u) {                             // w_k = n G_u m_k for the classes of size n: rows of M_u G_u^T
      Gemm64Params g; memset(&g, 0, sizeof(g));
      g.a = d_m.as<double>() + (size_t)first_u[u] * D; g.b = d_g.as<double>() + (size_t)u * DD; g.c = d_w.as<double>() + (size_t)first_u[u] * D;
      g.sa_i = D; g.sa_k = 1; g.sb_k = 1; g.sb_j = D;
      g.m = (int)count_u[u]; g.n = D; g.k = D; g.ldc = D; g.alpha = sizes_u[u]; g.beta = 0.0;
      TRY((gemm64<double, double>(g, 1, partials, partial_cap, s)));
    }
    hipLaunchKernelGGL(sub_rows_kernel, dim3(kd_blocks), dim3(256), 0, s, d_m.as<double>(), d_w.as<double>(), (const double *)nullptr, (long long)K, D, d_mw.as<double>());
    // between_stats = sum_u K_u mix_u + W^T W;  within_stats = scatter + sum_u n_u K_u mix_u + MW^T diag(n_k) MW
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(dd_blocks), dim3(256), 0, s, (const double *)nullptr, d_mix.as<double>(), d_count_u.as<double>(), U, DD, d_between.as<double>());
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(dd_blocks), dim3(256), 0, s, d_scatter.as<double>(), d_mix.as<double>(), d_nk_u.as<double>(), U, DD, d_within.as<double>());
    {
      Gemm64Params g; memset(&g, 0, sizeof(g));
      g.a = d_w.p; g.b = d_w.p; g.c = d_between.as<double>(); g.sa_i = 1; g.sa_k = D; g.sb_k = D; g.sb_j = 1;
      g.m = D; g.n = D; g.k = K; g.ldc = D; g.alpha = 1.0; g.beta = 1.0;
      TRY((gemm64<double, double>(g, 1, partials, partial_cap, s)));
      g.a = d_mw.p; g.b = d_mw.p; g.c = d_within.as<double>(); g.kscale = d_sizes_k.as<double>();
      TRY((gemm64<double, double>(g, 1, partials, partial_cap, s)));
    }
    // counts: within (N - K) + K = N, between K (every class weighs 1)
    hipLaunchKernelGGL(scale_kernel, dim3(dd_blocks), dim3(256), 0, s, d_within.as<double>(), DD, 1.0 / (double)n_rows);
    hipLaunchKernelGGL(scale_kernel, dim3(dd_blocks), dim3(256), 0, s, d_between.as<double>(), DD, 1.0 / (double)K);
    ASV_HIP_CHECK(hipGetLastError());
  }
  int status = 0;
  ASV_HIP_CHECK(hipMemcpyAsync(&status, d_status.p, 4, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemcpyAsync(mean_out, d_gmean.p, (size_t)D * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemcpyAsync(within_out, d_within.p, DD * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemcpyAsync(between_out, d_between.p, DD * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  ASV_REQUIRE(status == 0, "asv_plda_train: a covariance lost positive definiteness (non-positive pivot) - too few examples for %d dimensions?", D);
#undef TRY
  return ASV_OK;
}


// Sum and second moment of a set of vectors in float64: what PldaUnsupervisedAdaptor.add_stats accumulates vector by vector
// (plda_base.py:360-367) and the covariance ZCA whitening starts from (score/whiten/train_ZCA_Whitening.py:46-47).
extern "C" int asv_scatter_f64(const float *x, int ldx, int n_rows, int dim, double *sum_out, double *xtx_out, void *stream) {
  ASV_REQUIRE(x && sum_out && xtx_out && n_rows >= 1 && dim >= 1 && ldx >= dim, "asv_scatter_f64: bad argument");
  ASV_ON_OWNER(x, "asv_scatter_f64");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DevBuf d_xtx, d_sum, d_ones, partials;
  size_t partial_cap = 0;
  int rc;
  if ((rc = d_xtx.alloc((size_t)dim * dim * 8)) || (rc = d_sum.alloc((size_t)dim * 8))) return rc;
  Gemm64Params g; memset(&g, 0, sizeof(g));
  g.a = x; g.b = x; g.c = d_xtx.as<double>(); g.sa_i = 1; g.sa_k = ldx; g.sb_k = ldx; g.sb_j = 1;
  g.m = dim; g.n = dim; g.k = n_rows; g.ldc = dim; g.alpha = 1.0; g.beta = 0.0;
  if ((rc = gemm64<float, float>(g, 1, partials, partial_cap, s))) return rc;
  // column sums = X^T 1: the same kernel with a one-column right operand of ones (stride 0)
  if ((rc = d_ones.alloc(8))) return rc;
  const float one = 1.0f;
  ASV_HIP_CHECK(hipMemcpyAsync(d_ones.p, &one, 4, hipMemcpyHostToDevice, s));
  g.b = d_ones.p; g.sb_k = 0; g.sb_j = 0; g.n = 1; g.c = d_sum.as<double>(); g.ldc = 1;
  if ((rc = gemm64<float, float>(g, 1, partials, partial_cap, s))) return rc;
  ASV_HIP_CHECK(hipMemcpyAsync(sum_out, d_sum.p, (size_t)dim * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipMemcpyAsync(xtx_out, d_xtx.p, (size_t)dim * dim * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  return ASV_OK;
}


// Class-level second moments: sum_k n_k mu_k mu_k^T (mu_k = mean of class k) next to the global sum and X^T X - the
// CovarianceStats of Kaldi's ivector-compute-lda, which score/process.sh:218-229 calls for the LDA stage of the scoring chain.
extern "C" int asv_class_scatter_f64(const float *x, int ldx, int n_rows, int dim, const int *order, const long long *class_offsets, int n_classes,
                                     double *sum_out, double *xtx_out, double *class_scatter_out, void *stream) {
  ASV_REQUIRE(x && order && class_offsets && sum_out && xtx_out && class_scatter_out, "asv_class_scatter_f64: null argument");
  ASV_ON_OWNER(x, "asv_class_scatter_f64");
  ASV_REQUIRE(n_rows >= 1 && dim >= 1 && ldx >= dim && n_classes >= 1, "asv_class_scatter_f64: bad sizes");
  ASV_REQUIRE(class_offsets[0] == 0 && class_offsets[n_classes] == n_rows, "asv_class_scatter_f64: class_offsets must run from 0 to n_rows");
  int rc = asv_scatter_f64(x, ldx, n_rows, dim, sum_out, xtx_out, stream);
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int D = dim, K = n_classes;
  std::vector<double> sizes(K);
  for (int k = 0; k < K; ++k) {
    sizes[k] = (double)(class_offsets[k + 1] - class_offsets[k]);
    ASV_REQUIRE(sizes[k] >= 1.0, "asv_class_scatter_f64: class %d is empty", k);
  }
  DevBuf d_order, d_off, d_cmeans, d_sizes, d_out, partials;
  size_t partial_cap = 0;
  if ((rc = d_order.alloc((size_t)n_rows * 4)) || (rc = d_off.alloc(((size_t)K + 1) * 8)) || (rc = d_cmeans.alloc((size_t)K * D * 8)) ||
      (rc = d_sizes.alloc((size_t)K * 8)) || (rc = d_out.alloc((size_t)D * D * 8))) return rc;
  ASV_HIP_CHECK(hipMemcpyAsync(d_order.p, order, (size_t)n_rows * 4, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_off.p, class_offsets, ((size_t)K + 1) * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_sizes.p, sizes.data(), (size_t)K * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  hipLaunchKernelGGL(class_mean_kernel, dim3((unsigned)K, (unsigned)((D + 63) / 64)), dim3(256), 0, s, x, ldx, d_order.as<int>(), d_off.as<long long>(), D, d_cmeans.as<double>());
  Gemm64Params g; memset(&g, 0, sizeof(g));
  g.a = d_cmeans.p; g.b = d_cmeans.p; g.c = d_out.as<double>(); g.sa_i = 1; g.sa_k = D; g.sb_k = D; g.sb_j = 1; g.kscale = d_sizes.as<double>();
  g.m = D; g.n = D; g.k = K; g.ldc = D; g.alpha = 1.0; g.beta = 0.0;
  if ((rc = gemm64<double, double>(g, 1, partials, partial_cap, s))) return rc;
  ASV_HIP_CHECK(hipMemcpyAsync(class_scatter_out, d_out.p, (size_t)D * D * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  return ASV_OK;
}


// Two-covariance PLDA scorer (score/pyplda/gaussian-plda-scoring.py:23-50): see include/asv_amd.h.
extern "C" int asv_two_cov_trials(const float *enroll, int n_enroll, const float *test, int n_test, int dim, const double *gamma, const double *lambda,
                                  const double *c, const int32_t *ei, const int32_t *ti, int n_trials, double *scores, void *stream) {
  ASV_REQUIRE(enroll && test && gamma && lambda && c && ei && ti && scores, "asv_two_cov_trials: null argument");
  ASV_ON_OWNER(enroll, "asv_two_cov_trials");
  ASV_REQUIRE(n_enroll >= 1 && n_test >= 1 && dim >= 1 && dim <= 4096 && n_trials >= 0, "asv_two_cov_trials: bad sizes");
  if (n_trials == 0) return ASV_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t DD = (size_t)dim * dim;
  DevBuf d_g, d_l, d_c, d_el, d_tl, d_eg, d_tg, partials;
  size_t partial_cap = 0;
  int rc;
  if ((rc = d_g.alloc(DD * 8)) || (rc = d_l.alloc(DD * 8)) || (rc = d_c.alloc((size_t)dim * 8)) || (rc = d_el.alloc((size_t)n_enroll * dim * 8)) ||
      (rc = d_eg.alloc((size_t)n_enroll * dim * 8)) || (rc = d_tl.alloc((size_t)n_test * dim * 8)) || (rc = d_tg.alloc((size_t)n_test * dim * 8))) return rc;
  ASV_HIP_CHECK(hipMemcpyAsync(d_g.p, gamma, DD * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_l.p, lambda, DD * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d_c.p, c, (size_t)dim * 8, hipMemcpyHostToDevice, s));
  // rows x matrix products: (V M)[i][j] = sum_k V[i][k] M[k][j]
  auto rows_times = [&](const float *v, int n, const DevBuf &m, DevBuf &out) {
    Gemm64Params g; memset(&g, 0, sizeof(g));
    g.a = v; g.sa_i = dim; g.sa_k = 1; g.b = m.p; g.sb_k = dim; g.sb_j = 1; g.c = out.as<double>();
    g.m = n; g.n = dim; g.k = dim; g.ldc = dim; g.alpha = 1.0; g.beta = 0.0;
    return gemm64<float, double>(g, 1, partials, partial_cap, s);
  };
  // e^T L t = sum_j (sum_i e_i L_ij) t_j = <(E L)[e], t>;   t^T L e = <(T L)[t], e>
  if ((rc = rows_times(enroll, n_enroll, d_l, d_el)) || (rc = rows_times(test, n_test, d_l, d_tl)) || (rc = rows_times(enroll, n_enroll, d_g, d_eg)) ||
      (rc = rows_times(test, n_test, d_g, d_tg))) return rc;
  hipLaunchKernelGGL(two_cov_trials_kernel, dim3((n_trials + 3) / 4), dim3(256), 0, s, enroll, test, dim, d_el.as<double>(), d_tl.as<double>(), d_eg.as<double>(),
                     d_tg.as<double>(), d_c.as<double>(), ei, ti, n_trials, scores);
  ASV_HIP_CHECK(hipGetLastError());
  ASV_HIP_CHECK(hipStreamSynchronize(s));            // the scratch products are freed on return
  return ASV_OK;
}
