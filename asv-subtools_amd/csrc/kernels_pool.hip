// Row-map, feature packing, statistics / attentive pooling, elementwise and chunk-combine
// kernels.  All of these are HBM-bound streaming kernels: 16-byte loads per lane, lanes
// along the channel axis (the contiguous one), wavefront (DPP shuffle) reduction across the
// lanes that share a channel group, LDS reduction across the 4 waves of a workgroup.
#include "device_utils.h"

namespace asv {
namespace {

// ---------------------------------------------------------------------------------------
// row map: for every padded row, which segment it belongs to (-1: gap) + a validity bitmask
__global__ __launch_bounds__(256) void rowmap_kernel(const int32_t *seg_row0, const int32_t *seg_len, int segments,
                                                     int rows, int pitch, int width, int32_t *row_seg, uint32_t *row_valid) {
  const int row = blockIdx.x * 256 + threadIdx.x;      // rows is a multiple of 128; grid covers it in 64-row waves
  int seg = -1;
  if (row < rows) {
    int lo = 0, hi = segments;                         // largest s with seg_row0[s] <= row
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_row0[mid] <= row) lo = mid; else hi = mid;
    }
    if (segments > 0 && seg_row0[lo] <= row && row < seg_row0[lo] + seg_len[lo] && (row - seg_row0[lo]) % pitch < width) seg = lo;
    row_seg[row] = seg;
  }
  const unsigned long long b = __ballot(seg >= 0);
  if ((threadIdx.x & 63) == 0 && row < rows) {
    row_valid[row >> 5] = (uint32_t)b;
    if (row + 32 < rows) row_valid[(row >> 5) + 1] = (uint32_t)(b >> 32);
  }
}

// ---------------------------------------------------------------------------------------
// Kaldi float matrix [T_total][D] f32 (packed utterances) -> padded row layout, activation type
template <int ET>
__global__ __launch_bounds__(256) void pack_input_kernel(const float *feats, int feat_dim, const int32_t *seg_src0,
                                                         const int32_t *seg_row0, const int32_t *row_seg, int rows,
                                                         void *x, int ldx) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  const int pieces = ldx / VEC;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * pieces) return;
  const int row = (int)(gid / pieces), ch0 = (int)(gid % pieces) * VEC;
  const int seg = row_seg[row];
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = 0.0f;
  if (seg >= 0) {
    const float *src = feats + (size_t)(seg_src0[seg] + (row - seg_row0[seg])) * feat_dim;
    if ((feat_dim & 3) == 0) {
      // Kaldi matrices with a feature dimension that is a multiple of 4 (80, 40, 64 ...): whole 16-byte pieces
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q)
        if (ch0 + 4 * q < feat_dim) {
          const float4 t = *reinterpret_cast<const float4 *>(src + ch0 + 4 * q);
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        if (ch0 + i < feat_dim) v[i] = src[ch0 + i];
    }
  }
  if constexpr (ET != ET_F32) {
    uint4 o;
    o.x = pack_h16x2<ET>(v[0], v[1]); o.y = pack_h16x2<ET>(v[2], v[3]);
    o.z = pack_h16x2<ET>(v[4], v[5]); o.w = pack_h16x2<ET>(v[6], v[7]);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(x) + (size_t)row * ldx + ch0) = o;
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(x) + (size_t)row * ldx + ch0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// padded row layout -> packed [T_total][channels] f32 (single-layer entry point / tests)
template <int ET>
__global__ __launch_bounds__(256) void unpack_rows_kernel(const void *y, int ldy, int channels, const int32_t *seg_src0,
                                                          const int32_t *seg_row0, const int32_t *row_seg, int rows,
                                                          float *out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * channels) return;
  const int row = (int)(gid / channels), ch = (int)(gid % channels);
  const int seg = row_seg[row];
  if (seg < 0) return;
  out[(size_t)(seg_src0[seg] + (row - seg_row0[seg])) * channels + ch] = load_elem<ET>(y, (size_t)row * ldy + ch);
}

// ---------------------------------------------------------------------------------------
// shared reduction helper: VEC partial sums per lane -> full sum over the 4 waves' row slots,
// returned to every thread of the block (indexed by its channel group).
template <int VEC, int CG>
__device__ __forceinline__ void block_reduce_rows(float (&v)[VEC], float (*sm)[64], int wave, int cg, int rs) {
#pragma unroll
  for (int off = CG; off < 64; off <<= 1)
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] += __shfl_xor(v[i], off);
  __syncthreads();                                    // previous use of sm finished
  if (rs == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sm[wave][cg * VEC + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    v[i] = (sm[0][cg * VEC + i] + sm[1][cg * VEC + i]) + (sm[2][cg * VEC + i] + sm[3][cg * VEC + i]);
}

template <int VEC, int CG>
__device__ __forceinline__ void block_max_rows(float (&v)[VEC], float (*sm)[64], int wave, int cg, int rs) {
#pragma unroll
  for (int off = CG; off < 64; off <<= 1)
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = fmaxf(v[i], __shfl_xor(v[i], off));
  __syncthreads();
  if (rs == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) sm[wave][cg * VEC + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    v[i] = fmaxf(fmaxf(sm[0][cg * VEC + i], sm[1][cg * VEC + i]), fmaxf(sm[2][cg * VEC + i], sm[3][cg * VEC + i]));
}

template <int ET, int VEC>
__device__ __forceinline__ void load_vec(const void *base, size_t idx, float (&v)[VEC]) {
  if constexpr (ET != ET_F32) {
    const uint4 u = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(base) + idx);
    unpack_h16x8<ET>(u, v);
  } else {
    const float4 f = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + idx);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
}

// StatisticsPooling (reference libs/nnet/pooling.py:58-67): mean and variance over the segment's
// frames in ONE pass over HBM.  The reference is two-pass (mean, then sum (x-mean)^2); here each
// channel accumulates sum(x-c) and sum((x-c)^2) about the pivot c = the channel's value in the
// segment's first frame, which keeps the one-pass form  var = E[(x-c)^2] - (E[x-c])^2  well
// conditioned (the pivot is within a few std of the mean), then std = sqrt(max(var, eps)) or
// sqrt(var + eps).  grid = (ceil(C/64), segments).
// Tried and dropped (r2c): several workgroups per (segment, 64 channels) for the few, long segments of the ResNet trunk's SE
// means, meeting through a ticket - the device-scope fence in front of the ticket writes the whole L2 back (the producing
// convolution's output is still dirty in it): 220 us instead of 110 us, and a split count that depends on the batch breaks
// the bit-for-bit batch invariance of an utterance's embedding.
// NARROW (<= 32 channels in bf16: the SE means of the ResNet trunk's first stage): 4 lanes cover the channels and a wave
// takes 16 rows per step instead of leaving half of its lanes idle (the row order per lane changes with the channel count
// only - an utterance's result does not depend on the batch).
template <int ET, bool NARROW>
__global__ __launch_bounds__(256) void stats_pool_kernel(const PoolKernelParams p) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  constexpr int CG = NARROW ? 4 : 64 / VEC;          // lanes along channels
  constexpr int RS = 64 / CG;           // row slots per wave
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = lane % CG, rs = lane / CG;
  const int seg = blockIdx.y / p.groups, grp = blockIdx.y % p.groups, ch = blockIdx.x * (CG * VEC) + cg * VEC;
  const int row0 = p.seg_row0[seg] + grp, len = p.seg_len[seg] / p.row_stride;      // rows row0 + k*row_stride, k < len
  const size_t rstep = (size_t)p.row_stride * p.ldx;
  const bool active = ch < round_up_dev(p.channels, kChanAlign);

  float pivot[VEC], s[VEC], q[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { pivot[i] = 0.0f; s[i] = 0.0f; q[i] = 0.0f; }
  if (active) {
    load_vec<ET, VEC>(p.x, (size_t)row0 * p.ldx + ch, pivot);
    // 4 independent row streams per lane keep enough 16-byte loads in flight
    int r = wave * RS + rs;
    for (; r + 3 * 4 * RS < len; r += 4 * 4 * RS) {
      float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
      const size_t base = (size_t)row0 * p.ldx + ch;
      load_vec<ET, VEC>(p.x, base + (size_t)r * rstep, v0);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 4 * RS) * rstep, v1);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 8 * RS) * rstep, v2);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 12 * RS) * rstep, v3);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float d0 = v0[i] - pivot[i], d1 = v1[i] - pivot[i], d2 = v2[i] - pivot[i], d3 = v3[i] - pivot[i];
        s[i] += (d0 + d1) + (d2 + d3);
        q[i] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    }
    for (; r < len; r += 4 * RS) {
      float v[VEC];
      load_vec<ET, VEC>(p.x, (size_t)row0 * p.ldx + ch + (size_t)r * rstep, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { const float d = v[i] - pivot[i]; s[i] += d; q[i] += d * d; }
    }
  }
  block_reduce_rows<VEC, CG>(s, sm, wave, cg, rs);
  if (p.stddev) block_reduce_rows<VEC, CG>(q, sm, wave, cg, rs);
  if (wave == 0 && rs == 0 && active) {
    const float n = (float)len;
    float counts = n;
    if (p.unbiased == 1 && len > 1) counts = (float)(len - 1);      // pooling.py:63-64
    if (p.unbiased == 2) counts = (float)(len - 1);                 // torch.var default (ECAPA), NaN at len 1 like torch
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (ch + i >= p.channels) continue;
      const float dmean = s[i] / n;
      const size_t ocol = (size_t)grp * p.channels * (p.stddev ? 2 : 1);
      p.out[(size_t)seg * p.ld_out + ocol + ch + i] = pivot[i] + dmean;
      if (p.stddev) {
        // sum (x-mean)^2 = sum (x-c)^2 - n (mean-c)^2
        const float var = fmaxf(q[i] - n * dmean * dmean, 0.0f) / counts;
        const float sd = (p.var_mode == ASV_POOL_VAR_ADD) ? sqrtf(var + p.eps) : sqrtf(fmaxf(var, p.eps));
        p.out[(size_t)seg * p.ld_out + ocol + p.channels + ch + i] = sd;
      }
    }
  }
}

// Mean over the rows of a LONG segment (the SE squeeze of the 2-D trunk: 16 k rows x 32 channels per utterance) in two steps.  One
// workgroup per (segment, 64 channels) - stats_pool_kernel - leaves a 256-utterance batch with 256 workgroups of four waves: 4.1 TB/s.
// Here every kPoolChunkRows rows of a segment get a workgroup (grid.z), chunked from the SEGMENT'S first row: which rows are summed
// together, and in which order, depends on the utterance alone - bit-for-bit batch invariance holds (the round-2 attempt split by a
// batch-dependent count and met through a ticket behind a device-scope fence: slower and not invariant).  Plain sums: the pivot of
// the one-pass variance is not needed for a mean.
template <int ET, bool NARROW>
__global__ __launch_bounds__(256) void sum_chunk_kernel(const PoolKernelParams p) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  constexpr int CG = NARROW ? 4 : 64 / VEC;
  constexpr int RS = 64 / CG;
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = lane % CG, rs = lane / CG;
  const int seg = blockIdx.y, chunk = blockIdx.z, ch = blockIdx.x * (CG * VEC) + cg * VEC;
  const int len = p.seg_len[seg];
  const int r_begin = chunk * kPoolChunkRows, r_end = min(len, r_begin + kPoolChunkRows);
  if (r_begin >= len) return;                                   // (uniform for the workgroup; no partial is read for it)
  const size_t base = (size_t)p.seg_row0[seg] * p.ldx + ch;
  const bool active = ch < round_up_dev(p.channels, kChanAlign);
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.0f;
  if (active) {
    int r = r_begin + wave * RS + rs;
    for (; r + 3 * 4 * RS < r_end; r += 4 * 4 * RS) {           // 4 independent row streams per lane
      float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
      load_vec<ET, VEC>(p.x, base + (size_t)r * p.ldx, v0);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 4 * RS) * p.ldx, v1);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 8 * RS) * p.ldx, v2);
      load_vec<ET, VEC>(p.x, base + (size_t)(r + 12 * RS) * p.ldx, v3);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += (v0[i] + v1[i]) + (v2[i] + v3[i]);
    }
    for (; r < r_end; r += 4 * RS) {
      float v[VEC];
      load_vec<ET, VEC>(p.x, base + (size_t)r * p.ldx, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s[i] += v[i];
    }
  }
  block_reduce_rows<VEC, CG>(s, sm, wave, cg, rs);
  if (wave == 0 && rs == 0 && active) {
    float *dst = p.chunk_partial + ((size_t)seg * p.chunks + chunk) * p.ld_chunk + ch;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (ch + i < p.channels) dst[i] = s[i];
  }
}

__global__ __launch_bounds__(64) void sum_chunk_finish_kernel(const PoolKernelParams p) {
  const int seg = blockIdx.y, ch = blockIdx.x * 64 + threadIdx.x;
  if (ch >= p.channels) return;
  const int len = p.seg_len[seg];
  const int n = (len + kPoolChunkRows - 1) / kPoolChunkRows;
  float s = 0.0f;
  for (int k = 0; k < n; ++k) s += p.chunk_partial[((size_t)seg * p.chunks + k) * p.ld_chunk + ch];
  p.out[(size_t)seg * p.ld_out + ch] = s / (float)len;
}

// Fused-pooling finish (see the POOL epilogue of kernels_tdnn_v3.hip): one wave per (segment, 64 channels).
__global__ __launch_bounds__(64) void pool_finish_kernel(const PoolFinishParams p) {
  const int seg = blockIdx.y, ch = blockIdx.x * 64 + threadIdx.x;
  if (ch >= p.channels) return;
  const int row0 = p.seg_row0[seg], len = p.seg_len[seg];
  // merge (count, mean, M2) of the segment's half tiles in row order (Chan et al. pairwise update: sums of non-negative terms
  // and one difference of nearby means - well conditioned in f32; the float64 version of this loop was 20 of the x-vector
  // step's 700 us, most of it double-precision divisions); every partial holds sum (u - pv), sum (u - pv)^2 about its own pivot pv
  float n_acc = 0.0f, mean = 0.0f, m2 = 0.0f;
  const int ts = p.tile_shift ? p.tile_shift : 7;       // rows per partial: 128, or 64 (f32x chain)
  // blocks of 2^ts rows up to row `rows_shift`, blocks of `tail_rows` rows behind it (the 16-bit chain's 96-frame tiles for the last
  // round of workgroups, ChainTilePlan; tail_rows = 0: uniform blocks.  rows_shift may be 0: a batch of less than one round)
  const bool mixed = p.tail_rows > 0;
  const int r_shift = mixed ? p.rows_shift : 0x7fffffff;
  auto block_of = [&](int row) { return row < r_shift ? row >> ts : p.n_shift + (row - r_shift) / p.tail_rows; };
  auto block_row0 = [&](int h) { return (mixed && h >= p.n_shift) ? r_shift + (h - p.n_shift) * p.tail_rows : h << ts; };
  auto block_rows = [&](int h) { return (mixed && h >= p.n_shift) ? p.tail_rows : 1 << ts; };
  for (int h = block_of(row0); h <= block_of(row0 + len - 1); ++h) {
    const int h0 = block_row0(h);
    int first = -1;
    for (int k = 0; k < kHalo + 1 && first < 0; ++k)
      if (h0 + k < p.rows) first = p.row_seg[h0 + k];
    const int slot = seg - first;                       // segments are consecutive in row order
    const int a = max(row0, h0), b = min(row0 + len, h0 + block_rows(h));
    const int parts = p.lh_split ? 2 : 1;
    for (int part = 0; part < parts; ++part) {
      int cnt = b - a;
      if (p.lh_split) {                                 // this part holds the rows whose index has bit 2 == part
        // rows below x with bit 2 set: 4 per complete group of 8 + what the started group holds beyond its first 4
        // (closed form: the row-by-row count this replaces made the kernel 96 us for 640 utterances, r2a profile)
        const int hi_b = (b >> 3) * 4 + max((b & 7) - 4, 0), hi_a = (a >> 3) * 4 + max((a & 7) - 4, 0);
        cnt = part ? hi_b - hi_a : (b - a) - (hi_b - hi_a);
      }
      if (cnt == 0) continue;
      const float *src = p.partial + ((size_t)((h * p.pool_slots + slot) * parts + part) * 3) * p.ld_partial + ch;
      const float nh = (float)cnt, inv_nh = 1.0f / nh;
      const float sh = src[0], qh = src[p.ld_partial], pv = src[2 * p.ld_partial];
      const float dm = sh * inv_nh;                                     // mean of the part minus its pivot
      const float mean_h = pv + dm, m2_h = fmaxf(qh - sh * dm, 0.0f);
      const float tot = n_acc + nh, inv_tot = 1.0f / tot, delta = mean_h - mean;
      mean += delta * nh * inv_tot;
      m2 += m2_h + delta * delta * n_acc * nh * inv_tot;
      n_acc = tot;
    }
  }
  const float n = (float)len;
  float counts = n;
  if (p.unbiased == 1 && len > 1) counts = (float)(len - 1);
  if (p.unbiased == 2) counts = (float)(len - 1);
  p.out[(size_t)seg * p.ld_out + ch] = mean + (p.shift ? p.shift[ch] : 0.0f);
  if (p.stddev) {
    const float var = m2 / counts;
    p.out[(size_t)seg * p.ld_out + p.channels + ch] = (p.var_mode == ASV_POOL_VAR_ADD) ? sqrtf(var + p.eps) : sqrtf(fmaxf(var, p.eps));
  }
}

// ECAPA attentive statistics (ecapa_tdnn_xvector.py:182-188; group 1: a logit per channel) and the shared-weight heads of
// libs/nnet/pooling.py:322-587: every `group` consecutive channels use one logit column (group = channels: the single head of
// AttentiveStatisticsPooling, column 0; group = channels / heads: MultiHeadAttentionPooling).
template <int ET>
__global__ __launch_bounds__(256) void attentive_pool_kernel(const void *x, int ldx, const void *logits, int ldl,
                                                             int channels, const int32_t *seg_row0,
                                                             const int32_t *seg_len, float eps, float *out, int ld_out, int group,
                                                             int softplus2, const float *prior_logit, const float *prior_value) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  constexpr int CG = 64 / VEC;
  constexpr int RS = 64 / CG;
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = lane % CG, rs = lane / CG;
  const int seg = blockIdx.y, ch = blockIdx.x * 64 + cg * VEC;
  const int row0 = seg_row0[seg], len = seg_len[seg];
  const bool active = ch < round_up_dev(channels, kChanAlign);

  // the logits of frame r for this lane's VEC channels
  auto load_logits = [&](int r, float (&e)[VEC]) {
    if (group >= channels) {                                  // one head: column 0 weights every channel
      const float e0 = load_elem<ET>(logits, (size_t)(row0 + r) * ldl);
#pragma unroll
      for (int i = 0; i < VEC; ++i) e[i] = e0;
    } else if (group > 1) {                                   // heads over channel groups: column (channel / group)
#pragma unroll
      for (int i = 0; i < VEC; ++i) e[i] = load_elem<ET>(logits, (size_t)(row0 + r) * ldl + min(ch + i, channels - 1) / group);
    } else {
      load_vec<ET, VEC>(logits, (size_t)(row0 + r) * ldl + ch, e);
      if (softplus2) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) e[i] = 2.0f * logf(e[i] > 20.0f ? e[i] : log1pf(expf(e[i])));   // Softplus(beta 1, threshold 20), squared, log
      }
    }
  };
  float mx[VEC], se[VEC], sx[VEC], sxx[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { mx[i] = -INFINITY; se[i] = 0.0f; sx[i] = 0.0f; sxx[i] = 0.0f; }
  if constexpr (ET != ET_F32) {
    // throughput mode: ONE pass over logits and x (r2: the two passes + libm expf made this kernel 164 us on ECAPA C3).  Every
    // lane keeps a running maximum of its own frames and rescales its sums when it moves (v_exp_f32-based exponentials);
    // the lanes' partial sums are brought to the utterance's maximum before they are added.
    float lm[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) lm[i] = -INFINITY;
    if (active) {
      // two frames per trip: four 16-byte loads in flight per lane, one rescale of the sums for both frames
      int r = wave * RS + rs;
      for (; r + 4 * RS < len; r += 8 * RS) {
        float e0[VEC], v0[VEC], e1[VEC], v1[VEC];
        load_logits(r, e0);
        load_logits(r + 4 * RS, e1);
        load_vec<ET, VEC>(x, (size_t)(row0 + r) * ldx + ch, v0);
        load_vec<ET, VEC>(x, (size_t)(row0 + r + 4 * RS) * ldx + ch, v1);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float mn = fmaxf(lm[i], fmaxf(e0[i], e1[i]));
          const float sc = __expf(lm[i] - mn), w0 = __expf(e0[i] - mn), w1 = __expf(e1[i] - mn);   // first trip: exp(-inf) = 0
          se[i] = fmaf(se[i], sc, w0 + w1);
          sx[i] = fmaf(sx[i], sc, fmaf(w0, v0[i], w1 * v1[i]));
          sxx[i] = fmaf(sxx[i], sc, fmaf(w0 * v0[i], v0[i], w1 * v1[i] * v1[i]));
          lm[i] = mn;
        }
      }
      for (; r < len; r += 4 * RS) {
        float e[VEC], v[VEC];
        load_logits(r, e);
        load_vec<ET, VEC>(x, (size_t)(row0 + r) * ldx + ch, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float mn = fmaxf(lm[i], e[i]);
          const float sc = __expf(lm[i] - mn), w = __expf(e[i] - mn);
          se[i] = fmaf(se[i], sc, w);
          sx[i] = fmaf(sx[i], sc, w * v[i]);
          sxx[i] = fmaf(sxx[i], sc, w * v[i] * v[i]);
          lm[i] = mn;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) mx[i] = lm[i];
    block_max_rows<VEC, CG>(mx, sm, wave, cg, rs);
    if (prior_logit && active) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) mx[i] = fmaxf(mx[i], prior_logit[min(ch + i, channels - 1)]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float f = lm[i] == -INFINITY ? 0.0f : __expf(lm[i] - mx[i]);   // a lane without frames contributes nothing
      se[i] *= f; sx[i] *= f; sxx[i] *= f;
    }
  } else {
    // parity modes: the reference's own two steps (softmax over the frames, then the weighted moments), libm exponentials
    if (active)
      for (int r = wave * RS + rs; r < len; r += 4 * RS) {
        float e[VEC];
        load_logits(r, e);
#pragma unroll
        for (int i = 0; i < VEC; ++i) mx[i] = fmaxf(mx[i], e[i]);
      }
    block_max_rows<VEC, CG>(mx, sm, wave, cg, rs);
    if (prior_logit && active) {                                 // the prior is one more frame of every utterance
#pragma unroll
      for (int i = 0; i < VEC; ++i) mx[i] = fmaxf(mx[i], prior_logit[min(ch + i, channels - 1)]);
    }
    if (active)
      for (int r = wave * RS + rs; r < len; r += 4 * RS) {
        float e[VEC], v[VEC];
        load_logits(r, e);
        load_vec<ET, VEC>(x, (size_t)(row0 + r) * ldx + ch, v);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const float w = expf(e[i] - mx[i]);
          se[i] += w; sx[i] += w * v[i]; sxx[i] += w * v[i] * v[i];
        }
      }
  }
  block_reduce_rows<VEC, CG>(se, sm, wave, cg, rs);
  block_reduce_rows<VEC, CG>(sx, sm, wave, cg, rs);
  block_reduce_rows<VEC, CG>(sxx, sm, wave, cg, rs);
  if (wave == 0 && rs == 0 && active) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (ch + i >= channels) continue;
      if (prior_logit) {
        const float w = expf(prior_logit[ch + i] - mx[i]), v = prior_value[ch + i];
        se[i] += w; sx[i] += w * v; sxx[i] += w * v * v;
      }
      const float mean = sx[i] / se[i];
      // no fma contraction: with one frame (alpha = 1) x^2 - x * x must cancel exactly, as it does in the reference
      float m2 = mean * mean;
      asm volatile("" : "+v"(m2));                            // (HIP's __fmul_rn is a plain multiply and would be contracted)
      const float resid = sxx[i] / se[i] - m2;
      out[(size_t)seg * ld_out + ch + i] = mean;
      out[(size_t)seg * ld_out + channels + ch + i] = sqrtf(fmaxf(resid, eps));
    }
  }
}


// ---------------------------------------------------------------------------------------
// Learnable dictionary encoding pooling (libs/nnet/pooling.py:130-162).
// lde_weights_kernel: 8 frame rows per workgroup staged in LDS; thread (k = centre, s = channel slice) accumulates the
// squared distances |x_t - mu_k|^2 of its slice for the 8 rows (one mu load feeds 8 rows), the slices are summed through
// LDS, wave 0 turns every row's n_centres distances into softmax(-beta_k d_k) over the centres (one centre per lane).
constexpr int kLdeRows = 8;
template <int ET>
__global__ __launch_bounds__(256) void lde_weights_kernel(const void *x, int ldx, int channels, int rows, const float *mu, const float *beta, int n_centres,
                                                          float *weights) {
  extern __shared__ float lde_sh[];                      // xs[kLdeRows][channels] | part[4][kLdeRows][64]
  float *xs = lde_sh, *part = lde_sh + (size_t)kLdeRows * channels;
  const int row0 = blockIdx.x * kLdeRows;
  for (int i = threadIdx.x; i < kLdeRows * channels; i += 256) {
    const int f = i / channels, c = i - f * channels;
    xs[i] = row0 + f < rows ? load_elem<ET>(x, (size_t)(row0 + f) * ldx + c) : 0.0f;
  }
  __syncthreads();
  const int k = threadIdx.x & 63, sl = threadIdx.x >> 6;
  float d[kLdeRows];
#pragma unroll
  for (int f = 0; f < kLdeRows; ++f) d[f] = 0.0f;
  if (k < n_centres)
    for (int c = sl; c < channels; c += 4) {
      const float m = mu[(size_t)c * n_centres + k];
#pragma unroll
      for (int f = 0; f < kLdeRows; ++f) { const float r = xs[f * channels + c] - m; d[f] = fmaf(r, r, d[f]); }
    }
#pragma unroll
  for (int f = 0; f < kLdeRows; ++f) part[(sl * kLdeRows + f) * 64 + k] = d[f];
  __syncthreads();
  if (sl == 0) {
    const float b = k < n_centres ? beta[k] : 0.0f;
    for (int f = 0; f < kLdeRows; ++f) {
      if (row0 + f >= rows) break;
      const float dist = part[f * 64 + k] + part[(kLdeRows + f) * 64 + k] + part[(2 * kLdeRows + f) * 64 + k] + part[(3 * kLdeRows + f) * 64 + k];
      const float l = k < n_centres ? -b * dist : -INFINITY;
      float mx = l;
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float e = k < n_centres ? expf(l - mx) : 0.0f;
      float se = e;
      for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
      weights[(size_t)(row0 + f) * 64 + k] = e / se;
    }
  }
}

// lde_accumulate_kernel: one workgroup per (64 channels, segment); lane = channel, the 4 waves split the frames;
// acc[k] += w[t][k] x[t][c] and s0[k] += w[t][k] for KMAX >= n_centres centres (zero weights beyond), summed over the
// waves through LDS; out[c * n_centres + k] = (acc[k] - mu[c][k] s0[k]) / frames.
template <int ET, int KMAX>
__global__ __launch_bounds__(256) void lde_accumulate_kernel(const void *x, int ldx, int channels, const float *weights, const float *mu, int n_centres,
                                                             const int32_t *seg_row0, const int32_t *seg_len, float *out, int ld_out) {
  __shared__ float red[4][2 * KMAX][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int seg = blockIdx.y, c = blockIdx.x * 64 + lane;
  const int row0 = seg_row0[seg], len = seg_len[seg];
  float acc[KMAX], s0[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { acc[k] = 0.0f; s0[k] = 0.0f; }
  for (int r = wave; r < len; r += 4) {
    const float v = c < channels ? load_elem<ET>(x, (size_t)(row0 + r) * ldx + c) : 0.0f;
    const float *w = weights + (size_t)(row0 + r) * 64;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { const float wk = w[k]; acc[k] = fmaf(wk, v, acc[k]); s0[k] += wk; }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { red[wave][k][lane] = acc[k]; red[wave][KMAX + k][lane] = s0[k]; }
  __syncthreads();
  if (wave == 0 && c < channels) {
    const float inv = 1.0f / (float)len;
    for (int k = 0; k < n_centres; ++k) {
      const float a = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
      const float z = red[0][KMAX + k][lane] + red[1][KMAX + k][lane] + red[2][KMAX + k][lane] + red[3][KMAX + k][lane];
      out[(size_t)seg * ld_out + (size_t)c * n_centres + k] = (a - mu[(size_t)c * n_centres + k] * z) * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------
// elementwise: out = (a [*scale+shift]) (* seg_scale[seg]) (+ b) (+ c); gap rows -> 0
template <int ET>
__global__ __launch_bounds__(256) void eltwise_kernel(const EltwiseKernelParams p) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  const int pieces = round_up_dev(p.channels, VEC) / VEC;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)p.rows * pieces) return;
  const int row = (int)(gid / pieces), ch = (int)(gid % pieces) * VEC;
  float o[VEC], tb[VEC];
  bool valid = true;
  int seg = row;
  if (p.row_valid != nullptr) {
    valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
    seg = valid ? p.row_seg[row] : 0;
  }
  if (valid) {
    load_vec<ET, VEC>(p.a, (size_t)row * p.lda + ch, o);
    // per-channel / per-segment f32 tables: 16-byte loads when the whole piece is inside the channel range (all
    // table pitches and channel offsets are multiples of 4 floats), element-wise only for the ragged last piece
    const bool whole = ch + VEC <= p.channels;
    auto table = [&](const float *t, float (&v)[VEC]) {
      if (whole) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
          const float4 f = *reinterpret_cast<const float4 *>(t + ch + 4 * q);
          v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = (ch + i < p.channels) ? t[ch + i] : 0.0f;
      }
    };
    if (p.scale != nullptr) {
      float sc[VEC], sh[VEC];
      table(p.scale, sc); table(p.shift, sh);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = o[i] * sc[i] + sh[i];
    }
    if (p.seg_norm != nullptr) {
      const float *st = p.seg_norm + (size_t)seg * p.ld_segnorm;
      if (p.seg_norm_mode & 1) {
        float m[VEC];
        table(st, m);
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] -= m[i];
      }
      if (p.seg_norm_mode & 2) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (ch + i < p.channels) o[i] /= st[p.channels + ch + i];
      }
    }
    if (p.seg_scale != nullptr) {
      float ss[VEC];
      table(p.seg_scale + (size_t)seg * p.ld_segscale, ss);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = __fmul_rn(o[i], ss[i]);          // (separately rounded, like the reference's x * s + residual:
    }                                                                         //  never contracted into one fma - im2col_kernel's prologue repeats it)
    if (p.b != nullptr) {
      load_vec<ET, VEC>(p.b, (size_t)row * p.ldb + ch, tb);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = __fadd_rn(o[i], tb[i]);
    }
    if (p.c != nullptr) {
      float t[VEC];
      load_vec<ET, VEC>(p.c, (size_t)row * p.ldc + ch, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = (ch + i >= p.channels) ? 0.0f : apply_act(o[i], p.act);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = 0.0f;
  }
  if constexpr (ET != ET_F32) {
    uint4 u;
    u.x = pack_h16x2<ET>(o[0], o[1]); u.y = pack_h16x2<ET>(o[2], o[3]);
    u.z = pack_h16x2<ET>(o[4], o[5]); u.w = pack_h16x2<ET>(o[6], o[7]);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.out) + (size_t)row * p.ldo + ch) = u;
    if (p.out2 != nullptr) {
      // the value a separate addition would read back: the bf16 just stored
      unpack_h16x8<ET>(u, o);
    }
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out) + (size_t)row * p.ldo + ch) = make_float4(o[0], o[1], o[2], o[3]);
  }
  if (p.out2 != nullptr) {
    if (valid) {
      float t[VEC];
      if (p.d == p.b && p.ldd == p.ldb) {          // the addend is the residual already loaded (ECAPA: next input = residual + block output)
#pragma unroll
        for (int i = 0; i < VEC; ++i) t[i] = tb[i];
      } else {
        load_vec<ET, VEC>(p.d, (size_t)row * p.ldd + ch, t);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = (ch + i >= p.channels) ? 0.0f : o[i] + t[i];
    }
    if constexpr (ET != ET_F32) {
      uint4 u;
      u.x = pack_h16x2<ET>(o[0], o[1]); u.y = pack_h16x2<ET>(o[2], o[3]);
      u.z = pack_h16x2<ET>(o[4], o[5]); u.w = pack_h16x2<ET>(o[6], o[7]);
      *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.out2) + (size_t)row * p.ldo2 + ch) = u;
    } else {
      *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.out2) + (size_t)row * p.ldo2 + ch) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// chunk combine (framework.py:38-47): emb = (sum_{i<n-1} len_i*e_i + len_last*e_last) / T,
// separately rounded f32 mul/add like the reference's tensor ops.
__global__ __launch_bounds__(256) void combine_kernel(const float *seg_emb, int ld_seg, const int32_t *utt_seg0,
                                                      const int32_t *utt_nseg, const int32_t *seg_len, int n_utts,
                                                      int embed_dim, float *out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)n_utts * embed_dim) return;
  const int u = (int)(gid / embed_dim), e = (int)(gid % embed_dim);
  const int s0 = utt_seg0[u], n = utt_nseg[u];
  float acc = 0.0f;
  int total = 0;
  for (int i = 0; i < n - 1; ++i) {
    const int len = seg_len[s0 + i];
    acc = __fadd_rn(acc, __fmul_rn((float)len, seg_emb[(size_t)(s0 + i) * ld_seg + e]));
    total += len;
  }
  const int len = seg_len[s0 + n - 1];
  total += len;
  const float last = __fmul_rn((float)len, seg_emb[(size_t)(s0 + n - 1) * ld_seg + e]);
  out[(size_t)u * embed_dim + e] = __fdiv_rn(__fadd_rn(acc, last), (float)total);
}

// ---------------------------------------------------------------------------------------
// 2-D (grid) domains: rows are (time, frequency) positions, frequency fastest, `pitch` rows per frame
// (pitch - width zero rows between frames = the frequency zero padding of a 3x3 convolution).

// frames-domain feature rows [t][f] -> grid rows t*pitch + f with ONE channel (ResNet input unsqueeze)
template <int ET>
__global__ __launch_bounds__(256) void grid_from_frames_kernel(const void *x, int ldx, int feat_dim, const int32_t *fr_row0, const int32_t *g_row0,
                                                               const int32_t *g_row_seg, const uint32_t *g_row_valid, int g_rows, int pitch,
                                                               void *out, int ldo) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= g_rows) return;
  float v = 0.0f;
  if ((g_row_valid[row >> 5] >> (row & 31)) & 1u) {
    const int seg = g_row_seg[row], rel = row - g_row0[seg];
    const int t = rel / pitch, f = rel % pitch;
    if (f < feat_dim) v = load_elem<ET>(x, (size_t)(fr_row0[seg] + t) * ldx + f);
  }
  // channel 0 carries the value, the pad channels of the 16-wide pitch stay zero
  for (int c = 0; c < ldo; ++c) store_elem<ET>(out, (size_t)row * ldo + c, c == 0 ? v : 0.0f);
}

// im2col gather for strided convolutions: one thread per (output row, tap, 16-byte piece).
// Tried and dropped (r2k): one thread per (row, piece) walking the nine taps (the row's four dependent table lookups paid
// once): the stores of a wave then scatter over 16 rows x 64 B instead of covering 576 contiguous bytes per row -
// 598 vs 344 us on the 32 -> 64 stage.
template <int ET>
__global__ __launch_bounds__(256) void im2col_kernel(const Im2colParams p) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  const int pieces = p.channels / VEC;                  // channels is a multiple of 16
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per_row = (long long)p.n_taps * pieces;
  if (gid >= (long long)p.out_rows * per_row) return;
  const int row = (int)(gid / per_row), rem = (int)(gid % per_row), k = rem / pieces, piece = rem % pieces;
  uint4 v = make_uint4(0, 0, 0, 0);
  if ((p.out_row_valid[row >> 5] >> (row & 31)) & 1u) {
    const int seg = p.out_row_seg[row], rel = row - p.out_row0[seg];
    const int t = (rel / p.out_pitch) * p.stride + p.dt[k], f = (rel % p.out_pitch) * p.stride + p.df[k];
    const int in_frames = p.in_len[seg] / p.in_pitch;
    if (t >= 0 && t < in_frames && f >= 0 && f < p.in_width) {
      const size_t srow = (size_t)(p.in_row0[seg] + t * p.in_pitch + f);
      const size_t src = srow * p.ldi + (size_t)piece * VEC;
      if constexpr (ET != ET_F32) v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.in) + src);
      else v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(p.in) + src);
      if (p.b != nullptr || p.seg_scale != nullptr || p.act != ASV_ACT_NONE) {
        // elementwise prologue: exactly eltwise_kernel's operations on this piece (separately rounded multiply and add, then the
        // activation, then the rounding to the element type), so that fusing the pass changes no bit
        float o[VEC];
        if constexpr (ET != ET_F32) unpack_h16x8<ET>(v, o);
        else { o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w); }
        if (p.seg_scale != nullptr) {
          const float *ss = p.seg_scale + (size_t)seg * p.ld_segscale + (size_t)piece * VEC;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] = __fmul_rn(o[i], ss[i]);
        }
        if (p.b != nullptr) {
          float tb[VEC];
          load_vec<ET, VEC>(p.b, srow * p.ldb + (size_t)piece * VEC, tb);
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[i] = __fadd_rn(o[i], tb[i]);
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = apply_act(o[i], p.act);
        if constexpr (ET != ET_F32) v = pack_h16x8<ET>(o);
        else v = make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
      }
    }
  }
  const size_t dst = (size_t)row * p.ldo + (size_t)k * p.channels + (size_t)piece * VEC;
  if constexpr (ET != ET_F32) *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(p.out) + dst) = v;
  else *reinterpret_cast<uint4 *>(reinterpret_cast<float *>(p.out) + dst) = v;
}

// ResNetXvector's [B, C, F', T'] -> [B, C*F', T'] reshape for the frame-weighting poolings: one thread per (output row, 16-byte
// piece) gathers its 8 (4) elements c*F + f from the F consecutive grid rows of the frame - a [F][C] block of a few KiB that the
// other threads of the row read too (L1 / L2 hits).  Once per batch on the smallest map of the trunk: not a hot kernel.
template <int ET>
__global__ __launch_bounds__(256) void grid_flatten_kernel(const void *in, int ldi, int channels, int width, int in_pitch, const int32_t *in_row0,
                                                           const int32_t *out_row0, const int32_t *out_row_seg, const uint32_t *out_row_valid,
                                                           int out_rows, void *out, int ldo) {
  constexpr int VEC = (ET != ET_F32) ? 8 : 4;
  const int pieces = ldo / VEC;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)out_rows * pieces) return;
  const int row = (int)(gid / pieces), j0 = (int)(gid % pieces) * VEC;
  float v[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = 0.0f;
  if ((out_row_valid[row >> 5] >> (row & 31)) & 1u) {
    const int seg = out_row_seg[row];
    const size_t base = (size_t)(in_row0[seg] + (row - out_row0[seg]) * in_pitch);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const int j = j0 + i;
      if (j < channels * width) v[i] = load_elem<ET>(in, (base + j % width) * ldi + j / width);
    }
  }
  if constexpr (ET != ET_F32) {
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(out) + (size_t)row * ldo + j0) = pack_h16x8<ET>(v);
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + (size_t)row * ldo + j0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

int launch_grid_flatten(const void *in, int ldi, int channels, int width, int in_pitch, const int32_t *in_row0, const int32_t *out_row0,
                        const int32_t *out_row_seg, const uint32_t *out_row_valid, int out_rows, void *out, int ldo, int et, hipStream_t s) {
  const int vec = et != ET_F32 ? 8 : 4;
  const long long n = (long long)out_rows * (ldo / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(grid_flatten_kernel<ET_BF16>, grid, block, 0, s, in, ldi, channels, width, in_pitch, in_row0, out_row0, out_row_seg, out_row_valid, out_rows, out, ldo);
  else if (et == ET_F16) hipLaunchKernelGGL(grid_flatten_kernel<ET_F16>, grid, block, 0, s, in, ldi, channels, width, in_pitch, in_row0, out_row0, out_row_seg, out_row_valid, out_rows, out, ldo);
  else hipLaunchKernelGGL(grid_flatten_kernel<ET_F32>, grid, block, 0, s, in, ldi, channels, width, in_pitch, in_row0, out_row0, out_row_seg, out_row_valid, out_rows, out, ldo);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_grid_from_frames(const void *x, int ldx, int feat_dim, const int32_t *fr_row0, const int32_t *g_row0, const int32_t *g_row_seg,
                            const uint32_t *g_row_valid, int g_rows, int pitch, void *out, int ldo, int et, hipStream_t s) {
  const dim3 grid((g_rows + 255) / 256), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(grid_from_frames_kernel<ET_BF16>, grid, block, 0, s, x, ldx, feat_dim, fr_row0, g_row0, g_row_seg, g_row_valid, g_rows, pitch, out, ldo);
  else if (et == ET_F16) hipLaunchKernelGGL(grid_from_frames_kernel<ET_F16>, grid, block, 0, s, x, ldx, feat_dim, fr_row0, g_row0, g_row_seg, g_row_valid, g_rows, pitch, out, ldo);
  else hipLaunchKernelGGL(grid_from_frames_kernel<ET_F32>, grid, block, 0, s, x, ldx, feat_dim, fr_row0, g_row0, g_row_seg, g_row_valid, g_rows, pitch, out, ldo);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_im2col(const Im2colParams &p, int et, hipStream_t s) {
  const int vec = et != ET_F32 ? 8 : 4;
  const long long n = (long long)p.out_rows * p.n_taps * (p.channels / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(im2col_kernel<ET_BF16>, grid, block, 0, s, p);
  else if (et == ET_F16) hipLaunchKernelGGL(im2col_kernel<ET_F16>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(im2col_kernel<ET_F32>, grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_rowmap(const int32_t *seg_row0, const int32_t *seg_len, int segments, int rows, int pitch, int width, int32_t *row_seg,
                  uint32_t *row_valid, hipStream_t s) {
  hipLaunchKernelGGL(rowmap_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, seg_row0, seg_len, segments, rows, pitch, width, row_seg, row_valid);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_pack_input(const float *feats, int feat_dim, const int32_t *seg_src0, const int32_t *seg_row0,
                      const int32_t *row_seg, int rows, void *x, int ldx, int et, hipStream_t s) {
  const int vec = et != ET_F32 ? 8 : 4;
  const long long n = (long long)rows * (ldx / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(pack_input_kernel<ET_BF16>, grid, block, 0, s, feats, feat_dim, seg_src0, seg_row0, row_seg, rows, x, ldx);
  else if (et == ET_F16) hipLaunchKernelGGL(pack_input_kernel<ET_F16>, grid, block, 0, s, feats, feat_dim, seg_src0, seg_row0, row_seg, rows, x, ldx);
  else hipLaunchKernelGGL(pack_input_kernel<ET_F32>, grid, block, 0, s, feats, feat_dim, seg_src0, seg_row0, row_seg, rows, x, ldx);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_unpack_rows(const void *y, int ldy, int channels, const int32_t *seg_src0, const int32_t *seg_row0,
                       const int32_t *row_seg, int rows, float *out, int et, hipStream_t s) {
  const long long n = (long long)rows * channels;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(unpack_rows_kernel<ET_BF16>, grid, block, 0, s, y, ldy, channels, seg_src0, seg_row0, row_seg, rows, out);
  else if (et == ET_F16) hipLaunchKernelGGL(unpack_rows_kernel<ET_F16>, grid, block, 0, s, y, ldy, channels, seg_src0, seg_row0, row_seg, rows, out);
  else hipLaunchKernelGGL(unpack_rows_kernel<ET_F32>, grid, block, 0, s, y, ldy, channels, seg_src0, seg_row0, row_seg, rows, out);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_stats_pool(const PoolKernelParams &p, int segments, int et, hipStream_t s) {
  if (segments <= 0) return ASV_OK;
  if (p.chunks > 0) {
    ASV_REQUIRE(!p.stddev && p.groups == 1 && p.row_stride == 1 && p.chunk_partial != nullptr && p.ld_chunk >= p.channels, "stats_pool: the chunked form is mean-only over contiguous rows");
    const bool narrow = et != ET_F32 && p.channels <= 32;
    const dim3 cgrid(narrow ? 1 : (p.channels + 63) / 64, segments, p.chunks), block(256);
    if (et == ET_BF16 && narrow) hipLaunchKernelGGL((sum_chunk_kernel<ET_BF16, true>), cgrid, block, 0, s, p);
    else if (et == ET_F16 && narrow) hipLaunchKernelGGL((sum_chunk_kernel<ET_F16, true>), cgrid, block, 0, s, p);
    else if (et == ET_BF16) hipLaunchKernelGGL((sum_chunk_kernel<ET_BF16, false>), cgrid, block, 0, s, p);
    else if (et == ET_F16) hipLaunchKernelGGL((sum_chunk_kernel<ET_F16, false>), cgrid, block, 0, s, p);
    else hipLaunchKernelGGL((sum_chunk_kernel<ET_F32, false>), cgrid, block, 0, s, p);
    hipLaunchKernelGGL(sum_chunk_finish_kernel, dim3((p.channels + 63) / 64, segments), dim3(64), 0, s, p);
    ASV_HIP_CHECK(hipGetLastError());
    return ASV_OK;
  }
  const dim3 grid((p.channels + 63) / 64, segments * p.groups), block(256);
  if (et == ET_BF16 && p.channels <= 32) hipLaunchKernelGGL((stats_pool_kernel<ET_BF16, true>), dim3(1, segments * p.groups), block, 0, s, p);
  else if (et == ET_F16 && p.channels <= 32) hipLaunchKernelGGL((stats_pool_kernel<ET_F16, true>), dim3(1, segments * p.groups), block, 0, s, p);
  else if (et == ET_BF16) hipLaunchKernelGGL((stats_pool_kernel<ET_BF16, false>), grid, block, 0, s, p);
  else if (et == ET_F16) hipLaunchKernelGGL((stats_pool_kernel<ET_F16, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((stats_pool_kernel<ET_F32, false>), grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_pool_finish(const PoolFinishParams &p, int segments, hipStream_t s) {
  if (segments <= 0) return ASV_OK;
  hipLaunchKernelGGL(pool_finish_kernel, dim3((p.channels + 63) / 64, segments), dim3(64), 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_attentive_pool(const void *x, int ldx, const void *logits, int ldl, int channels, const int32_t *seg_row0,
                          const int32_t *seg_len, int segments, float eps, float *out, int ld_out, int et, int group, int softplus2, const float *prior_logit, const float *prior_value,
                          hipStream_t s) {
  if (segments <= 0) return ASV_OK;
  const dim3 grid((channels + 63) / 64, segments), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(attentive_pool_kernel<ET_BF16>, grid, block, 0, s, x, ldx, logits, ldl, channels, seg_row0, seg_len, eps, out, ld_out, group, softplus2, prior_logit, prior_value);
  else if (et == ET_F16) hipLaunchKernelGGL(attentive_pool_kernel<ET_F16>, grid, block, 0, s, x, ldx, logits, ldl, channels, seg_row0, seg_len, eps, out, ld_out, group, softplus2, prior_logit, prior_value);
  else hipLaunchKernelGGL(attentive_pool_kernel<ET_F32>, grid, block, 0, s, x, ldx, logits, ldl, channels, seg_row0, seg_len, eps, out, ld_out, group, softplus2, prior_logit, prior_value);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}


int launch_lde_pool(const void *x, int ldx, int channels, int rows, const float *mu, const float *beta, int n_centres, float *weights,
                    const int32_t *seg_row0, const int32_t *seg_len, int segments, float *out, int ld_out, int et, hipStream_t s) {
  if (segments <= 0 || rows <= 0) return ASV_OK;
  const size_t lds = ((size_t)kLdeRows * channels + 4 * kLdeRows * 64) * sizeof(float);
  ASV_REQUIRE(lds <= 64 * 1024, "lde pooling: %d channels need %zu bytes of LDS per workgroup (limit 64 KiB)", channels, lds);
  const dim3 g1((rows + kLdeRows - 1) / kLdeRows), g2((channels + 63) / 64, segments);
  if (et == ET_BF16) hipLaunchKernelGGL(lde_weights_kernel<ET_BF16>, g1, dim3(256), lds, s, x, ldx, channels, rows, mu, beta, n_centres, weights);
  else if (et == ET_F16) hipLaunchKernelGGL(lde_weights_kernel<ET_F16>, g1, dim3(256), lds, s, x, ldx, channels, rows, mu, beta, n_centres, weights);
  else hipLaunchKernelGGL(lde_weights_kernel<ET_F32>, g1, dim3(256), lds, s, x, ldx, channels, rows, mu, beta, n_centres, weights);
#define ASV_LDE_ACC(B, K) hipLaunchKernelGGL((lde_accumulate_kernel<B, K>), g2, dim3(256), 0, s, x, ldx, channels, weights, mu, n_centres, seg_row0, seg_len, out, ld_out)
#define ASV_LDE_ET(K) do { if (et == ET_BF16) ASV_LDE_ACC(ET_BF16, K); else if (et == ET_F16) ASV_LDE_ACC(ET_F16, K); else ASV_LDE_ACC(ET_F32, K); } while (0)
  if (n_centres <= 8) ASV_LDE_ET(8);
  else if (n_centres <= 16) ASV_LDE_ET(16);
  else if (n_centres <= 32) ASV_LDE_ET(32);
  else ASV_LDE_ET(64);
#undef ASV_LDE_ET
#undef ASV_LDE_ACC
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_eltwise(const EltwiseKernelParams &p, int et, hipStream_t s) {
  const int vec = et != ET_F32 ? 8 : 4;
  const long long n = (long long)p.rows * (round_up(p.channels, vec) / vec);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (et == ET_BF16) hipLaunchKernelGGL(eltwise_kernel<ET_BF16>, grid, block, 0, s, p);
  else if (et == ET_F16) hipLaunchKernelGGL(eltwise_kernel<ET_F16>, grid, block, 0, s, p);
  else hipLaunchKernelGGL(eltwise_kernel<ET_F32>, grid, block, 0, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_combine(const float *seg_emb, int ld_seg, const int32_t *utt_seg0, const int32_t *utt_nseg,
                   const int32_t *seg_len, int n_utts, int embed_dim, float *out, hipStream_t s) {
  const long long n = (long long)n_utts * embed_dim;
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seg_emb, ld_seg, utt_seg0, utt_nseg, seg_len, n_utts, embed_dim, out);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
