// Affine layers of the pooled ("utts") domain: one row per utterance, e.g. the x-vector embedding layer
// tdnn6 = Linear(3000 -> 512) behind statistics pooling (reference model/xvector.py:118-140,
// components.py:107-149 with context [0]) and the two Linear layers of every squeeze-excitation block
// (components.py:600-639, 682-720).  M is a few hundred rows, K up to a few thousand: nothing to tile
// for reuse, everything to gain from spreading K.
//
//   workgroup = 32 utterances x 32 output channels, 8 waves, wave w owns the w-th eighth of K and keeps
//   one 32x32 f32 accumulator; operands go global -> registers -> MFMA (no LDS staging: each byte is used
//   once per workgroup); the 8 partial tiles are summed through LDS in wave order, then the shared epilogue.
//   The K split is a property of the layer (always 8), never of the batch, so an utterance's embedding is
//   bit-identical whatever batch it is extracted in.
//
//   EXACT (f32 precision mode): v_mfma_f32_32x32x2_f32 on the f32 operands - plain f32 fma chains.
//   SPLIT (bf16 precision mode): activations and weights are split x = hi + lo into two bf16 halves and
//   hi*hi + hi*lo + lo*hi run on v_mfma_f32_32x32x16_bf16 (the dropped lo*lo term is ~2^-16 relative): f32-grade
//   results at 5x the rate of the f32 MFMA, so the pooled statistics never see a bf16 rounding.
#include "device_utils.h"

namespace asv {
namespace {

constexpr int UW = 8;                 // waves per workgroup = K slices (16 waves measured 16 us vs 11 us)

// One iteration covers 32 consecutive k.  A wave fetches its [32 rows][32 k] f32 tile with four fully coalesced
// loads (8 lanes = one 128-byte line of a row; reading "lane = row" straight from global costs 64 line
// lookups per instruction and was 5x slower), turns it through a private 36-float-pitch LDS tile (conflict-free
// both ways, no barrier: a wave's LDS operations complete in order) and picks up the MFMA operand layout:
// lane (row lr, half lh) holds k + 16 * lh + 0..15, fed to two MFMA k-steps (elements 0-7, then 8-15).  The bf16
// weight halves are stored in exactly that fragment order ([32-channel fragment][32-k step][j][lane][8]) and
// arrive as contiguous 1 KiB wave loads; the f32 weights of the EXACT variant go through the same LDS turn.
constexpr int TP = 36;                // floats per row of the transposition tile

template <bool SPLIT>
__global__ __launch_bounds__(UW * 64) void utts_gemm_kernel(const TdnnKernelParams p, int m_tiles) {
  // transposition tiles during the K loop, the partial accumulators after it (behind a barrier)
  constexpr int TURN_FLOATS = (SPLIT ? 1 : 2) * 32 * TP;
  static_assert(TURN_FLOATS >= 16 * 64, "the reduction buffer reuses the transposition tiles");
  __shared__ __attribute__((aligned(16))) float turn[UW][TURN_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // every XCD (= its own L2) gets a contiguous run of tiles, utterance tiles fastest: the workgroups that share a
  // 32-channel weight slab sit behind one L2 and the slab leaves HBM once, not once per XCD
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % m_tiles) * 32, n0 = (tile / m_tiles) * 32;

  const int ksteps = (p.cin_pad + 31) / 32;                        // 32-k steps of the layer
  const int spw = (ksteps + UW - 1) / UW;                          // steps per wave
  const int s_begin = wave * spw, s_end = min(s_begin + spw, ksteps);
  const int crow = lane >> 3, ck = (lane & 7) * 4;                 // coalesced fetch: 8 rows x 32 k per instruction
  const float *xg = reinterpret_cast<const float *>(p.x) + (size_t)(m0 + crow) * p.ldx + ck;
  const float *x2g = p.x2 ? reinterpret_cast<const float *>(p.x2) + (size_t)(m0 + crow) * p.ldx2 + ck : nullptr;
  float *tx = turn[wave];

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  struct Raw { float4 v[4]; };
  // branch-free: a lane past the end of K (cin_pad is a multiple of 16, not of 32) re-reads the row's last four
  // floats and drops them with a select - a predicated load costs an exec-mask branch and a vmcnt(0) each
  auto fetch = [&](const float *g, const float *g2, size_t ld, size_t ld2, int k, Raw &r) {
    const bool in = k + ck < p.cin_pad;
    const int kk = in ? k : p.cin_pad - 4 - ck;
#pragma unroll
    for (int q = 0; q < 4; ++q) r.v[q] = *reinterpret_cast<const float4 *>(g + (size_t)(8 * q) * ld + kk);
    if (g2) {
      float4 b[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = *reinterpret_cast<const float4 *>(g2 + (size_t)(8 * q) * ld2 + kk);
#pragma unroll
      for (int q = 0; q < 4; ++q) { r.v[q].x += b[q].x; r.v[q].y += b[q].y; r.v[q].z += b[q].z; r.v[q].w += b[q].w; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) if (!in) r.v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto turn16 = [&](const Raw &r, float *tb, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(tb + (8 * q + crow) * TP + ck) = r.v[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4 *>(tb + lr * TP + 16 * lh + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  };

  if (s_begin < s_end) {
    Raw xr;
    fetch(xg, x2g, p.ldx, p.ldx2, s_begin * 32, xr);
    if constexpr (SPLIT) {
      const size_t frag = (size_t)(n0 / 32) * ksteps;              // 1024 bf16 (hi) per (fragment, step)
      const uint16_t *whi = reinterpret_cast<const uint16_t *>(p.wfrag) + frag * 1024 + lane * 8;
      const uint16_t *wlo = reinterpret_cast<const uint16_t *>(p.wlo) + frag * 1024 + lane * 8;
      struct WFr { uint4 h[2], l[2]; };
      auto fetch_w = [&](int st, WFr &w) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          w.h[j] = *reinterpret_cast<const uint4 *>(whi + (size_t)st * 1024 + j * 512);
          w.l[j] = *reinterpret_cast<const uint4 *>(wlo + (size_t)st * 1024 + j * 512);
        }
      };
      // two steps in flight per wave (the operands come from L2 / HBM with ~1-2 us latency and a step is short):
      // a step consumes its register set, then re-fetches it for the step two ahead; steps past the end re-fetch
      // the last one (no branch, never used)
      auto step = [&](Raw &xq, WFr &wq, int refetch) {
        float xv[16];
        turn16(xq, tx, xv);
        const WFr wc = wq;
        const int nx = min(refetch, s_end - 1);
        fetch(xg, x2g, p.ldx, p.ldx2, nx * 32, xq);
        fetch_w(nx, wq);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          uint32_t xh[4], xl[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v0 = xv[8 * j + 2 * e], v1 = xv[8 * j + 2 * e + 1];
            xh[e] = pack_bf16x2(v0, v1);
            xl[e] = pack_bf16x2(v0 - __uint_as_float(xh[e] << 16), v1 - __uint_as_float(xh[e] & 0xffff0000u));
          }
          const uint4 xhv = make_uint4(xh[0], xh[1], xh[2], xh[3]), xlv = make_uint4(xl[0], xl[1], xl[2], xl[3]);
          // A operand = weights (rows = channels), B operand = utterances (columns); small terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wc.l[j]), __builtin_bit_cast(bf16x8_t, xhv), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wc.h[j]), __builtin_bit_cast(bf16x8_t, xlv), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wc.h[j]), __builtin_bit_cast(bf16x8_t, xhv), acc, 0, 0, 0);
        }
      };
      Raw xr2;
      WFr wa, wb;
      fetch_w(s_begin, wa);
      fetch(xg, x2g, p.ldx, p.ldx2, min(s_begin + 1, s_end - 1) * 32, xr2);
      fetch_w(min(s_begin + 1, s_end - 1), wb);
      for (int st = s_begin; st < s_end; st += 2) {
        step(xr, wa, st + 2);
        if (st + 1 < s_end) step(xr2, wb, st + 3);
      }
    } else {
      const float *wg = reinterpret_cast<const float *>(p.w) + (size_t)(n0 + crow) * p.cin_pad + ck;
      float *tw = tx + 32 * TP;
      Raw wr;
      fetch(wg, nullptr, p.cin_pad, 0, s_begin * 32, wr);
      for (int st = s_begin; st < s_end; ++st) {
        float xv[16], wv[16];
        turn16(xr, tx, xv);
        turn16(wr, tw, wv);
        const int nx = min(st + 1, s_end - 1);
        fetch(xg, x2g, p.ldx, p.ldx2, nx * 32, xr);
        fetch(wg, nullptr, p.cin_pad, 0, nx * 32, wr);
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[e], xv[e], acc, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                    // every wave is done with its transposition tile
#pragma unroll
  for (int r = 0; r < 16; ++r) turn[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  // acc[r] of lane (lh, lr): channel n0 + (r & 3) + 8 * (r >> 2) + 4 * lh, utterance m0 + lr
#pragma unroll
  for (int it = 0; it < 1024 / (UW * 64); ++it) {
    const int idx = it * (UW * 64) + tid, r = idx >> 6, l = idx & 63;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < UW; ++w) s += turn[w][r * 64 + l];
    const int ch = n0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), row = m0 + (l & 31);
    if (ch < p.cout_store) {
      const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
      const float scale = p.scale ? p.scale[ch] : 1.0f, shift = p.shift ? p.shift[ch] : 0.0f;
      const float e = tdnn_epilogue<ET_F32>(p, s, row, ch, p.bias[ch], scale, shift, valid);
      reinterpret_cast<float *>(p.y)[(size_t)row * p.ldy + ch] = e;
      if (p.final_out != nullptr && valid && ch < p.final_ld) {
        // for_extract_embedding's weighted mean over ONE chunk (framework.py:44-52), bit for bit what combine_kernel does
        const float len = (float)p.final_len[row];
        p.final_out[(size_t)row * p.final_ld + ch] = __fdiv_rn(__fadd_rn(0.0f, __fmul_rn(len, e)), len);
      }
    }
  }
}

}  // namespace

// rows_valid: number of utterances; rows beyond it inside the last 32-row tile are written as zeros
int launch_utts_gemm(const TdnnKernelParams &p, int rows_valid, bool split, hipStream_t s) {
  ASV_REQUIRE(p.n_taps == 1 && p.taps[0] == 0, "utts gemm: pooled-domain layers have context [0]");
  ASV_REQUIRE(p.cin_pad % 16 == 0 && p.ldx % 4 == 0 && (p.x2 == nullptr || p.ldx2 % 4 == 0), "utts gemm: K %d / pitch %d not vectorisable", p.cin_pad, p.ldx);
  ASV_REQUIRE(p.seg_bias == nullptr && p.seg_scale == nullptr, "utts gemm: per-segment terms belong to frame-level layers");
  ASV_REQUIRE(!split || (p.wfrag != nullptr && p.wlo != nullptr), "utts gemm: split weights missing");
  const int m_tiles = (std::min(rows_valid, p.rows) + 31) / 32;
  if (m_tiles == 0) return ASV_OK;
  const dim3 grid(m_tiles * ((p.cout_store + 31) / 32)), block(UW * 64);
  if (split) hipLaunchKernelGGL((utts_gemm_kernel<true>), grid, block, 0, s, p, m_tiles);
  else hipLaunchKernelGGL((utts_gemm_kernel<false>), grid, block, 0, s, p, m_tiles);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
