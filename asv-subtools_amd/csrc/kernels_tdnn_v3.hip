// bf16 TDNN / 1x1-conv implicit GEMM, large-tile variant 3: feature window through LDS,
// weight fragments straight from L2 into registers.
//
// Measured on the variant-2 kernel (tools/gemm_ablate, 52k frames x 512 -> 512, 3 taps): the pure
// MFMA + barrier skeleton needs 71 us, the ds_read_b128 stream alone 91 us, the LDS-DMA stream alone
// 59 us - the kernel was LDS-read bound (6 fragment reads per 8 MFMAs at ~128 B/clk/CU), not MFMA
// or HBM/L2 bound.  This variant removes a third of the LDS reads and most LDS-DMA traffic:
//
//   * weights are pre-packed on the host in MFMA-fragment order
//         Wf[n_frag32][tap][chunk64][k_group16][lane][8 x bf16]
//     so a wave fetches the fragment of one k-group with ONE perfectly coalesced 1 KiB
//     global_load_dwordx4 (the two M-waves that share a channel range hit in L1).  Each fragment
//     register is re-loaded for the NEXT step right after its MFMAs were issued (prefetch distance:
//     one step, no extra registers).
//   * only the feature window (264 frames x 64 channels, shared by all taps of a chunk) goes through
//     LDS, by LDS-DMA into a 3-stage ring (prefetch distance: two chunks).  The window is read-only for
//     the whole chunk, so the tap steps of a chunk run WITHOUT any barrier: one barrier per chunk,
//     and a counted `s_waitcnt vmcnt(12)` that waits for the window of the next chunk but not for the
//     weight fragments / window pieces issued during the last step.
//   * 4 LDS fragment reads per 8 MFMAs; epilogue as in variant 2 (4 consecutive channels per lane,
//     LDS-staged 16-byte stores).
#include "device_utils.h"

namespace asv {
namespace {

constexpr int BN = 256;
constexpr int ROWB = 128;
constexpr int N_STAGES = 3;
constexpr int BK = 64;
static_assert(BN == kBigTileN, "weight padding must match the N tile");
// WM = number of 128-frame wave rows per workgroup:
//   WM = 2: 256 x 256 tile, 8 waves, 128 KiB LDS, one workgroup per CU
//   WM = 1: 128 x 256 tile, 4 waves,  64 KiB LDS, TWO workgroups per CU - the prologue (first window + weight
//           fragments in flight) and the epilogue of one workgroup overlap with the main loop of the other
template <int WM> struct Geom3 {
  static constexpr int BM = 128 * WM;
  static constexpr int WIN = BM + 2 * kHalo;           // 264 | 136
  static constexpr int A_STAGE = WIN * ROWB;           // 33792 | 17408
  static constexpr int A_GROUPS = WIN / 8;             // 33 | 17 eight-row groups
  static constexpr int WAVES = 4 * WM;
  static constexpr int PIECES = (A_GROUPS + WAVES - 1) / WAVES;   // LDS-DMA pieces per wave per window: 5
  static constexpr int LDS_BYTES = WAVES * 16384;      // epilogue scratch: 16 KiB per wave; >= 3 window stages
  static_assert(N_STAGES * A_STAGE <= LDS_BYTES, "window ring must fit");
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// one LDS-DMA instruction (see kernels_tdnn_v2.hip for why this is inline asm)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// ABL: 0 full, 2 MFMA only (no loads of any kind in the loop), 4 no epilogue stores
template <int ABL, bool GENERIC, bool POOL, int WM>
__global__ __launch_bounds__(256 * WM, 2) void tdnn_gemm_big3_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  using G = Geom3<WM>;
  constexpr int BM = G::BM, A_STAGE = G::A_STAGE, A_GROUPS = G::A_GROUPS;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * BM;
  const int n0 = (tile % n_tiles) * BN;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *zero = reinterpret_cast<const unsigned char *>(p.zero16);
  const size_t x_pitch = (size_t)p.ldx * 2;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte *)lds);
  const int g_row = lane >> 3, g_slot = lane & 7;

  const int nchunks = (p.cin_pad + BK - 1) / BK;
  const int n_taps = p.n_taps;

  // feature window of chunk c -> ring stage st (33 eight-row groups: 4 per wave + a 5th for wave 0)
  auto issue_A = [&](int c, int st) {
#pragma unroll
    for (int i = 0; i < G::PIECES; ++i) {
      const int grp = wave + i * G::WAVES;
      if (grp < A_GROUPS) {
        const int w = grp * 8 + g_row;
        const int row = m0 - kHalo + w;
        const int ch = c * BK + swz(w, g_slot) * 8;
        const bool ok = row >= 0 && row < p.rows && ch < p.cin_pad;
        const unsigned char *src = ok ? xg + (size_t)row * x_pitch + (size_t)ch * 2 : zero;
        glds16(src, __builtin_amdgcn_readfirstlane(lds_base + st * A_STAGE + grp * 1024));
      }
    }
  };

  // weight fragments: this wave's two 32-channel fragments, [tap][chunk][k-group] blocks of 1 KiB
  const size_t frag_stride = (size_t)n_taps * nchunks * 4096;            // bytes per 32-channel fragment
  const unsigned char *wf_base0 = reinterpret_cast<const unsigned char *>(p.wfrag) + (size_t)((n0 + wn * 64) / 32) * frag_stride + (size_t)lane * 16;
  const unsigned char *wf_base1 = wf_base0 + frag_stride;
  uint4 wf[4][2];
  auto load_wf = [&](int kg, int c, int t) {
    const size_t off = ((size_t)t * nchunks + c) * 4096 + (size_t)kg * 1024;
    wf[kg][0] = *reinterpret_cast<const uint4 *>(wf_base0 + off);
    wf[kg][1] = *reinterpret_cast<const uint4 *>(wf_base1 + off);
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  struct XFrags { uint4 x[4]; };
  auto load_x = [&](const unsigned char *Ab, int d, int kg, XFrags &f) {
    const int slot = kg * 2 + lh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int w = wm * 128 + i * 32 + lr + kHalo + d;
      f.x[i] = *reinterpret_cast<const uint4 *>(Ab + w * ROWB + swz(w, slot) * 16);
    }
  };
  auto mma = [&](const XFrags &f, int kg) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        // A operand = weights (rows = channels), B operand = frames (cols = frames)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[kg][j]), __builtin_bit_cast(bf16x8_t, f.x[i]), acc[i][j], 0, 0, 0);
  };

  // ---- prologue: two windows in flight, weight fragments of step 0
  issue_A(0, 0);
  if (nchunks > 1) issue_A(1, 1);
#pragma unroll
  for (int kg = 0; kg < 4; ++kg) load_wf(kg, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int nsteps = nchunks * n_taps;
  int s = 0;
  for (int c = 0; c < nchunks; ++c) {
    const unsigned char *Ab = lds + (c % N_STAGES) * A_STAGE;
    for (int t = 0; t < n_taps; ++t, ++s) {
      int cn = c, tn = t + 1;
      if (tn == n_taps) { tn = 0; cn = c + 1; }
      const bool has_next = (ABL != 2) && (s + 1 < nsteps);
      const int d = p.taps[t];
      XFrags x0, x1;
      if (ABL != 2 || s == 0) load_x(Ab, d, 0, x0);
      if (ABL != 2) load_x(Ab, d, 1, x1);
      mma(x0, 0);
      if (has_next) load_wf(0, cn, tn);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL != 2) load_x(Ab, d, 2, x0);
      mma(ABL == 2 ? x0 : x1, 1);
      if (has_next) load_wf(1, cn, tn);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL != 2) load_x(Ab, d, 3, x1);
      mma(x0, 2);
      if (has_next) load_wf(2, cn, tn);
      __builtin_amdgcn_sched_barrier(0);
      mma(ABL == 2 ? x0 : x1, 3);
      if (has_next) load_wf(3, cn, tn);
      __builtin_amdgcn_sched_barrier(0);
    }
    // chunk boundary: refill the stage chunk c-1 used (free since the previous barrier) with chunk c+2,
    // then make sure chunk c+1's window has landed everywhere.  Younger than that window in this wave's
    // VMEM queue are only: the 8 fragment loads of the last step and the 4-5 pieces issued just now.
    const bool more = (ABL != 2) && (c + 2 < nchunks);
    if (more) issue_A(c + 2, (c + 2) % N_STAGES);
    if (c + 1 < nchunks) {
      if (more) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // every wave is done reading the ring: the epilogue reuses the LDS
  asm volatile("" ::: "memory");

  if constexpr (POOL) {
    // ---- epilogue with fused statistics pooling (pooling.py:58-67 folded into the producing layer):
    // the layer's output never reaches HBM.  Each wave owns 128 frames x 64 channels.  Per 32-frame
    // fragment it writes u = max(acc + b, lo) * scale (the output minus the BN shift: >= 0, well
    // conditioned for one-pass moments) as f32 to an LDS scratch [32 frames][64 ch], then lane = channel
    // sums the 32 rows.  The row -> segment lookup is wave-uniform, so a segment boundary is a scalar
    // branch: flush (sum u, sum u^2) of the finished segment to the partial buffer
    //   P[half-tile of 128 rows][segment slot][stat][channel]
    // and continue.  pool_finish_kernel adds the half-tiles of each segment in row order.
    constexpr int SPITCH = 68;                       // floats per scratch row: 64 + 4 keeps ds_write_b128 conflict-free
    float *scrf = reinterpret_cast<float *>(lds + wave * (32 * SPITCH * 4));
    const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
    const int rbase_w = m0 + wm * 128;
    const int half = rbase_w >> 7;
    int first_seg = -1;
#pragma unroll
    for (int k = 0; k < kHalo + 1; ++k)
      if (first_seg < 0 && rbase_w + k < p.rows) first_seg = p.row_seg[rbase_w + k];
    const int ch_l = n0 + wn * 64 + lane;
    const int rowseg_lo = p.row_seg[rbase_w + lane], rowseg_hi = p.row_seg[rbase_w + 64 + lane];
    float ps = 0.0f, pq = 0.0f;
    int cur_seg = -1;
    auto flush = [&]() {
      const int slot = cur_seg - first_seg;
      if (slot >= 0 && slot < p.pool_slots && ch_l < p.ld_partial) {
        float *dst = p.pool_partial + ((size_t)(half * p.pool_slots + slot) * 2) * p.ld_partial + ch_l;
        dst[0] = ps;
        dst[p.ld_partial] = pq;
      }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool valid = (p.row_valid[(rbase_w + i * 32) >> 5] >> lr) & 1u;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = n0 + wn * 64 + j * 32 + 8 * q + 4 * lh;
          const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
          const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
          float4 u;
          u.x = valid ? fmaxf(acc[i][j][q * 4 + 0] + b4.x, act_lo) * sc4.x : 0.0f;
          u.y = valid ? fmaxf(acc[i][j][q * 4 + 1] + b4.y, act_lo) * sc4.y : 0.0f;
          u.z = valid ? fmaxf(acc[i][j][q * 4 + 2] + b4.z, act_lo) * sc4.z : 0.0f;
          u.w = valid ? fmaxf(acc[i][j][q * 4 + 3] + b4.w, act_lo) * sc4.w : 0.0f;
          *reinterpret_cast<float4 *>(scrf + lr * SPITCH + j * 32 + 8 * q + 4 * lh) = u;
        }
      // column sums of this fragment; rows are consumed in order, segments are contiguous in rows.
      // rowseg_lo/hi hold row_seg of the wave's 128 rows (lane l: rows l and 64 + l), so the per-row
      // segment id is a v_readlane with a constant lane: no memory access in the loop.
      const int rs_vec = (i < 2) ? rowseg_lo : rowseg_hi;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int sg = __builtin_amdgcn_readlane(rs_vec, (i & 1) * 32 + r);          // wave-uniform
        if (sg >= 0 && sg != cur_seg) {
          if (cur_seg >= 0) flush();
          cur_seg = sg; ps = 0.0f; pq = 0.0f;
        }
        const float v = scrf[r * SPITCH + lane];
        ps += v;
        pq = fmaf(v, v, pq);
      }
    }
    if (cur_seg >= 0) flush();
    return;
  }
  // ---- epilogue --------------------------------------------------------------------------
  // acc[i][j][r]: frame = m0 + wm*128 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  unsigned char *scr = lds + wave * 16384;       // [128 frames][64 channels] bf16, 128-B rows, swizzled slots
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  uint32_t vmask = 0;                            // bit i: this lane's frame of m-fragment i is a real frame
#pragma unroll
  for (int i = 0; i < 4; ++i) vmask |= ((p.row_valid[(m0 + wm * 128 + i * 32) >> 5] >> lr) & 1u) << i;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = n0 + wn * 64 + j * 32 + 8 * q + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(p.bias + ch);
      const float4 sc4 = p.scale ? *reinterpret_cast<const float4 *>(p.scale + ch) : make_float4(1.f, 1.f, 1.f, 1.f);
      const float4 sh4 = p.shift ? *reinterpret_cast<const float4 *>(p.shift + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      const int slot = j * 4 + q;                // channel offset j*32 + 8*q + 4*lh -> 16-B slot, 8-B half lh
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (vmask >> i) & 1u;
        const int frow = i * 32 + lr;            // row inside the wave's scratch tile
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (GENERIC) {
            float z = acc[i][j][q * 4 + e] + b[e];
            z = p.affine_first ? apply_act(z * sc[e] + sh[e], p.act1) : apply_act(z, p.act1) * sc[e] + sh[e];
            z = apply_act(z, p.act2);
            y[e] = valid ? z : 0.0f;
          } else {
            y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], valid);
          }
        }
        uint2 pk;
        pk.x = pack_bf16x2(y[0], y[1]);
        pk.y = pack_bf16x2(y[2], y[3]);
        // odd rows keep their two 8-byte halves swapped so rows r, r+1 (same slot) hit different banks
        *reinterpret_cast<uint2 *>(scr + frow * ROWB + swz(frow, slot) * 16 + ((lh ^ (frow & 1)) * 8)) = pk;
      }
    }
  }
  // the scratch tile belongs to this wave only: LDS ops of one wave complete in order
  {
    unsigned char *yg = reinterpret_cast<unsigned char *>(p.y);
    const size_t y_pitch = (size_t)p.ldy * 2;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int piece = it * 64 + lane, frow = piece >> 3, slot = piece & 7;
      uint4 v = *reinterpret_cast<const uint4 *>(scr + frow * ROWB + swz(frow, slot) * 16);
      if (frow & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      const int ch = n0 + wn * 64 + slot * 8;
      const int row = m0 + wm * 128 + frow;
      if (ABL == 4) { asm volatile("" ::"v"(v.x), "v"(v.w)); continue; }
      if (ch < p.cout_store) *reinterpret_cast<uint4 *>(yg + (size_t)row * y_pitch + (size_t)ch * 2) = v;
    }
  }
}

}  // namespace

bool tdnn_big3_supported(const TdnnKernelParams &p, bool bf16, bool out_f32) {
  TdnnKernelParams q = p;
  q.pool_partial = nullptr;                       // the fused-pooling form has the same requirements otherwise
  return p.wfrag != nullptr && tdnn_big_supported(q, bf16, out_f32);
}

int launch_tdnn_big3_variant(const TdnnKernelParams &p, int variant, hipStream_t s) {
  ASV_REQUIRE(p.rows % 256 == 0, "tdnn(big3): rows %d not a multiple of 256", p.rows);
  ASV_REQUIRE(p.wfrag != nullptr, "tdnn(big3): fragment-packed weights missing");
  const bool two_per_cu = variant < 100;                 // variants >= 100: the 256x256 / one-workgroup-per-CU geometry
  if (!two_per_cu) variant -= 100;
  const int bm = two_per_cu ? 128 : 256;
  const int m_tiles = p.rows / bm, n_tiles = round_up(p.cout_store, BN) / BN;
  const dim3 grid(m_tiles * n_tiles), block(two_per_cu ? 256 : 512);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  if (p.pool_partial != nullptr) {
    ASV_REQUIRE(fast && p.row_seg != nullptr && p.pool_slots >= 1, "tdnn(big3): fused pooling needs the plain epilogue and a row map");
    if (two_per_cu) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 1>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, true, 2>), grid, block, 0, s, p, m_tiles, n_tiles);
    ASV_HIP_CHECK(hipGetLastError());
    return ASV_OK;
  }
  if (two_per_cu) {
    switch (variant) {
      case 2: hipLaunchKernelGGL((tdnn_gemm_big3_kernel<2, false, false, 1>), grid, block, 0, s, p, m_tiles, n_tiles); break;
      case 4: hipLaunchKernelGGL((tdnn_gemm_big3_kernel<4, false, false, 1>), grid, block, 0, s, p, m_tiles, n_tiles); break;
      default:
        if (fast) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, false, 1>), grid, block, 0, s, p, m_tiles, n_tiles);
        else hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, true, false, 1>), grid, block, 0, s, p, m_tiles, n_tiles);
    }
  } else {
    switch (variant) {
      case 2: hipLaunchKernelGGL((tdnn_gemm_big3_kernel<2, false, false, 2>), grid, block, 0, s, p, m_tiles, n_tiles); break;
      case 4: hipLaunchKernelGGL((tdnn_gemm_big3_kernel<4, false, false, 2>), grid, block, 0, s, p, m_tiles, n_tiles); break;
      default:
        if (fast) hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, false, false, 2>), grid, block, 0, s, p, m_tiles, n_tiles);
        else hipLaunchKernelGGL((tdnn_gemm_big3_kernel<0, true, false, 2>), grid, block, 0, s, p, m_tiles, n_tiles);
    }
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_tdnn_big3(const TdnnKernelParams &p, hipStream_t s) { return launch_tdnn_big3_variant(p, p.big_one_per_cu ? 100 : 0, s); }

}  // namespace asv
