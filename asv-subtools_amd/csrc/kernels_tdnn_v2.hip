// bf16 TDNN / 1x1-conv implicit GEMM, large-tile variant for the frame layers that carry the
// FLOPs (the x-vector tdnn1-5, ECAPA's 1024x1024 / 3072x1536 convolutions).
//
//   workgroup tile 256 frames x 256 out-channels, 8 waves (2 along frames x 4 along channels),
//   each wave 128 x 64 = 4 x 2 MFMA 32x32x16 tiles (128 f32 accumulators per lane)
//
//   * operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip): one wave
//     instruction moves 8 rows x 128 B.  The LDS image is lane-linear, so the bank-conflict
//     swizzle (16-byte slot s of row r lives at s ^ ((r>>1)&7)) is applied to the per-lane
//     SOURCE address and again on the ds_read side (cdna_hip_programming.md rule 21).
//     Out-of-range rows / channel tails are redirected to a 16-byte zero page.
//   * two LDS stages each for the feature window (264 rows: 256 + 2*HALO, shared by all taps
//     of a 64-channel chunk) and for the weight tile; the loads of step s+1 (and of the next
//     chunk's window) are issued before the MFMAs of step s; one barrier per step.
//   * the weight fragment is the MFMA "A" operand and the frame fragment the "B" operand, so a
//     lane ends up with one FRAME and 4 consecutive CHANNELS per register quad: the epilogue
//     (bias, ReLU, folded BN, gap-row zeroing) packs 4 bf16 = 8 bytes per quad, stages the wave's
//     128x64 tile in LDS and stores it to HBM as full 128-byte row segments (16 B per lane).
#include "device_utils.h"

namespace asv {
namespace {

constexpr int BM = 256, BN = 256;
constexpr int WIN = BM + 2 * kHalo;          // 264
constexpr int ROWB = 128;
constexpr int A_STAGE = WIN * ROWB;          // 33792
constexpr int B_STAGE = BN * ROWB;           // 32768
constexpr int LDS_BYTES = 2 * A_STAGE + 2 * B_STAGE;   // 133120: one workgroup per CU
constexpr int A_GROUPS = WIN / 8;            // 33 eight-row groups
constexpr int B_GROUPS = BN / 8;             // 32
constexpr int BK = 64;                       // bf16 elements per chunk
static_assert(BN == kBigTileN, "weight padding must match the N tile");
static_assert(WIN % 8 == 0, "window must be a whole number of 8-row load groups");

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// One LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to the 1 KiB of LDS
// starting at the wave-uniform byte address `lds_dst`.  Issued from inline asm on purpose: hipcc
// cannot prove that the ds_reads of the current stage do not alias the stage being filled and would
// put `s_waitcnt vmcnt(0)` in front of them, serialising load and compute (cdna_hip_programming.md
// 5.7).  The data is ordered by our own `s_waitcnt vmcnt(0)` + barrier at the end of each step.
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
__device__ __forceinline__ void wait_all_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// ABL != 0 builds ablation variants for tools/gemm_ablate (cdna_hip_programming.md 5.4 rule 17):
//   1 no LDS-DMA after the prologue, 2 additionally no ds_reads in the loop (pure MFMA + barrier),
//   3 loads + ds_reads but no MFMA, 4 full main loop without the epilogue stores,
//   5 like 3 but every workgroup loads the SAME window / weight tile (cache-hot operands),
//   7 LDS-DMA + barrier only (no ds_read, no MFMA), 8 ds_read + barrier only
template <int ABL, bool GENERIC>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_big_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * BM;
  const int n0 = (tile % n_tiles) * BN;
  const int m0l = (ABL == 5) ? 0 : m0, n0l = (ABL == 5) ? 0 : n0;     // where the operand loads point

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *wg = reinterpret_cast<const unsigned char *>(p.w);
  const unsigned char *zero = reinterpret_cast<const unsigned char *>(p.zero16);
  const size_t x_pitch = (size_t)p.ldx * 2;
  const size_t w_tap_pitch = (size_t)p.cin_pad * 2;
  const size_t w_row_pitch = w_tap_pitch * p.n_taps;

  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte *)lds);

  // this lane's position inside an 8-row load group
  const int g_row = lane >> 3, g_slot = lane & 7;

  // One weight tile (tap t, chunk c) = 32 eight-row groups -> 4 DMA pieces per wave; one feature
  // window (chunk c) = 33 groups -> 4 pieces per wave + a 5th for wave 0.  Pieces are issued one
  // at a time, interleaved with the MFMA groups of the running step (each LDS-DMA costs the
  // issuing wave ~60-100 cycles of issue time; spread out, the partner wave's MFMAs cover it).
  auto issue_B_piece = [&](int c, int t, int st, int i) {
    const int grp = wave * (B_GROUPS / 8) + i;
    const int n = grp * 8 + g_row;
    const int ch = c * BK + swz(n, g_slot) * 8;
    const unsigned char *src = (ch < p.cin_pad) ? wg + (size_t)(n0l + n) * w_row_pitch + (size_t)t * w_tap_pitch + (size_t)ch * 2 : zero;
    glds16(src, __builtin_amdgcn_readfirstlane(lds_base + 2 * A_STAGE + st * B_STAGE + grp * 1024));
  };
  auto issue_A_piece = [&](int c, int st, int i) {
    const int grp = wave + i * 8;              // 33 groups over 8 waves: wave 0 takes the 33rd
    if (grp < A_GROUPS) {
      const int w = grp * 8 + g_row;
      const int row = m0l - kHalo + w;
      const int ch = c * BK + swz(w, g_slot) * 8;
      const bool ok = row >= 0 && row < p.rows && ch < p.cin_pad;
      const unsigned char *src = ok ? xg + (size_t)row * x_pitch + (size_t)ch * 2 : zero;
      glds16(src, __builtin_amdgcn_readfirstlane(lds_base + st * A_STAGE + grp * 1024));
    }
  };
  auto issue_B = [&](int c, int t, int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_B_piece(c, t, st, i);
  };
  auto issue_A = [&](int c, int st) {
#pragma unroll
    for (int i = 0; i < 5; ++i) issue_A_piece(c, st, i);
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragments of one k-group: 4 frame fragments + 2 weight fragments (16 bytes per lane each)
  struct Frags { uint4 xf[4], wf[2]; };
  auto load_frags = [&](const unsigned char *Ab, const unsigned char *Bb, int d, int kg, Frags &f) {
    const int slot = kg * 2 + lh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int w = wm * 128 + i * 32 + lr + kHalo + d;
      f.xf[i] = *reinterpret_cast<const uint4 *>(Ab + w * ROWB + swz(w, slot) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = wn * 64 + j * 32 + lr;
      f.wf[j] = *reinterpret_cast<const uint4 *>(Bb + n * ROWB + swz(n, slot) * 16);
    }
  };
  auto mma_frags = [&](const Frags &f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        // A operand = weights (rows = channels), B operand = frames (cols = frames)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.wf[j]), __builtin_bit_cast(bf16x8_t, f.xf[i]), acc[i][j], 0, 0, 0);
  };
  // one K step = up to 4 k-groups, software pipelined: fragments of group g+1 are read from LDS
  // while the 8 MFMAs of group g run; sched_barrier keeps the compiler from hoisting every
  // ds_read of the step to the top (which spills the accumulators)
  Frags fa;
  if (ABL == 2) load_frags(lds, lds + 2 * A_STAGE, 0, 0, fa);
  // (cn, tn, bst): next step's weight tile and its stage; (ca, ast): next chunk's window and its stage
  auto compute_step = [&](const unsigned char *Ab, const unsigned char *Bb, int d, bool do_B, int cn, int tn, int bst, bool do_A, int ca, int ast) {
    if (ABL == 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { mma_frags(fa); __builtin_amdgcn_sched_barrier(0); }
      return;
    }
    if (ABL == 7) {
      if (do_B) issue_B(cn, tn, bst);
      if (do_A) issue_A(ca, ast);
      return;
    }
    if (ABL == 3 || ABL == 5 || ABL == 8) {
      if (do_B && ABL != 8) issue_B(cn, tn, bst);
      if (do_A && ABL != 8) issue_A(ca, ast);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Frags f;
        load_frags(Ab, Bb, d, g, f);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(f.xf[i].x), "v"(f.xf[i].w));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(f.wf[j].x), "v"(f.wf[j].w));
      }
      return;
    }
    Frags f0, f1;
    load_frags(Ab, Bb, d, 0, f0);
    if (do_B) issue_B_piece(cn, tn, bst, 0);
    if (do_A) { issue_A_piece(ca, ast, 0); issue_A_piece(ca, ast, 4); }
    load_frags(Ab, Bb, d, 1, f1);
    __builtin_amdgcn_s_setprio(1);
    mma_frags(f0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (do_B) issue_B_piece(cn, tn, bst, 1);
    if (do_A) issue_A_piece(ca, ast, 1);
    load_frags(Ab, Bb, d, 2, f0);
    __builtin_amdgcn_s_setprio(1);
    mma_frags(f1);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (do_B) issue_B_piece(cn, tn, bst, 2);
    if (do_A) issue_A_piece(ca, ast, 2);
    load_frags(Ab, Bb, d, 3, f1);
    __builtin_amdgcn_s_setprio(1);
    mma_frags(f0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (do_B) issue_B_piece(cn, tn, bst, 3);
    if (do_A) issue_A_piece(ca, ast, 3);
    __builtin_amdgcn_s_setprio(1);
    mma_frags(f1);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nchunks = (p.cin_pad + BK - 1) / BK;
  const int nsteps = nchunks * p.n_taps;

  issue_A(0, 0);
  issue_B(0, 0, 0);
  wait_all_vmem();
  __syncthreads();                       // step 0 staged

  int c = 0, t = 0;
  for (int s = 0; s < nsteps; ++s) {
    int cn = c, tn = t + 1;
    if (tn == p.n_taps) { tn = 0; cn = c + 1; }
    const bool do_B = (ABL != 1 && ABL != 2) && (s + 1 < nsteps);
    const bool do_A = (ABL != 1 && ABL != 2) && (t == 0) && (c + 1 < nchunks);      // whole chunk of slack for the window
    {
      const unsigned char *Ab = lds + (c & 1) * A_STAGE;
      const unsigned char *Bb = lds + 2 * A_STAGE + (s & 1) * B_STAGE;
      const int d = p.taps[t];
      compute_step(Ab, Bb, d, do_B, cn, tn, (s + 1) & 1, do_A, c + 1, (c + 1) & 1);      // a channel tail (cin_pad % 64) was staged as zeros
    }
    wait_all_vmem();                     // this wave's LDS-DMA of the next step has landed ...
    __syncthreads();                     // ... everyone's has; stages of step s are free
    c = cn; t = tn;
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

bool tdnn_big_supported(const TdnnKernelParams &p, bool bf16, bool out_f32) {
  return bf16 && !out_f32 && p.x2 == nullptr && p.seg_bias == nullptr && p.seg_scale == nullptr && p.res == nullptr && p.pool_partial == nullptr &&
         p.zero16 != nullptr && p.rows % BM == 0 && p.cout_store % 8 == 0 && p.cout_store >= 192 && p.cin_pad >= 64;
}

int launch_tdnn_big(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows % BM == 0, "tdnn(big): rows %d not a multiple of %d", p.rows, BM);
  for (int t = 0; t < p.n_taps; ++t)
    ASV_REQUIRE(p.taps[t] >= -kHalo && p.taps[t] <= kHalo, "tdnn(big): tap offset %d exceeds the %d-frame halo", p.taps[t], kHalo);
  const int m_tiles = p.rows / BM;
  const int n_tiles = round_up(p.cout_store, BN) / BN;
  const dim3 grid(m_tiles * n_tiles), block(512);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  if (fast) hipLaunchKernelGGL((tdnn_gemm_big_kernel<0, false>), grid, block, 0, s, p, m_tiles, n_tiles);
  else hipLaunchKernelGGL((tdnn_gemm_big_kernel<0, true>), grid, block, 0, s, p, m_tiles, n_tiles);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

int launch_tdnn_big_variant(const TdnnKernelParams &p, int variant, hipStream_t s) {
  const int m_tiles = p.rows / BM, n_tiles = round_up(p.cout_store, BN) / BN;
  const dim3 grid(m_tiles * n_tiles), block(512);
  switch (variant) {
    case 1: hipLaunchKernelGGL((tdnn_gemm_big_kernel<1, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 2: hipLaunchKernelGGL((tdnn_gemm_big_kernel<2, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 3: hipLaunchKernelGGL((tdnn_gemm_big_kernel<3, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 4: hipLaunchKernelGGL((tdnn_gemm_big_kernel<4, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 5: hipLaunchKernelGGL((tdnn_gemm_big_kernel<5, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 7: hipLaunchKernelGGL((tdnn_gemm_big_kernel<7, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 8: hipLaunchKernelGGL((tdnn_gemm_big_kernel<8, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    case 6: hipLaunchKernelGGL((tdnn_gemm_big_kernel<0, true>), grid, block, 0, s, p, m_tiles, n_tiles); break;
    default: hipLaunchKernelGGL((tdnn_gemm_big_kernel<0, false>), grid, block, 0, s, p, m_tiles, n_tiles); break;
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
