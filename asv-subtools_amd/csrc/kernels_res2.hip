// ECAPA-TDNN Res2NetBlock (model/ecapa_tdnn_xvector.py:42-75 of the reference) as ONE kernel.
//
//   x_0 .. x_7 = the eight 128-channel groups of the block input;  y_0 = x_0;
//   y_i = BN(ReLU(TDNN_{[-d, 0, d]}(y_{i-1} + x_i)))   (i = 1: TDNN(x_1));   output = cat(y_0 .. y_7)
//
// Seven dependent 128 -> 128 convolutions: as seven launches (the r1 path) each one reads two and writes one
// 76800 x 128 bf16 slice through HBM / L2 for 98 kFLOP per row - 37 us per launch at 200 TFLOP/s, 21 % of an ECAPA
// step.  Here a workgroup owns 128 output rows and walks the seven branches with the running tensor in LDS:
//   * window = 128 rows + 32 recomputed rows on each side (branch i needs y_{i-1} at rows +-d: after seven branches
//     7 d <= 28 rows of the margin are stale, the 128 central rows are exact; +19 % MFMA work at d = 4);
//   * three LDS images of the window, [row][128 ch] bf16 with the 16-byte slots XOR-swizzled by (row & 15):
//       A  input of the running convolution (read with the three taps as shifted rows; 4 zero rows above / below),
//       B  its output y_i (written by the epilogue, streamed to HBM in full 256-byte rows afterwards),
//       X  the next group x_{i+1}, fetched by LDS-DMA while the convolution runs;
//     the epilogue adds X to y_i and writes A for the next branch (bf16(bf16(y) + x): the rounding of the per-layer path);
//   * 8 waves = 4 channel fragments x 2 row halves, each 96 rows x 32 channels (3 accumulators), K = 3 taps x 128 as
//     24 k-groups: weight fragments from L2 (fragment order of pack_tdnn_weight_frags), 3 ds_read_b128 per 3 MFMAs;
//   * two barriers per branch.  One workgroup (512 threads, 147 KiB LDS) per CU.
//   * ALL 24 weight fragments of a branch live in registers (96 VGPRs), fetched during the previous branch's epilogue and
//     issued BEFORE that epilogue's window DMA: the vector-memory counter retires in order, so a K loop that fetches its
//     fragments a few k-groups ahead (the first version) waits at its first s_waitcnt for the 48 KiB window of the next group
//     that was issued just before it - measured (ASV_AMD_RES2_DBG, r2f): 20 k cycles per K loop for 4.6 k of MFMA issue.
//     Now the K loop has no vector-memory operation at all and the window really lands behind it.  The fragment loads are
//     inline asm with counted s_waitcnt (hipcc's own waits would count only the loads it knows of and so, in hardware terms,
//     wait for the younger DMA pieces as well); for the same reason the per-channel constants of all branches are staged in
//     LDS once (a global load inside the loop is the youngest vector-memory operation: consuming it drains the DMA).
//
// Round 5: two window sizes (template parameter FR = row fragments of the window).  The tile count of a batch rarely fills whole rounds
// of the chip's 256 CUs (one workgroup per CU): ECAPA's 256 x 300-frame step is 77 824 rows = 608 tiles of 128 rows = 2.4 rounds, i.e.
// three.  FR = 7 (160 output rows + the same margins: 487 tiles = 1.9 rounds, i.e. two, and 71 % instead of 67 % of the window is
// output) needs 7 / 6 of the time per tile: 2 x 7 = 14 against 3 x 6 = 18.  The launcher picks the form with the smaller
// rounds x fragments product; every output row is computed from the same operands in the same order in both (the margins only decide
// which workgroup computes it), so the choice cannot change a value.  Seven fragments on 8 waves = 4 for the first row half, 3 for the
// second; waves w and w + 4 share a SIMD, so every SIMD runs 7.  To fit the larger window the images B and X share one buffer: the
// epilogue reads x_{i+1}[r][c] and writes y_i[r][c] to the same bytes, and a wave issues its pieces of the next window only behind its
// own reads of the rows it streams to HBM - the SAME rows (piece i of wave w = rows 32 i + 4 w .. + 3 = what wave w streams in
// iteration i - 1), so no other wave's read can meet the DMA.
#include <algorithm>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int RW = kRes2Width;            // 128 channels per branch
constexpr int RMARGIN = 32;               // recomputed rows per side (>= 7 branches x dilation 4)
constexpr int RPAD = 4;                   // zero rows around A (taps of the outermost window rows)
constexpr int RROWB = RW * 2;             // 256 B per row
template <int FR>
struct Res2Geom {
  static constexpr int WIN = FR * 32;                       // window rows: 192 / 224
  static constexpr int M = WIN - 2 * RMARGIN;               // output rows per workgroup: 128 / 160
  static constexpr int NA = (FR + 1) / 2;                   // row fragments of the first row half (the second has FR - NA)
  static constexpr int A_BYTES = (WIN + 2 * RPAD) * RROWB;  // 51200 / 59392
  static constexpr int B_OFF = A_BYTES;                     // B and X: one image
  static constexpr int PAR_OFF = B_OFF + WIN * RROWB;       // bias | scale | shift of all (<= 7) branches, f32 [branch][3][128]
  static constexpr int LDS = PAR_OFF + 7 * 3 * RW * 4;      // 111104 / 127488
  static_assert(LDS <= 163840, "160 KiB of LDS per CU");
  static_assert(WIN % 32 == 0 && (WIN / 4) % 8 == 0 && (M * 16) % 512 == 0, "whole fragments, whole DMA / streaming rounds of the 8 waves");
};
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
static_assert(7 * kHalo <= RMARGIN && kHalo <= RPAD, "margin must cover seven branches of the largest dilation");

typedef __attribute__((address_space(3))) unsigned char res2_lds_byte;

__device__ __forceinline__ void res2_glds16(const void *gsrc, uint32_t lds_dst) {
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

template <int ET, int FR>
__global__ __launch_bounds__(512, 2) void res2_chain_kernel(const Res2KernelParams p) {
  using G = Res2Geom<FR>;
  constexpr int RWIN = G::WIN, RM = G::M, NA = G::NA, B_OFF = G::B_OFF, X_OFF = G::B_OFF, PAR_OFF = G::PAR_OFF;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nf = wave & 3, rh = wave >> 2;               // channel fragment, row half
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * RM;
  const int rbase = rh * NA * 32;                         // first window row of this wave
  const bool full = (FR % 2 == 0) || rh == 0;             // this wave has NA fragments (otherwise NA - 1): wave-uniform
  const int d = p.dilation;
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(res2_lds_byte *)lds);
  const unsigned char *hg = reinterpret_cast<const unsigned char *>(p.x);
  unsigned char *og = reinterpret_cast<unsigned char *>(p.y);
  const size_t x_pitch = (size_t)p.ldx * 2, y_pitch = (size_t)p.ldy * 2;

  // window of group g -> LDS image at `off` whose first row has buffer index `row0` (the swizzle uses the buffer row)
  auto issue_window = [&](int g, uint32_t off, int row0) {
    // (the lane index re-materialised per call: visible as a kernel invariant, hipcc keeps the 6 / 7 source addresses of a lane - 64-bit
    //  pairs - live across the K loop, and the FR = 7 form then spills them: scratch reloads next to the DMA pieces, each with a wait of
    //  the compiler's that drains the pieces issued before it)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#pragma unroll
    for (int i = 0; i < RWIN / 4 / 8; ++i) {                 // 48 / 56 four-row pieces, 6 / 7 per wave
      const int piece = wave + i * 8;
      const int r = piece * 4 + (lane_e >> 4);                // window row
      const int grow = min(max(m0 - RMARGIN + r, 0), p.rows - 1);   // beyond the ends: the first / last row are zero gap rows
      const int slot = (lane_e & 15) ^ ((row0 + r) & 15);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + off + (uint32_t)(row0 * RROWB) + (uint32_t)piece * 1024u);
      res2_glds16(hg + (size_t)grow * x_pitch + (size_t)g * RROWB + (size_t)slot * 16, dst);
    }
  };

  int n_stamp = 0;
  auto stamp = [&]() {
    if (p.dbg != nullptr && lane == 0 && n_stamp < 14) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + n_stamp] = __builtin_amdgcn_s_memtime();
    ++n_stamp;
  };
  stamp();                                                    // 0
  if (p.dbg != nullptr && lane == 0) p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 14] = __builtin_amdgcn_s_memrealtime();
  // weight fragments of one branch: [tap][chunk][k-group] = 24 blocks of 1 KiB per 32-channel fragment
  u32x4_t wf[24];
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto load_weights = [&](int b) {
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.wfrag) + (size_t)(b - 1) * (4 * 3 * 2 * 4 * 1024) + (size_t)nf * (3 * 2 * 4 * 1024);
#pragma unroll
    for (int k = 0; k < 24; ++k)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(wf[k]) : "v"(lane16), "s"(wb + (size_t)k * 1024) : "memory");
  };
  // after a counted s_waitcnt: pins every later use of the fragments behind it (volatile asm statements keep their order)
  auto fragments_ready = [&]() {
#pragma unroll
    for (int k = 0; k < 24; ++k) asm volatile("" : "+v"(wf[k]));
  };
  load_weights(1);
  {
    float *par = reinterpret_cast<float *>(lds + PAR_OFF);
    for (int i = tid; i < p.branches * 3 * (RW / 4); i += 512) {
      const int br = i / (3 * (RW / 4)), which = (i / (RW / 4)) % 3, c4 = (i % (RW / 4)) * 4;
      const float *src = (which == 0 ? p.bias : (which == 1 ? p.scale : p.shift)) + br * RW + c4;
      *reinterpret_cast<float4 *>(par + (br * 3 + which) * RW + c4) = *reinterpret_cast<const float4 *>(src);
    }
  }
  issue_window(1, 0, RPAD);                                   // A = x_1
  if (p.branches > 1) issue_window(2, X_OFF, 0);              // X = x_2
  if (tid < 2 * RPAD * 16) {                                  // the zero rows of A
    const int r = tid >> 4, row = r < RPAD ? r : RWIN + r;
    *reinterpret_cast<uint4 *>(lds + row * RROWB + (tid & 15) * 16) = make_uint4(0, 0, 0, 0);
  }
  // y_0 = x_0: the pass-through group, 128 rows x 256 B
#pragma unroll
  for (int it = 0; it < RM * 16 / 512; ++it) {
    // (FR = 7: the last tile may overhang the matrix.  Its rows are clamped onto the last row - a zero gap row in x and in y, written
    //  with the zeros it holds - not masked: a masked-off wave would skip the store, and the counted waits below count stores)
    const int idx = it * 512 + tid, row = min(m0 + (idx >> 4), p.rows - 1), slot = idx & 15;
    *reinterpret_cast<uint4 *>(og + (size_t)row * y_pitch + slot * 16) = *reinterpret_cast<const uint4 *>(hg + (size_t)row * x_pitch + slot * 16);
  }
  // validity of this lane's three window rows (gap rows and rows outside the matrix produce zeros)
  bool valid[NA];
#pragma unroll
  for (int rf = 0; rf < NA; ++rf) {
    const int grow = m0 - RMARGIN + rbase + rf * 32 + lr;
    valid[rf] = grow >= 0 && grow < p.rows && ((p.row_valid[grow >> 5] >> (grow & 31)) & 1u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  stamp();                                                    // 1: first windows in
  bool dma_behind = false;                                    // a window DMA was issued after this branch's weight fetch
#pragma unroll 1
  for (int b = 1; b <= p.branches; ++b) {
    const float *bias = reinterpret_cast<const float *>(lds + PAR_OFF) + (b - 1) * 3 * RW + nf * 32 + 4 * lh, *scale = bias + RW, *shift = bias + 2 * RW;
    // accumulators start from the bias: acc[rf][4 q + e] = channel nf*32 + 8 q + 4 lh + e of row rh*96 + rf*32 + lr
    f32x16_t acc[NA];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4 *>(bias + 8 * q);
#pragma unroll
      for (int rf = 0; rf < NA; ++rf) { acc[rf][q * 4 + 0] = b4.x; acc[rf][q * 4 + 1] = b4.y; acc[rf][q * 4 + 2] = b4.z; acc[rf][q * 4 + 3] = b4.w; }
    }
    // the fragments are older than everything issued behind them (a branch > 1: this wave's 6 window pieces if a window was
    // fetched, then its 4 row stores): wait for exactly those to remain.  Branch 1: the start-up wait below covered them.
    if (b > 1) {
      if (dma_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RWIN / 4 / 8 + RM * 16 / 512) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RM * 16 / 512) : "memory");
    }
    fragments_ready();
    auto x_addr = [&](int t) -> uint32_t {                     // fragment 0 of tap t, slot lh (k-group bits enter by XOR)
      const int row = RPAD + rbase + lr + (t - 1) * d;
      return (uint32_t)(row * RROWB + ((lh ^ (row & 15)) << 4));
    };
    uint4 xc[NA], xn[NA];
    uint32_t xa = x_addr(0);
#pragma unroll
    // (a wave without its last fragment - FR = 7, second row half - still reads those rows: addresses inside the LDS block, values unused;
    //  only the matrix instruction is skipped: one scalar branch per k-group instead of two)
    for (int rf = 0; rf < NA; ++rf) xc[rf] = *reinterpret_cast<const uint4 *>(lds + xa + rf * 32 * RROWB);
    // K loop: k-group g = (tap t, 64-channel chunk c, 16-channel group kg), fully unrolled (24 x 3 MFMAs); rows one k-group ahead
#pragma unroll
    for (int tc = 0; tc < 6; ++tc) {
      const int t = tc >> 1, c = tc & 1;
      const uint32_t xa_next = x_addr(tc + 1 < 6 ? (tc + 1) >> 1 : t);
      const int c_next = (tc + 1) & 1;
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const bool wrap = kg == 3;
        const uint32_t a = (wrap ? xa_next : xa) ^ (uint32_t)((((wrap ? c_next : c) * 8 + ((kg + 1) & 3) * 2)) << 4);
        if (!(tc == 5 && kg == 3)) {
#pragma unroll
          for (int rf = 0; rf < NA; ++rf) xn[rf] = *reinterpret_cast<const uint4 *>(lds + a + rf * 32 * RROWB);
        }
#pragma unroll
        for (int rf = 0; rf < NA; ++rf)
          if (rf < NA - 1 || full) acc[rf] = mfma16<ET>(__builtin_bit_cast(uint4, wf[tc * 4 + kg]), xc[rf], acc[rf]);
#pragma unroll
        for (int rf = 0; rf < NA; ++rf) xc[rf] = xn[rf];
      }
      xa = xa_next;
    }
    // the fragment registers are free: fetch the next branch's (they fly during the epilogue; issued BEFORE the window DMA below)
    if (b < p.branches) load_weights(b + 1);
    if (b <= 3) stamp();                                        // 2, 6, 10: K loop done
    // this wave's pieces of X (issued a branch ago) have landed: everything older than the 24 fragment loads just issued
    if (b < p.branches) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // nobody reads A any more; X is complete
    asm volatile("" ::: "memory");
    if (b <= 3) stamp();                                        // 3, 7, 11: wait + barrier

    // epilogue: y = BN(ReLU(acc)) (zeros in gap rows) -> B; y + x_{b+1} -> A for the next branch
    const bool more = b < p.branches;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 sc4 = *reinterpret_cast<const float4 *>(scale + 8 * q), sh4 = *reinterpret_cast<const float4 *>(shift + 8 * q);
      const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      const int slot = nf * 4 + q;
#pragma unroll
      for (int rf = 0; rf < NA; ++rf) {
        if (!(rf < NA - 1 || full)) continue;
        const int rw = rbase + rf * 32 + lr;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = valid[rf] ? fmaf(max_lo(acc[rf][q * 4 + e], 0.0f), sc[e], sh[e]) : 0.0f;
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        const uint32_t boff = (uint32_t)(rw * RROWB + ((slot ^ (rw & 15)) << 4) + lh * 8);
        uint2 xv = make_uint2(0u, 0u);
        if (more) xv = *reinterpret_cast<const uint2 *>(lds + X_OFF + boff);       // x_{b+1} at these bytes ... (one image: read first)
        *reinterpret_cast<uint2 *>(lds + B_OFF + boff) = pk;                        // ... y_b from now on
        if (more) {
          uint2 sv;
          float p0, p1, p2, p3, x0, x1, x2, x3;
          unpack_h16x2<ET>(pk.x, p0, p1); unpack_h16x2<ET>(pk.y, p2, p3);
          unpack_h16x2<ET>(xv.x, x0, x1); unpack_h16x2<ET>(xv.y, x2, x3);
          sv.x = pack_h16x2<ET>(p0 + x0, p1 + x1);
          sv.y = pack_h16x2<ET>(p2 + x2, p3 + x3);
          const int ra = RPAD + rw;
          *reinterpret_cast<uint2 *>(lds + ra * RROWB + ((slot ^ (ra & 15)) << 4) + lh * 8) = sv;
        }
      }
    }
    if (b <= 3) stamp();                                        // 4, 8, 12: epilogue
    __builtin_amdgcn_s_barrier();                               // B and A complete; X consumed
    asm volatile("" ::: "memory");
    if (b <= 3) stamp();                                        // 5, 9, 13: barrier
    // y_b: the central rows of B -> HBM, 256-byte rows.  The rows wave w reads in iteration `it` (RMARGIN + 32 it + 4 w .. + 3) are the
    // rows its own window piece it + 1 overwrites (issue_window: piece = w + 8 i covers rows 32 i + 4 w .. + 3), the margin pieces cover
    // rows nobody streams: a wave's reads are retired (lgkmcnt) before its DMA is issued, and no other wave's DMA touches them
    // (named registers, not an array: hipcc's alloca promotion moved a 4-element array of these into LDS - 32 KiB of it)
    constexpr int NIT = RM * 16 / 512;                          // 4 / 5
    auto y_read = [&](int it) {
      const int idx = it * 512 + tid, rw = RMARGIN + (idx >> 4), slot = idx & 15;
      return *reinterpret_cast<const uint4 *>(lds + B_OFF + rw * RROWB + ((slot ^ (rw & 15)) << 4));
    };
    auto y_store = [&](int it, const uint4 v) {
      const int idx = it * 512 + tid, slot = idx & 15, row = min(m0 + (idx >> 4), p.rows - 1);       // (overhang: zeros onto the last gap row)
      *reinterpret_cast<uint4 *>(og + (size_t)row * y_pitch + (size_t)b * RROWB + slot * 16) = v;
    };
    const uint4 y0 = y_read(0), y1 = y_read(1), y2 = y_read(2), y3 = y_read(3), y4 = NIT > 4 ? y_read(4) : make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_behind = b + 2 <= p.branches;
    if (dma_behind) issue_window(b + 2, X_OFF, 0);              // lands during the next branch's K loop
    y_store(0, y0); y_store(1, y1); y_store(2, y2); y_store(3, y3);
    if (NIT > 4) y_store(4, y4);
  }
  if (p.dbg != nullptr && lane == 0) {
    p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 13] = __builtin_amdgcn_s_memtime();            // end (overwrites stamp 13)
    p.dbg[((size_t)blockIdx.x * 8 + wave) * 16 + 15] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace

// window fragments for this batch: the form with the smaller (rounds of the chip's CUs) x (fragments per tile); ASV_AMD_RES2_FR = 6 | 7
// overrides (A/B; read once per process unless ASV_AMD_LIVE_TUNE is set)
static int res2_window_frags(int rows) {
  static const int forced = getenv("ASV_AMD_RES2_FR") != nullptr ? atoi(getenv("ASV_AMD_RES2_FR")) : 0;
  const bool live = getenv("ASV_AMD_LIVE_TUNE") != nullptr;          // (one lookup per launch; the A/B tests set it after the first launch)
  const int f = live ? (getenv("ASV_AMD_RES2_FR") != nullptr ? atoi(getenv("ASV_AMD_RES2_FR")) : 0) : forced;
  if (f == 6 || f == 7) return f;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  const long long t6 = rows / Res2Geom<6>::M, t7 = (rows + Res2Geom<7>::M - 1) / Res2Geom<7>::M;
  const long long c6 = ((t6 + cus - 1) / cus) * 6, c7 = ((t7 + cus - 1) / cus) * 7;
  return c7 < c6 ? 7 : 6;
}

int launch_res2_chain(const Res2KernelParams &p, hipStream_t s) {
  ASV_REQUIRE(p.rows % 128 == 0 && p.rows >= 128, "res2: rows %d not a multiple of 128", p.rows);
  ASV_REQUIRE(p.branches >= 1 && p.branches <= 7 && p.dilation >= 1 && p.dilation <= kHalo, "res2: %d branches, dilation %d", p.branches, p.dilation);
  ASV_REQUIRE(p.x && p.y && p.wfrag && p.bias && p.scale && p.shift && p.row_valid, "res2: null argument");
  if (res2_window_frags(p.rows) == 7) {
    const dim3 grid((p.rows + Res2Geom<7>::M - 1) / Res2Geom<7>::M);          // the last tile may overhang the matrix: its loads clamp, its stores are guarded
    if (p.et == ET_F16) hipLaunchKernelGGL((res2_chain_kernel<ET_F16, 7>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((res2_chain_kernel<ET_BF16, 7>), grid, dim3(512), 0, s, p);
  } else {
    const dim3 grid(p.rows / Res2Geom<6>::M);
    if (p.et == ET_F16) hipLaunchKernelGGL((res2_chain_kernel<ET_F16, 6>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((res2_chain_kernel<ET_BF16, 6>), grid, dim3(512), 0, s, p);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
