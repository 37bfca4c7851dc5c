// The f32x precision mode (f32 rows, three 16-bit matrix instructions per product on [hi | lo] operand halves) on the persistent
// 8-phase structure of kernels_tdnn_p8.hip (round 5): the wide plain frame layers of the parity-grade mode - ECAPA's 1024 -> 1024
// and 3072 -> 1536 layers are 67 % of its f32x step, on tdnn_gemm_x3_kernel at 0.35 - 0.37 of the mode's 833 TFLOP/s (a chunk barrier
// and a workgroup-wide conversion pass per 48 matrix instructions).
//
// What carries over unchanged: 256 x 256 tiles, one workgroup of 8 waves per CU walking its tiles, both operands through LDS-DMA into
// two buffers of four 16 KiB half-tiles (128 rows x 128 bytes, slots XOR-swizzled on the source address and the read address), two
// phases per K-tile with counted waits, the wave rows staggered by one barrier, the next tile's first K-tile requested in front of
// the epilogue, parameters and validity words by LDS-DMA.  The RAW / WAR argument is that file's.
//
// What differs:
//   * a K-tile is 32 channels (x one tap): a row of the feature half-tiles is 32 f32 = 128 bytes, a row of the weight half-tiles is
//     one output channel's [hi 32 x 16-bit | lo 32 x 16-bit] = 128 bytes - weights pre-split on the host (runtime.hip
//     pack_tdnn_weight_x3p: w * 2^s = hi + lo, the layer's power of two as for tdnn_gemm_x3_kernel), in plain [cout][tap][chunk]
//     order;
//   * the feature fragments are split in registers by the wave that uses them (device_utils.h x3_split: v_cvt_pk_f16_f32,
//     v_fma_mix_f32, v_cvt_pk_f16_f32): 4 fragments = 8 ds_read_b128 + 64 VALU operations per phase, in the phase's load section -
//     under the partner wave row's 24 matrix instructions; the range watch of the half split moves to the accumulators (below);
//   * 3 matrix instructions per (frame fragment, channel fragment, k-group): w_hi x_hi, w_hi x_lo, w_lo x_hi, one accumulator;
//   * epilogue in f32: acc / 2^s + bias -> [ReLU] -> folded BN, four passes of 32 rows per wave through buffer 1, 16-byte stores.
// Same sums as tdnn_gemm_x3_kernel in another order (chunks of 32 channels x taps, three products per k-group): equal to f32
// rounding, not bit for bit; tests/test_gpu_kernels.py runs both against the f64 oracle.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int PX_HALF = 128 * 128;
constexpr int PX_OFF_B0 = 0, PX_OFF_A0 = PX_HALF, PX_OFF_B1 = 2 * PX_HALF, PX_OFF_A1 = 3 * PX_HALF;
constexpr int PX_BUF = 4 * PX_HALF;
constexpr int PX_PAR_SLOT = 4096;
constexpr int PX_LDS_BYTES = 2 * PX_BUF + 2 * PX_PAR_SLOT;
constexpr int PX_ROWB = 128;

typedef __attribute__((address_space(3))) unsigned char px_lds_byte;

__device__ __forceinline__ void px_glds(const void *sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory", "m0");
}

template <int XET, bool ONE_TAP>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_p8x_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[PX_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  const int total = m_tiles * n_tiles;
  const int grid = gridDim.x;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *wg = reinterpret_cast<const unsigned char *>(p.wx3p);
  const uint32_t x_pitch = (uint32_t)p.ldx * 4u;
  const int nchunks = p.cin_pad / 32;
  const int n_taps = ONE_TAP ? 1 : p.n_taps;
  const uint32_t w_pitch = (uint32_t)n_taps * (uint32_t)nchunks * 128u;      // bytes per output channel
  const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(px_lds_byte *)lds);
  const int nkt = nchunks * n_taps;
  const int last_row = p.rows - 1;
  const int g_row = lane >> 3, g_slot = lane & 7;

  struct TileAddr { int m0, n0; int a_row[2]; uint32_t a_slot[2], a_voff[2][2], b_off[2]; };
  auto tile_addr = [&](int it) {
    TileAddr t;
    const int tile = xcd_swizzle(it, total);
    t.m0 = (tile / n_tiles) * 256;
    t.n0 = (tile % n_tiles) * 256;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (wave * 2 + i) * 8 + g_row;
      const uint32_t slot16 = (uint32_t)(g_slot ^ ((r >> 1) & 7)) * 16u;
      t.a_row[i] = t.m0 + (r >> 6) * 128 + (r & 63);
      t.a_slot[i] = slot16;
      t.a_voff[0][i] = (uint32_t)t.a_row[i] * x_pitch + slot16;
      t.a_voff[1][i] = (uint32_t)min(t.a_row[i] + 64, last_row) * x_pitch + slot16;
      t.b_off[i] = (uint32_t)(t.n0 + (r >> 5) * 64 + (r & 31)) * w_pitch + slot16;
    }
    return t;
  };
  int v_taps = p.taps[0];
  if (!ONE_TAP) {
#pragma unroll
    for (int t = 1; t < ASV_MAX_TAPS; ++t) v_taps = (lane == t) ? p.taps[t] : v_taps;
  }
  // half-tile `which` (0 HB0, 1 HA0, 2 HB1, 3 HA1) of K-tile (chunk c of 32 channels, tap t) -> buffer b
  auto stage = [&](const TileAddr &T, int which, int c, int t, int b) {
    const uint32_t dst0 = lds_base + (uint32_t)b * PX_BUF + (uint32_t)which * PX_HALF + (uint32_t)wave * 2048u;
    if (which & 1) {
      const unsigned char *base = xg + (size_t)c * 128;
      if (ONE_TAP) {
#pragma unroll
        for (int i = 0; i < 2; ++i) px_glds(base, T.a_voff[which == 3][i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
      } else {
        const int d = __builtin_amdgcn_readlane(v_taps, t) + (which == 3 ? 64 : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = min(max(T.a_row[i] + d, 0), last_row);
          px_glds(base, (uint32_t)row * x_pitch + T.a_slot[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
        }
      }
    } else {
      const unsigned char *base = wg + ((size_t)t * nchunks + (size_t)c) * 128 + (which == 2 ? (size_t)32 * w_pitch : 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) px_glds(base, T.b_off[i], __builtin_amdgcn_readfirstlane(dst0 + i * 1024u));
    }
  };
  auto stage_tile_params = [&](const TileAddr &T, int slot) {
    const uint32_t dst = lds_base + 2u * PX_BUF + (uint32_t)slot * PX_PAR_SLOT + (uint32_t)wave * 1024u;
    if (wave < 3) {
      const float *src = (wave == 0) ? p.bias : (wave == 1 ? p.scale : p.shift);
      if (src != nullptr) {
        px_glds(src + T.n0, (uint32_t)lane * 16u, __builtin_amdgcn_readfirstlane(dst));
      } else {
        const float dflt = (wave == 1) ? 1.0f : 0.0f;
        *reinterpret_cast<float4 *>(lds + 2 * PX_BUF + slot * PX_PAR_SLOT + wave * 1024 + lane * 16) = make_float4(dflt, dflt, dflt, dflt);
      }
    } else if (wave == 3) {
      px_glds(p.row_valid + (T.m0 >> 5), lane < 2 ? (uint32_t)lane * 16u : 0u, __builtin_amdgcn_readfirstlane(dst));
    }
  };

  // fragment read addresses.  Feature rows: k-group kg (16 channels = 64 bytes) of this lane = 8 consecutive f32 = the 16-byte slots
  // 4 kg + 2 lh and + 1; weight rows: hi half of k-group kg = slot 2 kg + lh, lo half = slot 4 + 2 kg + lh
  const uint32_t sw = (uint32_t)((lr >> 1) & 7);
  const uint32_t a_base = (uint32_t)(wm * 64 + lr) * PX_ROWB, b_base = (uint32_t)(wn * 32 + lr) * PX_ROWB;
  uint32_t a_addr[2][2], b_addr[2][2];        // [kg][piece] / [kg][hi | lo]
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_addr[kg][h] = a_base + ((((uint32_t)(kg * 4 + lh * 2 + h)) ^ sw) * 16u);
      b_addr[kg][h] = b_base + ((((uint32_t)(h * 4 + kg * 2 + lh)) ^ sw) * 16u);
    }
  }
  auto barrier = [&]() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto kt_ct = [&](int kt, int &c, int &t) { c = kt / n_taps; t = kt - c * n_taps; };
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  const float unscale = p.w_unscale;
  float *yg = reinterpret_cast<float *>(p.y);
  // Range watch of the half split (device_utils.h): here on the ACCUMULATORS, once per tile, not on every split - a feature beyond the
  // IEEE-half range splits into hi = +-inf (lo = -+inf / NaN), and every product of that frame row is then +-inf or NaN (0 x inf), so
  // every accumulator of the row ends non-finite: 128 v_cmp_class per tile instead of 12 VALU operations per split x 8 splits per K-tile
  bool bad = false;

  int it = blockIdx.x;
  int slot = 0;
  TileAddr cur = tile_addr(it);
  stage_tile_params(cur, 0);
  stage(cur, 0, 0, 0, 0); stage(cur, 1, 0, 0, 0); stage(cur, 2, 0, 0, 0); stage(cur, 3, 0, 0, 0);
  if (nkt > 1) {
    int c1, t1;
    kt_ct(1, c1, t1);
    stage(cur, 0, c1, t1, 1); stage(cur, 1, c1, t1, 1);
  }
#pragma unroll 1
  while (true) {
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    barrier();
    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    uint4 wbh[2][2], wbl[2][2];               // [hb][kg]: hi / lo halves of the weight fragments of HB0 / HB1
    X3Frag xa[2][2];                          // [i2][kg]: the split frame fragments of the A half-tile in use
    if (wm == 1) __builtin_amdgcn_s_barrier();
    // one accumulator quadrant: (frame fragments i0, i0 + 1) x (channel fragment j): 2 k-groups x 3 products x 2 = 12 matrix instructions;
    // term-major inside a k-group, so an accumulator is touched every second instruction
    auto mma_q = [&](int j, int i0) {
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const uint4 a = (term == 2) ? wbl[j][kg] : wbh[j][kg];
            const uint4 b = (term == 1) ? xa[i2][kg].lo : xa[i2][kg].hi;
            acc[i0 + i2][j] = mfma16<XET>(a, b, acc[i0 + i2][j]);
          }
    };
    int c1 = 0, t1 = 0, c2 = 0, t2 = 0;
    auto adv = [&](int &c, int &t) { if (++t == n_taps) { t = 0; ++c; } };
    adv(c1, t1); adv(c2, t2); adv(c2, t2);
    auto ktile2 = [&](int kt, auto tail_c) {
      constexpr int TAIL = decltype(tail_c)::value;
      const int b = kt & 1;
      const unsigned char *L = lds + (uint32_t)b * PX_BUF;
      // ---- PA: HB0, HA0 (raw f32), HB1
      uint4 raw[2][2][2];                       // [i2][kg][piece]
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        wbh[0][kg] = *reinterpret_cast<const uint4 *>(L + PX_OFF_B0 + b_addr[kg][0]);
        wbl[0][kg] = *reinterpret_cast<const uint4 *>(L + PX_OFF_B0 + b_addr[kg][1]);
      }
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int h = 0; h < 2; ++h) raw[i2][kg][h] = *reinterpret_cast<const uint4 *>(L + PX_OFF_A0 + i2 * (32 * PX_ROWB) + a_addr[kg][h]);
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        wbh[1][kg] = *reinterpret_cast<const uint4 *>(L + PX_OFF_B1 + b_addr[kg][0]);
        wbl[1][kg] = *reinterpret_cast<const uint4 *>(L + PX_OFF_B1 + b_addr[kg][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (TAIL <= 1) { stage(cur, 2, c1, t1, b ^ 1); stage(cur, 3, c1, t1, b ^ 1); }
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = x3_split<XET, true>(raw[i2][kg][0], raw[i2][kg][1]);
      if (TAIL <= 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma_q(0, 0);
      mma_q(1, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
      // ---- PB: HA1
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
          for (int h = 0; h < 2; ++h) raw[i2][kg][h] = *reinterpret_cast<const uint4 *>(L + PX_OFF_A1 + i2 * (32 * PX_ROWB) + a_addr[kg][h]);
      __builtin_amdgcn_sched_barrier(0);
      if (TAIL == 0) { stage(cur, 0, c2, t2, b); stage(cur, 1, c2, t2, b); }
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) xa[i2][kg] = x3_split<XET, true>(raw[i2][kg][0], raw[i2][kg][1]);
      if (TAIL == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else if (TAIL == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      mma_q(1, 2);
      mma_q(0, 2);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      barrier();
      adv(c1, t1); adv(c2, t2);
    };
    using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
    int kt = 0;
    for (; kt + 2 < nkt; ++kt) ktile2(kt, J0{});
    if (kt + 1 < nkt) { ktile2(kt, J1{}); ++kt; }
    ktile2(kt, J2{});
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const bool has_next = it + grid < total;
    TileAddr nxt = cur;
    if (has_next) {
      nxt = tile_addr(it + grid);
      stage_tile_params(nxt, slot ^ 1);
      stage(nxt, 0, 0, 0, 0); stage(nxt, 1, 0, 0, 0); stage(nxt, 2, 0, 0, 0); stage(nxt, 3, 0, 0, 0);
    }
    // ---- epilogue: f32 rows, four passes of 32 rows per wave through buffer 1 ([32 rows][64 channels] f32 = 8 KiB per wave, 16-byte
    // slots XOR-swizzled by the row)
    {
      // (lane-derived values re-materialised per tile: left visible as loop invariants, hipcc keeps the store addresses derived from them
      // live across the K loop - and spills them)
      int lr_e = lr, lh_e = lh, lane_e = lane;
      asm volatile("" : "+v"(lr_e), "+v"(lh_e), "+v"(lane_e));
      const float *par = reinterpret_cast<const float *>(lds + 2 * PX_BUF + slot * PX_PAR_SLOT);
      const uint32_t *vw = reinterpret_cast<const uint32_t *>(lds + 2 * PX_BUF + slot * PX_PAR_SLOT + 3072);
      unsigned char *scr = lds + PX_BUF + wave * 8192;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (vw[wm * 4 + i] >> lr_e) & 1u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh_e;
            const float4 b4 = *reinterpret_cast<const float4 *>(par + chl);
            const float4 sc4 = *reinterpret_cast<const float4 *>(par + 256 + chl);
            const float4 sh4 = *reinterpret_cast<const float4 *>(par + 512 + chl);
            const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (XET == ET_F16) bad |= __builtin_amdgcn_classf(acc[i][j][q * 4 + e], 0x207);       // sNaN | qNaN | -inf | +inf
              const float z = fmaxf(fmaf(acc[i][j][q * 4 + e], unscale, b[e]), act_lo) * sc[e] + sh[e];      // tdnn_gemm_x3_kernel's plain epilogue
              y[e] = valid ? z : 0.0f;
            }
            const int sl = j * 8 + 2 * q + lh_e;               // 16-byte slot of channels j*32 + 8 q + 4 lh .. + 4 inside the 256-byte row
            *reinterpret_cast<float4 *>(scr + lr_e * 256 + ((sl ^ (lr_e & 15)) << 4)) = make_float4(y[0], y[1], y[2], y[3]);
          }
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
          const int frow = s8 * 4 + (lane_e >> 4), sl = lane_e & 15;
          const float4 v = *reinterpret_cast<const float4 *>(scr + frow * 256 + ((sl ^ (frow & 15)) << 4));
          const int ch = cur.n0 + wn * 64 + sl * 4;
          const int row = cur.m0 + wm * 128 + i * 32 + frow;
          if (ch < p.cout_store) *reinterpret_cast<float4 *>(yg + (size_t)row * p.ldy + ch) = v;
        }
      }
    }
    if (!has_next) break;
    barrier();
    if (nkt > 1) {
      int c1n, t1n;
      kt_ct(1, c1n, t1n);
      stage(nxt, 0, c1n, t1n, 1); stage(nxt, 1, c1n, t1n, 1);
    }
    cur = nxt;
    slot ^= 1;
    it += grid;
  }
  x3_publish_range(bad ? 0x8000u : 0u, p.status);
}

}  // namespace

// Layers the f32x 8-phase kernel takes: f32 rows, all three products, the plain epilogue, whole 32-channel chunks, split weights packed
bool tdnn_p8x_supported(const TdnnKernelParams &p) {
  const bool fits32 = (unsigned long long)p.rows * (unsigned long long)p.ldx * 4ull < (1ull << 32) &&
                      (unsigned long long)round_up(p.cout_store, 256) * (unsigned long long)p.n_taps * (unsigned long long)(p.cin_pad / 32) * 128ull < (1ull << 32);
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first;
  return p.wx3p != nullptr && fits32 && fast && p.x2 == nullptr && p.seg_bias == nullptr && p.seg_scale == nullptr && p.res == nullptr &&
         p.pool_partial == nullptr && p.rows % 256 == 0 && p.rows >= 256 && p.cin_pad % 32 == 0 && p.cin_pad >= 32 && p.cout_store % 4 == 0 &&
         p.cout_store >= 192 && p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS && p.row_valid != nullptr && (p.x3_terms & 7) == 7 && p.w_unscale > 0.0f &&
         (p.x3_et == ET_F16 || p.x3_et == ET_BF16) && p.ldx % 4 == 0 && p.ldy % 4 == 0;
}

int launch_tdnn_p8x(const TdnnKernelParams &p, hipStream_t s) {
  ASV_REQUIRE(tdnn_p8x_supported(p), "tdnn(p8x): layer shape not supported (rows %d cin %d cout %d taps %d)", p.rows, p.cin_pad, p.cout_store, p.n_taps);
  const int m_tiles = p.rows / 256, n_tiles = round_up(p.cout_store, 256) / 256;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  const dim3 grid(std::min(m_tiles * n_tiles, cus)), block(512);
  const bool one = p.n_taps == 1 && p.taps[0] == 0;
  if (p.x3_et == ET_F16) {
    if (one) hipLaunchKernelGGL((tdnn_gemm_p8x_kernel<ET_F16, true>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_p8x_kernel<ET_F16, false>), grid, block, 0, s, p, m_tiles, n_tiles);
  } else {
    if (one) hipLaunchKernelGGL((tdnn_gemm_p8x_kernel<ET_BF16, true>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_p8x_kernel<ET_BF16, false>), grid, block, 0, s, p, m_tiles, n_tiles);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
