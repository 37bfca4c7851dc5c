// TDNN / 1x1-conv layer as an implicit GEMM on the MI355X matrix cores.
//
//   Y[row, co] = epilogue( sum_{tap} sum_{ci} X[row + tap, ci] * W[co, tap, ci] )
//
// replaces TdnnAffine.forward -> F.conv1d plus the ReLU / eval-BatchNorm passes that follow
// it (reference libs/nnet/components.py:107-149, 410-431).  Only the taps listed in the
// layer's `context` are multiplied (the reference multiplies the dense 5/7/9-tap kernel by a
// 0/1 mask every call, components.py:133-147).
//
// Tiling (v1):  128 frames x 128 out-channels per workgroup, 4 waves (2x2), each wave a
// 64x64 block = 2x2 MFMA 32x32 tiles, f32 accumulators in registers.
//   * K is walked as (channel chunk of 128 bytes) x (tap).  The A operand for ALL taps of a
//     chunk comes from ONE LDS-staged feature window of 128+2*HALO frames: tap d just reads
//     the window shifted by d rows, so activations are fetched from L2/HBM once per chunk
//     instead of once per tap.  Segment edges need no masks: the row layout keeps HALO zero
//     rows between utterances (asv_internal.h).
//   * LDS rows are 128 B (8 x 16-B slots); slot s of row r is stored at s ^ ((r>>1)&7), which
//     makes every ds_read_b128 of a fragment (32 consecutive rows, any window shift) and
//     every ds_write_b128 of the staging pass bank-conflict free (MI355X_MICROARCH.md LDS).
//   * bf16: v_mfma_f32_32x32x16_bf16, 8 k-values per lane per instruction.
//     f32 : v_mfma_f32_32x32x2_f32 x4 per 16-byte fragment; exact f32 fma chain (parity mode).
//     Both feed A and B with the same lane->k mapping, so any hardware k ordering inside the
//     instruction cancels out.
//   * Global->register prefetch of the next step is issued before the MFMAs of the current
//     step and written to the other LDS stage after them (one barrier per step).
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;                     // bytes per LDS row = one K chunk
constexpr int B_STAGE = BN * ROWB;            // 16384
constexpr int kWideHalo = 84;                 // window halo of the 2-D (grid) instantiations: |tap| <= pitch + 1 <= 84
constexpr int kMidHalo = 24;                  // grids of pitch <= 23 (ResNet stages 3 and 4 at 80-dim input: pitch 21 / 11): 176-row window, 77 KB, two workgroups per CU
// window geometry for a given halo: HALO = 4 -> 136 rows, 67.6 KB of LDS, 2 workgroups / CU;
// HALO = 84 (3x3 convolutions over row-flattened (time, frequency) grids) -> 296 rows, 108.5 KB
template <int HALO> struct Geom {
  static constexpr int WIN = BM + 2 * HALO;
  static constexpr int A_STAGE = WIN * ROWB;
  static constexpr int LDS_BYTES = 2 * A_STAGE + 2 * B_STAGE;
  static constexpr int A_PIECES = WIN * 8;
  static constexpr int NA = (A_PIECES + 255) / 256;      // 16-byte window pieces per thread
};
static_assert(kRowTile % BM == 0, "row padding must be a multiple of the M tile");

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

template <int ET, bool OUT16, bool GENERIC, int HALO>
__global__ __launch_bounds__(256, HALO <= 24 ? 2 : 1) void tdnn_gemm_kernel(const TdnnKernelParams p, int m_tiles, int n_tiles) {
  using G = Geom<HALO>;
  constexpr int A_STAGE = G::A_STAGE, A_PIECES = G::A_PIECES, NA = G::NA;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS_BYTES];
  constexpr bool H16 = ET != ET_F32;        // 16-bit elements (bf16 / f16) on the 32x32x16 matrix instruction; f32 on the exact 32x32x2 one
  constexpr int OUT_ET = OUT16 ? ET : ET_F32;
  constexpr int ES = H16 ? 2 : 4;           // element bytes
  constexpr int BK = ROWB / ES;             // elements per chunk: 64 | 32
  constexpr int E16 = 16 / ES;              // elements per 16-byte piece: 8 | 4
  constexpr int KGE = 2 * E16;              // elements per k-group (both lane halves): 16 | 8

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * BM;
  const int n0 = (tile % n_tiles) * BN;

  const unsigned char *xg = reinterpret_cast<const unsigned char *>(p.x);
  const unsigned char *x2g = reinterpret_cast<const unsigned char *>(p.x2);
  const unsigned char *wg = reinterpret_cast<const unsigned char *>(p.w);
  const size_t x_pitch = (size_t)p.ldx * ES, x2_pitch = (size_t)p.ldx2 * ES;
  const size_t w_tap_pitch = (size_t)p.cin_pad * ES;
  const size_t w_row_pitch = w_tap_pitch * p.n_taps;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  uint4 regA[NA], regB[4];

  auto gload_B = [&](int c, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid, n = q >> 3, slot = q & 7;
      const int ch = c * BK + slot * E16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ch < p.cin_pad)
        v = *reinterpret_cast<const uint4 *>(wg + (size_t)(n0 + n) * w_row_pitch + (size_t)t * w_tap_pitch + (size_t)ch * ES);
      regB[i] = v;
    }
  };
  auto gload_A = [&](int c) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = i * 256 + tid, w = q >> 3, slot = q & 7;
      const int row = m0 - HALO + w, ch = c * BK + slot * E16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < A_PIECES && row >= 0 && row < p.rows && ch < p.cin_pad) {
        v = *reinterpret_cast<const uint4 *>(xg + (size_t)row * x_pitch + (size_t)ch * ES);
        if (x2g != nullptr) {
          const uint4 v2 = *reinterpret_cast<const uint4 *>(x2g + (size_t)row * x2_pitch + (size_t)ch * ES);
          if constexpr (H16) v = add_h16x8<ET>(v, v2);
          else v = add_f32x4(v, v2);
        }
      }
      regA[i] = v;
    }
  };
  auto sstore_B = [&](int stage) {
    unsigned char *Bb = lds + 2 * A_STAGE + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * 256 + tid, n = q >> 3, slot = q & 7;
      *reinterpret_cast<uint4 *>(Bb + n * ROWB + swz(n, slot) * 16) = regB[i];
    }
  };
  auto sstore_A = [&](int stage) {
    unsigned char *Ab = lds + stage * A_STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = i * 256 + tid, w = q >> 3, slot = q & 7;
      if (q < A_PIECES) *reinterpret_cast<uint4 *>(Ab + w * ROWB + swz(w, slot) * 16) = regA[i];
    }
  };
  auto mma_group = [&](const unsigned char *Ab, const unsigned char *Bb, int d, int kg) {
    const int slot = kg * 2 + lh;
    uint4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int w = wm * 64 + i * 32 + lr + HALO + d;
      a[i] = *reinterpret_cast<const uint4 *>(Ab + w * ROWB + swz(w, slot) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = wn * 64 + j * 32 + lr;
      b[j] = *reinterpret_cast<const uint4 *>(Bb + n * ROWB + swz(n, slot) * 16);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr (H16) {
          acc[i][j] = mfma16<ET>(a[i], b[j], acc[i][j]);
        } else {
          const f32x4_t af = __builtin_bit_cast(f32x4_t, a[i]), bf = __builtin_bit_cast(f32x4_t, b[j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], acc[i][j], 0, 0, 0);
        }
      }
  };

  const int nchunks = (p.cin_pad + BK - 1) / BK;
  // split-K: this workgroup owns chunks [c_begin, c_end)
  int c_begin = 0, c_end = nchunks;
  if (p.ksplit > 1) {
    c_begin = (int)(((long long)nchunks * blockIdx.y) / p.ksplit);
    c_end = (int)(((long long)nchunks * (blockIdx.y + 1)) / p.ksplit);
  }
  const int nsteps = (c_end - c_begin) * p.n_taps;

  // prologue: stage step 0
  if (nsteps > 0) {
    gload_A(c_begin);
    gload_B(c_begin, 0);
    sstore_A(c_begin & 1);
    sstore_B(0);
  }
  __syncthreads();

  int c = c_begin, t = 0;
  for (int s = 0; s < nsteps; ++s) {
    int cn = c, tn = t + 1;
    if (tn == p.n_taps) { tn = 0; cn = c + 1; }
    const bool has_next = (s + 1 < nsteps);
    const bool next_chunk = has_next && (tn == 0);
    if (has_next) {
      gload_B(cn, tn);
      if (next_chunk) gload_A(cn);
    }
    {
      const unsigned char *Ab = lds + (c & 1) * A_STAGE;
      const unsigned char *Bb = lds + 2 * A_STAGE + (s & 1) * B_STAGE;
      const int d = p.taps[t];
      const int rem = p.cin_pad - c * BK;
      const int ng = (rem >= BK) ? (BK / KGE) : (rem / KGE);
      if (ng == BK / KGE) {
#pragma unroll
        for (int kg = 0; kg < BK / KGE; ++kg) mma_group(Ab, Bb, d, kg);
      } else {
        for (int kg = 0; kg < ng; ++kg) mma_group(Ab, Bb, d, kg);
      }
    }
    if (has_next) {
      sstore_B((s + 1) & 1);
      if (next_chunk) sstore_A(cn & 1);
    }
    __syncthreads();
    c = cn; t = tn;
  }

  // epilogue.  C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  if (p.ksplit > 1) {
    float *part = p.partial + (size_t)blockIdx.y * p.rows * p.ld_partial;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = n0 + wn * 64 + j * 32 + lr;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (ch < p.ld_partial) part[(size_t)row * p.ld_partial + ch] = acc[i][j][r];
        }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = n0 + wn * 64 + j * 32 + lr;
    const float bias = p.bias[ch];
    const float scale = p.scale ? p.scale[ch] : 1.0f;
    const float shift = p.shift ? p.shift[ch] : 0.0f;
    const bool ch_ok = ch < p.cout_store;
    const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rbase = m0 + wm * 64 + i * 32;
      const uint32_t vbits = p.row_valid[rbase >> 5];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rf = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int row = rbase + rf;
        const bool valid = (vbits >> rf) & 1u;
        float yv;
        if constexpr (GENERIC) yv = tdnn_epilogue<ET>(p, acc[i][j][r], row, ch, bias, scale, shift, valid);
        else yv = tdnn_epilogue_fast(acc[i][j][r], bias, act_lo, scale, shift, valid);
        if (ch_ok) store_elem<OUT_ET>(p.y, (size_t)row * p.ldy + ch, yv);
      }
    }
  }
}

// Plain-VALU self-check kernel: one thread per output element, natural k order, same packed
// weights and the same epilogue.  Used by ASV_FLAG_REF_KERNELS and the kernel parity tests.
template <int ET, bool OUT16>
__global__ __launch_bounds__(256) void tdnn_ref_kernel(const TdnnKernelParams p) {
  constexpr int OUT_ET = OUT16 ? ET : ET_F32;
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
  const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= p.rows || ch >= p.cout_store) return;
  float acc = 0.0f;
  for (int t = 0; t < p.n_taps; ++t) {
    const int rr = row + p.taps[t];
    if (rr < 0 || rr >= p.rows) continue;
    const size_t wbase = ((size_t)ch * p.n_taps + t) * p.cin_pad;
    for (int ci = 0; ci < p.cin_pad; ++ci) {
      float xv = load_elem<ET>(p.x, (size_t)rr * p.ldx + ci);
      if (p.x2 != nullptr) {
        xv += load_elem<ET>(p.x2, (size_t)rr * p.ldx2 + ci);
        if constexpr (ET != ET_F32) xv = h16_bits_to_f32<ET>(f32_to_h16_bits<ET>(xv));
      }
      acc = fmaf(xv, load_elem<ET>(p.w, wbase + ci), acc);
    }
  }
  const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
  const float scale = p.scale ? p.scale[ch] : 1.0f, shift = p.shift ? p.shift[ch] : 0.0f;
  const float yv = tdnn_epilogue<ET>(p, acc, row, ch, p.bias[ch], scale, shift, valid);
  store_elem<OUT_ET>(p.y, (size_t)row * p.ldy + ch, yv);
}

// second half of a split-K layer: sum the slices in order, then the usual epilogue
template <int ET, bool OUT16>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const TdnnKernelParams p) {
  constexpr int OUT_ET = OUT16 ? ET : ET_F32;
  const int ch = blockIdx.x * 64 + (threadIdx.x & 63);
  const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (row >= p.rows || ch >= p.cout_store) return;
  float acc = 0.0f;
  for (int z = 0; z < p.ksplit; ++z) acc += p.partial[((size_t)z * p.rows + row) * p.ld_partial + ch];
  const bool valid = (p.row_valid[row >> 5] >> (row & 31)) & 1u;
  const float scale = p.scale ? p.scale[ch] : 1.0f, shift = p.shift ? p.shift[ch] : 0.0f;
  store_elem<OUT_ET>(p.y, (size_t)row * p.ldy + ch, tdnn_epilogue<ET>(p, acc, row, ch, p.bias[ch], scale, shift, valid));
}

}  // namespace

// et: ET_F32 / ET_BF16 / ET_F16 = element type of the activations and packed weights (a `bool bf16` converts to the first two)
int launch_splitk_epilogue(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s) {
  const dim3 grid((p.cout_store + 63) / 64, (p.rows + 3) / 4), block(256);
#define ASV_SPLITK(ETV) do { if (out_f32) hipLaunchKernelGGL((splitk_epilogue_kernel<ETV, false>), grid, block, 0, s, p); \
                             else hipLaunchKernelGGL((splitk_epilogue_kernel<ETV, true>), grid, block, 0, s, p); } while (0)
  if (et == ET_BF16) ASV_SPLITK(ET_BF16);
  else if (et == ET_F16) ASV_SPLITK(ET_F16);
  else hipLaunchKernelGGL((splitk_epilogue_kernel<ET_F32, false>), grid, block, 0, s, p);
#undef ASV_SPLITK
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

namespace {
template <int ET>
int launch_tdnn_mfma_et(const TdnnKernelParams &p, bool out_f32, bool fast, int halo, dim3 grid, dim3 block, int m_tiles, int n_tiles, hipStream_t s) {
  if (halo > kHalo) {
    // wide-window instantiations (grid domains): plain epilogue, no second input
    ASV_REQUIRE(fast && p.x2 == nullptr, "tdnn: layers with |tap| > %d support the plain epilogue only", kHalo);
    if constexpr (ET != ET_F32) {
      ASV_REQUIRE(!out_f32, "tdnn: wide-window 16-bit layers produce 16-bit rows");
      if (halo <= kMidHalo) hipLaunchKernelGGL((tdnn_gemm_kernel<ET, true, false, kMidHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
      else hipLaunchKernelGGL((tdnn_gemm_kernel<ET, true, false, kWideHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
    } else {
      if (halo <= kMidHalo) hipLaunchKernelGGL((tdnn_gemm_kernel<ET_F32, false, false, kMidHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
      else hipLaunchKernelGGL((tdnn_gemm_kernel<ET_F32, false, false, kWideHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
    }
  } else if constexpr (ET != ET_F32) {
    if (out_f32) {
      if (fast) hipLaunchKernelGGL((tdnn_gemm_kernel<ET, false, false, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
      else hipLaunchKernelGGL((tdnn_gemm_kernel<ET, false, true, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
    } else {
      if (fast) hipLaunchKernelGGL((tdnn_gemm_kernel<ET, true, false, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
      else hipLaunchKernelGGL((tdnn_gemm_kernel<ET, true, true, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
    }
  } else {
    ASV_REQUIRE(out_f32, "tdnn: f32 activations always produce f32");
    if (fast) hipLaunchKernelGGL((tdnn_gemm_kernel<ET_F32, false, false, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
    else hipLaunchKernelGGL((tdnn_gemm_kernel<ET_F32, false, true, kHalo>), grid, block, 0, s, p, m_tiles, n_tiles);
  }
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}
}  // namespace

int launch_tdnn_mfma(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s) {
  ASV_REQUIRE(p.rows % BM == 0, "tdnn: rows %d not a multiple of %d", p.rows, BM);
  ASV_REQUIRE(p.cin_pad % kChanAlign == 0, "tdnn: cin_pad %d not a multiple of %d", p.cin_pad, kChanAlign);
  ASV_REQUIRE(p.n_taps >= 1 && p.n_taps <= ASV_MAX_TAPS, "tdnn: bad tap count %d", p.n_taps);
  int halo = 0;
  for (int t = 0; t < p.n_taps; ++t) halo = std::max(halo, std::abs(p.taps[t]));
  ASV_REQUIRE(halo <= kWideHalo, "tdnn: tap offset %d exceeds the widest supported window (%d rows)", halo, kWideHalo);
  const int m_tiles = p.rows / BM;
  const int n_tiles = round_up(p.cout_store, BN) / BN;
  const dim3 grid(m_tiles * n_tiles, p.ksplit > 1 ? p.ksplit : 1), block(256);
  if (p.ksplit > 1) ASV_REQUIRE(p.partial != nullptr && p.ld_partial >= p.cout_store, "tdnn: split-K needs a partial buffer");
  // the hot instantiations carry the short epilogue only; tanh / sigmoid / per-segment terms /
  // residual / "bn-relu" order go to the GENERIC ones
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr &&
                    p.seg_scale == nullptr && p.res == nullptr;
  if (et == ET_BF16) return launch_tdnn_mfma_et<ET_BF16>(p, out_f32, fast, halo, grid, block, m_tiles, n_tiles, s);
  if (et == ET_F16) return launch_tdnn_mfma_et<ET_F16>(p, out_f32, fast, halo, grid, block, m_tiles, n_tiles, s);
  return launch_tdnn_mfma_et<ET_F32>(p, out_f32, fast, halo, grid, block, m_tiles, n_tiles, s);
}

int launch_tdnn_ref(const TdnnKernelParams &p, int et, bool out_f32, hipStream_t s) {
  const dim3 grid((p.cout_store + 63) / 64, (p.rows + 3) / 4), block(256);
#define ASV_REF(ETV) do { if (out_f32) hipLaunchKernelGGL((tdnn_ref_kernel<ETV, false>), grid, block, 0, s, p); \
                          else hipLaunchKernelGGL((tdnn_ref_kernel<ETV, true>), grid, block, 0, s, p); } while (0)
  if (et == ET_BF16) ASV_REF(ET_BF16);
  else if (et == ET_F16) ASV_REF(ET_F16);
  else hipLaunchKernelGGL((tdnn_ref_kernel<ET_F32, false>), grid, block, 0, s, p);
#undef ASV_REF
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
