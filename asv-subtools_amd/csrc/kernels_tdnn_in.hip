// The input layer of a TDNN-family extractor as its own kernel: TdnnAffine + ReLU + eval BatchNorm (components.py:107-149,
// 410-431) straight from the caller's packed Kaldi feature matrix [sum T][D] f32 (libs/support/kaldi_io.py:466-496).
//
// Why: with D = 80 (fbank) and 5 taps the layer's whole K extent is 400 - the generic wide kernel (kernels_tdnn_v3.hip) walked it
// as two 64-channel chunks of a 4-stage LDS-DMA ring behind a separate packing pass (pack_input_kernel: f32 -> 16-bit rows with
// halo gaps, 13.6 us + 42 MB of traffic per C2 step) and ran at 0.27 of the matrix peak: prologue, two chunk barriers and epilogue
// around 200 matrix instructions per wave.  Here
//   * the window of a tile - 136 frames x D channels - is gathered ONCE, directly from the caller's f32 matrix through the
//     per-row source index (row_src: -1 = gap row = zeros), converted to the 16-bit element type in registers and written to
//     one LDS image (row pitch an odd multiple of 16 bytes: conflict-free 16-byte fragment reads); no packing pass, no staging ring;
//   * all taps read that image shifted by their offset: ONE workgroup barrier per tile, none in the K loop;
//   * weight fragments come from L2 a whole tap (NG k-groups = 8 NG matrix instructions) ahead, re-fetched in place;
//   * 128 frames x 256 channels per workgroup, 4 waves (128 x 64 each), two workgroups per CU; the epilogue is the wide kernel's
//     (bias -> ReLU -> folded BN, packed conversion, wave-private LDS transpose, 16-byte row-segment stores).
#include <cstdlib>

#include "device_utils.h"

namespace asv {
namespace {

constexpr int IBM = 128, IBN = 256;
constexpr int IWIN = IBM + 2 * kHalo;          // 136 window rows
constexpr int IMAX_PITCH = 208;                // 96 channels x 2 B + 16
constexpr int ISCR = 4 * 32 * 128;             // epilogue scratch per wave: [128 frames][64 channels] 16-bit = 16 KiB
constexpr int IN_LDS = 4 * ISCR + 3 * 256 * 4; // the image (<= 28 KiB) lives inside the scratch region
static_assert(IWIN * IMAX_PITCH <= 4 * ISCR, "the window image must fit in the epilogue scratch");
static_assert(IBN == kBigTileN, "weight padding must match the N tile");

__device__ __forceinline__ int iswz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// NG = 16-channel k-groups per tap = cin_pad / 16 (2 .. 6)
template <int ET, int NG>
__global__ __launch_bounds__(256, 2) void tdnn_input_kernel(const TdnnKernelParams p, const TdnnInputSource src, int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[IN_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int tile = xcd_swizzle(blockIdx.x, m_tiles * n_tiles);
  const int m0 = (tile / n_tiles) * IBM;
  const int n0 = (tile % n_tiles) * IBN;
  constexpr int CIN = NG * 16;
  constexpr int PITCH = CIN * 2 + ((NG & 1) ? 0 : 16);        // odd multiple of 16 bytes
  constexpr int PPR = CIN / 4;                                 // 16-byte f32 pieces per window row
  const int n_taps = p.n_taps;

  float *lds_par = reinterpret_cast<float *>(lds + 4 * ISCR);
  if (tid < 192) {
    const int which = tid >> 6, idx = (tid & 63) * 4;
    float4 v = (which == 1) ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *s = (which == 0) ? p.bias : (which == 1 ? p.scale : p.shift);
    if (s != nullptr) v = *reinterpret_cast<const float4 *>(s + n0 + idx);
    *reinterpret_cast<float4 *>(lds_par + which * 256 + idx) = v;
  }

  // ---- weight fragments of tap 0 first (oldest in the vector-memory queue), then the window gather
  const int nchunks = (CIN + 63) / 64;
  const size_t frag_stride = (size_t)n_taps * nchunks * 4096;            // bytes per 32-channel fragment
  const unsigned char *wf0 = reinterpret_cast<const unsigned char *>(p.wfrag) + (size_t)((n0 + wn * 64) / 32) * frag_stride + (size_t)lane * 16;
  const unsigned char *wf1 = wf0 + frag_stride;
  auto w_off = [&](int t, int g) -> size_t { return ((size_t)(t * nchunks + (g >> 2)) * 4 + (g & 3)) * 1024; };
  uint4 wf[NG][2];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    wf[g][0] = *reinterpret_cast<const uint4 *>(wf0 + w_off(0, g));
    wf[g][1] = *reinterpret_cast<const uint4 *>(wf1 + w_off(0, g));
  }
  {
    const int fd = src.feat_dim;
    const bool vec4 = (fd & 3) == 0;
    constexpr int ITEMS = IWIN * PPR;
#pragma unroll
    for (int it = 0; it < (ITEMS + 255) / 256; ++it) {
      const int item = it * 256 + tid;
      if (item < ITEMS) {
        const int w = item / PPR, pc = item - w * PPR;
        const int grow = m0 - kHalo + w;
        const int srow = (grow >= 0 && grow < p.rows) ? src.row_src[grow] : -1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (srow >= 0) {
          const float *f = src.feats + (size_t)srow * fd + pc * 4;
          if (vec4) {
            if (pc * 4 < fd) v = *reinterpret_cast<const float4 *>(f);
          } else {
            if (pc * 4 + 0 < fd) v.x = f[0];
            if (pc * 4 + 1 < fd) v.y = f[1];
            if (pc * 4 + 2 < fd) v.z = f[2];
            if (pc * 4 + 3 < fd) v.w = f[3];
          }
        }
        uint2 o;
        o.x = pack_h16x2<ET>(v.x, v.y);
        o.y = pack_h16x2<ET>(v.z, v.w);
        *reinterpret_cast<uint2 *>(lds + w * PITCH + pc * 8) = o;
      }
    }
  }
  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // the image is complete
  asm volatile("" ::: "memory");

  // ---- K loop: taps x NG k-groups, no barrier.  k-group (t, g): frame fragment i = rows i*32 + lr + HALO + d_t of the image,
  // 8 channels at byte (2 g + lh) * 16; its 8 matrix instructions carry the 4 image reads of the next k-group and the in-place
  // re-fetch of their weight fragments for the next tap.
  const int v_taps = p.taps[lane < 9 ? lane : 0];
  struct XF { uint4 x[4]; };
  auto x_addr = [&](int t) -> uint32_t { return (uint32_t)((lr + kHalo + __builtin_amdgcn_readlane(v_taps, t)) * PITCH + lh * 16); };
  auto load_x = [&](uint32_t xa, int g, int i, XF &f) { f.x[i] = *reinterpret_cast<const uint4 *>(lds + xa + g * 32 + i * 32 * PITCH); };
  XF xa_, xb_;
  uint32_t xa = x_addr(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_x(xa, 0, i, xa_);
#pragma unroll 1
  for (int t = 0; t < n_taps; ++t) {
    const int tn = (t + 1 < n_taps) ? t + 1 : t;       // the last tap re-fetches its own fragments (valid memory, never used)
    const uint32_t xan = x_addr(tn);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const XF &xc = (g & 1) ? xb_ : xa_;
      XF &xn = (g & 1) ? xa_ : xb_;
      const bool wrap = g + 1 == NG;
      const uint32_t xnext = wrap ? xan : xa;
      const int gn = wrap ? 0 : g + 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        load_x(xnext, gn, q, xn);
        const int j = q >> 1, i0 = (q & 1) * 2;
        acc[i0][j] = mfma16<ET>(wf[g][j], xc.x[i0], acc[i0][j]);
        acc[i0 + 1][j] = mfma16<ET>(wf[g][j], xc.x[i0 + 1], acc[i0 + 1][j]);
        if (q == 1) wf[g][0] = *reinterpret_cast<const uint4 *>(wf0 + w_off(tn, g));
        if (q == 3) wf[g][1] = *reinterpret_cast<const uint4 *>(wf1 + w_off(tn, g));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr ((NG & 1) != 0) {                      // an odd number of k-groups per tap swaps the roles of the two register sets
      XF tmp = xa_; xa_ = xb_; xb_ = tmp;
    }
    xa = xan;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // every wave is done reading the image: the epilogue reuses the LDS
  asm volatile("" ::: "memory");

  // ---- epilogue (kernels_tdnn_v3.hip): acc[i][j][r]: frame = m0 + i*32 + lr, channel = n0 + wn*64 + j*32 + 8*(r>>2) + 4*lh + (r&3)
  unsigned char *scr = lds + wn * ISCR;
  const float act_lo = (p.act1 == ASV_ACT_RELU) ? 0.0f : -INFINITY;
  uint32_t vmask = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) vmask |= ((p.row_valid[(m0 + i * 32) >> 5] >> lr) & 1u) << i;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int chl = wn * 64 + j * 32 + 8 * q + 4 * lh;
      const float4 b4 = *reinterpret_cast<const float4 *>(lds_par + chl);
      const float4 sc4 = *reinterpret_cast<const float4 *>(lds_par + 256 + chl);
      const float4 sh4 = *reinterpret_cast<const float4 *>(lds_par + 512 + chl);
      const float b[4] = {b4.x, b4.y, b4.z, b4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
      const int slot = j * 4 + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool valid = (vmask >> i) & 1u;
        const int frow = i * 32 + lr;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = tdnn_epilogue_fast(acc[i][j][q * 4 + e], b[e], act_lo, sc[e], sh[e], true);
        uint2 pk;
        pk.x = pack_h16x2<ET>(y[0], y[1]);
        pk.y = pack_h16x2<ET>(y[2], y[3]);
        pk.x = valid ? pk.x : 0u;                      // gap rows are zeros
        pk.y = valid ? pk.y : 0u;
        *reinterpret_cast<uint2 *>(scr + frow * 128 + iswz(frow, slot) * 16 + ((lh ^ (frow & 1)) * 8)) = pk;
      }
    }
  }
  {
    unsigned char *yg = reinterpret_cast<unsigned char *>(p.y);
    const size_t y_pitch = (size_t)p.ldy * 2;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int piece = it * 64 + lane, frow = piece >> 3, slot = piece & 7;
      uint4 v = *reinterpret_cast<const uint4 *>(scr + frow * 128 + iswz(frow, slot) * 16);
      if (frow & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      const int ch = n0 + wn * 64 + slot * 8;
      const int row = m0 + frow;
      if (ch < p.cout_store) *reinterpret_cast<uint4 *>(yg + (size_t)row * y_pitch + (size_t)ch * 2) = v;
    }
  }
}

}  // namespace

bool tdnn_input_supported(const TdnnKernelParams &p, int et) {
  const bool fast = (p.act1 == ASV_ACT_NONE || p.act1 == ASV_ACT_RELU) && p.act2 == ASV_ACT_NONE && !p.affine_first && p.seg_bias == nullptr;
  static const bool off = getenv("ASV_AMD_NO_INPUT_KERNEL") != nullptr && atoi(getenv("ASV_AMD_NO_INPUT_KERNEL")) != 0;
  return !off && et != ET_F32 && fast && p.wfrag != nullptr && p.x2 == nullptr && p.seg_scale == nullptr && p.res == nullptr && p.pool_partial == nullptr &&
         p.rows % IBM == 0 && p.cin_pad >= 32 && p.cin_pad <= 96 && p.cin_pad % 16 == 0 && p.cout_store >= 192 && p.cout_store % 8 == 0 && p.halo <= kHalo;
}

int launch_tdnn_input(const TdnnKernelParams &p, const TdnnInputSource &src, hipStream_t s) {
  ASV_REQUIRE(tdnn_input_supported(p, p.et) && src.feats != nullptr && src.row_src != nullptr && src.feat_dim >= 1 && src.feat_dim <= p.cin_pad,
              "tdnn(input): unsupported layer (cin_pad %d, cout %d, feature dim %d)", p.cin_pad, p.cout_store, src.feat_dim);
  const int m_tiles = p.rows / IBM, n_tiles = round_up(p.cout_store, IBN) / IBN;
  const dim3 grid(m_tiles * n_tiles), block(256);
#define ASV_IN(NGV) do { if (p.et == ET_F16) hipLaunchKernelGGL((tdnn_input_kernel<ET_F16, NGV>), grid, block, 0, s, p, src, m_tiles, n_tiles); \
                         else hipLaunchKernelGGL((tdnn_input_kernel<ET_BF16, NGV>), grid, block, 0, s, p, src, m_tiles, n_tiles); } while (0)
  switch (p.cin_pad / 16) {
    case 2: ASV_IN(2); break;
    case 3: ASV_IN(3); break;
    case 4: ASV_IN(4); break;
    case 5: ASV_IN(5); break;
    default: ASV_IN(6); break;
  }
#undef ASV_IN
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // namespace asv
