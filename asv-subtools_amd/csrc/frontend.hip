// Kaldi-compatible log-mel filterbank front-end on the device: waveform samples ->
// [frames][num_bins (+ energy)] features, the matrices the extraction path starts from.  Replaces
// torchaudio.compliance.kaldi.fbank in reference pytorch/libs/egs/kaldi_features.py:72-137 and the kaldifeat C++ copy in
// reference runtime/kaldifeat/csrc (feature-window.cc, feature-fbank.cc, mel-computations.cc), whose float arithmetic the
// host-side tables below follow line by line (that code is compiled in place as the parity reference, oracle/Makefile.ref).
//
// Also here: MFCC output (feature-mfcc.cc), 16-bit PCM input, per-utterance and Kaldi sliding-window CMVN, the energy VAD of
// runtime/extractor/torch_asv_extractor.cc:14-62 and voiced-frame selection.
//
// Two feature kernels, one wave per frame in both:
//   fbank512_kernel  16 kHz windows (257..512 samples): samples, window, pre-emphasis and a radix-4 Stockham FFT in
//                    registers with three LDS exchanges, segmented mel filters (details at the kernel)
//   fbank_kernel     any other window up to 2048 samples: everything of a frame in LDS - gather the window (mirror-padded
//                    when snip_edges is off) -> mean removal -> raw log energy -> pre-emphasis -> window function -> zero
//                    padding -> in-place radix-2 complex FFT (bit-reversed load, log2(N) butterfly passes; a single wave
//                    needs no barrier between passes: its LDS operations complete in order) -> |X|^2 or |X| of bins
//                    0..N/2-1 -> triangular mel filters (lane = mel bin over its own range of FFT bins) -> log, floor
// HBM: 4 (2 for PCM16) bytes per sample in (frames overlap 2.5x, the re-reads hit L1/L2), 4 * dim bytes per frame out.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "asv_internal.h"

namespace asv {
namespace {

constexpr int kMaxFft = 2048;

struct FbankKernelParams {
  const void *wave;             // all utterances, packed: float or int16 samples (the kernels' SAMPLE type)
  const long long *sample_off;  // [n_utts + 1]
  const long long *frame_off;   // [n_utts + 1]
  int n_utts;
  long long total_frames;
  int length, shift, padded, log2n;
  float preemph;
  int remove_dc, snip_edges, use_energy, raw_energy, htk_compat, use_log, use_power;
  float log_energy_floor;       // -inf when there is no floor
  int num_bins;
  const float *window;          // [length]
  const float2 *twiddle;        // [padded / 2]: exp(-2 pi i k / padded)
  const float *mel_w;           // [num_bins][mel_stride] weights of FFT bins first[b] .. first[b] + count[b] - 1
  const int *mel_first, *mel_count;
  int mel_stride;
  // the same filters cut into segments of <= 8 consecutive FFT bins (fbank512_kernel: one lane per segment)
  const float *seg_w;           // [n_seg][8], zero padded
  const int *seg_first;         // [n_seg]
  const int *bin_seg;           // [num_bins + 1]: segments of bin b are bin_seg[b] .. bin_seg[b + 1] - 1
  int n_seg;
  int fpw;                      // consecutive frames one wave of fbank512_kernel walks
  int num_ceps;                 // > 0: MFCC output
  const float *dct;             // [num_ceps][num_bins], rows scaled by the lifter
  float c0_scale;               // sqrt 2 for htk_compat without energy, else 1
  float *out;                   // [total_frames][dim]
};

// Sum over the 64 lanes, the same value in every lane: four DPP steps inside each row of 16 (quad xor 1, quad xor 2,
// half-row mirror, row mirror), then the four row totals through v_readlane.
__device__ __forceinline__ float wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}


// Cepstra of one frame from its log mel energies in LDS (feature-mfcc.cc:118-148): lane c sums row c of the liftered DCT
// matrix; C0 is replaced by the log energy when asked for; htk_compat rotates C0 / the energy to the last column.
__device__ __forceinline__ void write_cepstra(const FbankKernelParams &p, const float *logmel, float log_energy, float *dst, int lane) {
  for (int c = lane; c < p.num_ceps; c += 64) {
    const float *row = p.dct + (size_t)c * p.num_bins;
    float acc = 0.0f;
    for (int b = 0; b < p.num_bins; ++b) acc = fmaf(logmel[b], row[b], acc);
    if (c == 0) acc = p.use_energy ? fmaxf(log_energy, p.log_energy_floor) : acc * p.c0_scale;
    dst[p.htk_compat ? (c == 0 ? p.num_ceps - 1 : c - 1) : c] = acc;
  }
}

template <typename SAMPLE>
__global__ __launch_bounds__(256) void fbank_kernel(const FbankKernelParams p) {
  extern __shared__ float lds[];                       // per wave: re[padded] | im[padded]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long frame = (long long)blockIdx.x * 4 + wave;
  if (frame >= p.total_frames) return;
  float *re = lds + (size_t)wave * 2 * p.padded, *im = re + p.padded;
  // which utterance: binary search in the frame offsets (wave-uniform)
  int lo = 0, hi = p.n_utts;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (p.frame_off[mid] <= frame) lo = mid; else hi = mid;
  }
  const long long s0 = p.sample_off[lo], ns = p.sample_off[lo + 1] - s0;
  const long long f = frame - p.frame_off[lo];
  long long first = f * p.shift;
  // The reference pads snip_edges=false input by mirroring (feature-window.cc:116-133): padded[i] for i < left is
  // wave[left - 1 - i]; past the end it is wave[n - 1 - (i - left - n)].  With left = (length - shift) / 2 the frame start
  // in the padded signal is f * shift, i.e. first sample index (unpadded) = f * shift - left.
  const long long left = (p.length - p.shift) / 2;
  if (!p.snip_edges) first = f * p.shift - left;
  float sum = 0.0f;
  for (int i = lane; i < p.length; i += 64) {
    long long k = first + i;
    if (k < 0) k = -k - 1;
    if (k >= ns) k = 2 * ns - 1 - k;
    const float v = (float)static_cast<const SAMPLE *>(p.wave)[s0 + k];
    re[i] = v;
    sum += v;
  }
  if (p.remove_dc) {
    const float mean = wave_sum(sum) / (float)p.length;
    for (int i = lane; i < p.length; i += 64) re[i] -= mean;
  }
  float log_energy = 0.0f;
  if (p.use_energy && p.raw_energy) {
    float e = 0.0f;
    for (int i = lane; i < p.length; i += 64) e += re[i] * re[i];
    log_energy = logf(fmaxf(wave_sum(e), 1.1920928955078125e-07f));
  }
  // pre-emphasis (needs the left neighbour: read everything first, then write) and window
  {
    float cur[kMaxFft / 64], prev[kMaxFft / 64];
    int n = 0;
    for (int i = lane; i < p.length; i += 64, ++n) { cur[n] = re[i]; prev[n] = i > 0 ? re[i - 1] : 0.0f; }
    n = 0;
    for (int i = lane; i < p.length; i += 64, ++n) {
      float y = cur[n];
      if (p.preemph != 0.0f) y = (i > 0) ? cur[n] - p.preemph * prev[n] : cur[n] * (1.0f - p.preemph);
      im[i] = y * p.window[i];                          // staged in `im`, moved to bit-reversed `re` below
    }
  }
  float e2 = 0.0f;
  if (p.use_energy && !p.raw_energy) {
    for (int i = lane; i < p.length; i += 64) e2 += im[i] * im[i];
    log_energy = logf(fmaxf(wave_sum(e2), 1.1920928955078125e-07f));
  }
  // bit-reversed copy into re (zero padding beyond the window), im <- 0 afterwards
  {
    float tmp[kMaxFft / 64];
    int n = 0;
    for (int i = lane; i < p.padded; i += 64, ++n) tmp[n] = i < p.length ? im[i] : 0.0f;
    n = 0;
    for (int i = lane; i < p.padded; i += 64, ++n) re[__brev((unsigned)i) >> (32 - p.log2n)] = tmp[n];
    for (int i = lane; i < p.padded; i += 64) im[i] = 0.0f;
  }
  // radix-2 decimation-in-time butterflies
  const int half = p.padded >> 1;
  for (int s = 0; s < p.log2n; ++s) {
    const int span = 1 << s;                            // butterfly distance
    for (int b = lane; b < half; b += 64) {
      const int j = b & (span - 1), base = ((b >> s) << (s + 1)) + j;
      const float2 w = p.twiddle[j << (p.log2n - 1 - s)];
      const float ar = re[base], ai = im[base], br = re[base + span], bi = im[base + span];
      const float tr = br * w.x - bi * w.y, ti = br * w.y + bi * w.x;
      re[base] = ar + tr; im[base] = ai + ti;
      re[base + span] = ar - tr; im[base + span] = ai - ti;
    }
  }
  // spectrum of bins 0 .. N/2 - 1 (the reference drops the Nyquist bin, feature-fbank.cc:66-68)
  for (int i = lane; i < half; i += 64) {
    const float pw = re[i] * re[i] + im[i] * im[i];
    re[i] = p.use_power ? pw : sqrtf(pw);
  }
  const int dim = p.num_ceps > 0 ? p.num_ceps : p.num_bins + (p.use_energy ? 1 : 0);
  float *dst = p.out + (size_t)frame * dim;
  const bool take_log = p.use_log || p.num_ceps > 0;
  for (int b = lane; b < p.num_bins; b += 64) {
    const float *w = p.mel_w + (size_t)b * p.mel_stride;
    const int i0 = p.mel_first[b], cnt = p.mel_count[b];
    float acc = 0.0f;
    for (int k = 0; k < cnt; ++k) acc = fmaf(re[i0 + k], w[k], acc);
    if (take_log) acc = logf(fmaxf(acc, 1.1920928955078125e-07f));
    if (p.num_ceps > 0) im[b] = acc;
    else dst[b + ((p.use_energy && !p.htk_compat) ? 1 : 0)] = acc;
  }
  if (p.num_ceps > 0) { write_cepstra(p, im, log_energy, dst, lane); return; }
  if (p.use_energy && lane == 0) dst[p.htk_compat ? p.num_bins : 0] = fmaxf(log_energy, p.log_energy_floor);
}


// ---- the 16 kHz kernel: 512-point frames (windows of 257..512 samples) -------------------------------------------------
// One wave per frame, a wave walks FPW consecutive frames (their windows overlap, the re-reads hit L1/L2).  The real
// 512-point transform is a 256-point complex one over (even, odd) sample pairs + a split step:
//   z[n] = x[2n] + i x[2n+1];  Z = FFT256(z);  X[k] = (Z[k] + Z*[256-k]) / 2  -  i W512^k (Z[k] - Z*[256-k]) / 2.
// FFT256 is a radix-4 Stockham: 64 lanes = 64 butterflies per pass, 4 passes; lane j holds points j + 64 r, which is
// both the order the samples are loaded in and the order the last pass leaves the bins in, so only the three exchanges
// between passes go through LDS (2 KiB per wave, in place: a wave's LDS operations complete in order).  Window values,
// pass twiddles and split twiddles depend on the lane only and live in registers across the frames of the wave.
struct cplx { float x, y; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ void dft4(cplx v[4]) {
  const cplx a0 = {v[0].x + v[2].x, v[0].y + v[2].y}, a1 = {v[0].x - v[2].x, v[0].y - v[2].y};
  const cplx a2 = {v[1].x + v[3].x, v[1].y + v[3].y}, a3 = {v[1].y - v[3].y, v[3].x - v[1].x};      // (v1 - v3) * -i
  v[0] = {a0.x + a2.x, a0.y + a2.y}; v[1] = {a1.x + a3.x, a1.y + a3.y};
  v[2] = {a0.x - a2.x, a0.y - a2.y}; v[3] = {a1.x - a3.x, a1.y - a3.y};
}
// orders a wave's LDS writes before its following LDS reads of other lanes' data (no instruction beyond the wait)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Exchange buffer indices (float2 slots).  A ds_write_b64 is served 16 consecutive lanes at a time over 32 banks, so the
// 16 slots of a lane group must differ mod 16; the reads (lane + 64 r) stay contiguous per 32 lanes under each padding.
//   pass 1 writes 4 j + r, pass 2 writes 16 (j / 4) + j % 4 + 4 r, pass 3 writes 64 (j / 16) + j % 16 + 16 r.
__device__ __forceinline__ int pad1(int i) { return i + (i >> 4); }          // 16 lanes: 4 j + (j >> 2) -> all residues mod 16
__device__ __forceinline__ int pad2(int i) { return i + ((i >> 4) << 2); }   // groups of 4 at 16 g -> 20 g: 0,4,8,12 mod 16
__device__ __forceinline__ int pad3(int i) { return i; }                     // 16 consecutive slots already
__device__ __forceinline__ cplx tw512(const float2 *t, int idx) {          // W512^idx for idx < 512 from the 256-entry table
  const float2 w = t[idx & 255];
  return (idx & 256) ? cplx{-w.x, -w.y} : cplx{w.x, w.y};
}

constexpr int kFastMaxSeg = 256;     // mel filter segments (<= 8 FFT bins each) the kernel has room for: 4 lanes-passes

template <bool MFCC, typename SAMPLE>
__global__ __launch_bounds__(256) void fbank512_kernel(const FbankKernelParams p) {
  __shared__ float2 xch[4][320];          // 256 points (padded to 316 slots); later |X|^2 [256] + 8 zeros + segment sums [256]
  __shared__ float4 segw[2][kFastMaxSeg];
  __shared__ int segk[kFastMaxSeg];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < p.n_seg; i += 256) {
    segw[0][i] = reinterpret_cast<const float4 *>(p.seg_w)[2 * i];
    segw[1][i] = reinterpret_cast<const float4 *>(p.seg_w)[2 * i + 1];
    segk[i] = p.seg_first[i];
  }
  __syncthreads();
  const long long frame0 = ((long long)blockIdx.x * 4 + wave) * p.fpw;
  if (frame0 >= p.total_frames) return;
  float2 *my = xch[wave];
  float *pw = reinterpret_cast<float *>(my);
  // lane constants
  float we[4], wo[4];
  cplx t2[3], t3[3], t4[3], ts[4];
  for (int r = 0; r < 4; ++r) {
    const int i = 2 * (lane + 64 * r);
    we[r] = i < p.length ? p.window[i] : 0.0f;
    wo[r] = i + 1 < p.length ? p.window[i + 1] : 0.0f;
    ts[r] = tw512(p.twiddle, lane + 64 * r);
  }
  for (int r = 1; r < 4; ++r) {
    t2[r - 1] = tw512(p.twiddle, (lane & 3) * r * 32);                    // W16^(k r) = W512^(32 k r)
    t3[r - 1] = tw512(p.twiddle, (lane & 15) * r * 8);                    // W64^(k r)
    t4[r - 1] = tw512(p.twiddle, lane * r * 2);                           // W256^(k r)
  }
  const int o2 = ((lane >> 2) << 4) + (lane & 3), o3 = ((lane >> 4) << 6) + (lane & 15);
  const int partner = (64 - lane) & 63;
  // utterance of the first frame
  int u = 0;
  {
    int hi = p.n_utts;
    while (hi - u > 1) {
      const int mid = (u + hi) >> 1;
      if (p.frame_off[mid] <= frame0) u = mid; else hi = mid;
    }
  }
  long long f_begin = p.frame_off[u], f_end = p.frame_off[u + 1], s0 = p.sample_off[u], ns = p.sample_off[u + 1] - s0;
  const long long left = p.snip_edges ? 0 : (p.length - p.shift) / 2;
  const int dim = MFCC ? p.num_ceps : p.num_bins + (p.use_energy ? 1 : 0);
  const bool take_log = MFCC || p.use_log;
  const float inv_len = 1.0f / (float)p.length;
  // the samples of frame `frame` (utterances are walked forwards) -> e[r] = x[2 (lane + 64 r)], o[r] = x[2 (lane + 64 r) + 1]
  auto fetch = [&](long long frame, float e[4], float o[4]) {
    while (frame >= f_end) {                                                // empty utterances have f_begin == f_end
      ++u;
      f_begin = f_end; f_end = p.frame_off[u + 1]; s0 = p.sample_off[u]; ns = p.sample_off[u + 1] - s0;
    }
    const long long first = (frame - f_begin) * p.shift - left;
    const SAMPLE *src = static_cast<const SAMPLE *>(p.wave) + s0;
    // loads are unconditional (indices clamped into the window), the values past the window are masked afterwards
    if (first >= 0 && first + p.length <= ns) {
      const SAMPLE *q = src + first;
      for (int r = 0; r < 4; ++r) {
        const int i = 2 * (lane + 64 * r);
        e[r] = (float)q[min(i, p.length - 1)];
        o[r] = (float)q[min(i + 1, p.length - 1)];
      }
    } else {                                                                // mirrored edges (snip_edges off)
      for (int r = 0; r < 4; ++r) {
        const int i = 2 * (lane + 64 * r);
        long long k0 = first + min(i, p.length - 1), k1 = first + min(i + 1, p.length - 1);
        k0 = k0 < 0 ? -k0 - 1 : k0;
        k0 = k0 >= ns ? 2 * ns - 1 - k0 : k0;
        k1 = k1 < 0 ? -k1 - 1 : k1;
        k1 = k1 >= ns ? 2 * ns - 1 - k1 : k1;
        e[r] = (float)src[k0];
        o[r] = (float)src[k1];
      }
    }
  };
  float en[4], on[4];                                                       // next frame's samples, in flight during this frame
  fetch(frame0, en, on);
  for (int it = 0; it < p.fpw; ++it) {
    const long long frame = frame0 + it;
    if (frame >= p.total_frames) break;
    float e[4], o[4];
    for (int r = 0; r < 4; ++r) {
      const int i = 2 * (lane + 64 * r);
      e[r] = i < p.length ? en[r] : 0.0f;
      o[r] = i + 1 < p.length ? on[r] : 0.0f;
    }
    if (it + 1 < p.fpw && frame + 1 < p.total_frames) fetch(frame + 1, en, on);
    if (p.remove_dc) {
      const float mean = wave_sum((e[0] + o[0]) + (e[1] + o[1]) + (e[2] + o[2]) + (e[3] + o[3])) * inv_len;
      for (int r = 0; r < 4; ++r) {
        const int i = 2 * (lane + 64 * r);
        e[r] = i < p.length ? e[r] - mean : 0.0f;
        o[r] = i + 1 < p.length ? o[r] - mean : 0.0f;
      }
    }
    float log_energy = 0.0f;
    if (p.use_energy && p.raw_energy) {
      float q = 0.0f;
      for (int r = 0; r < 4; ++r) q += e[r] * e[r] + o[r] * o[r];
      log_energy = logf(fmaxf(wave_sum(q), 1.1920928955078125e-07f));
    }
    cplx v[4];
    {
      // pre-emphasis: the left neighbour of even sample 2n is the odd sample of point n - 1 (lane - 1, or lane 63 of r - 1)
      float prev[4];
      for (int r = 0; r < 4; ++r) {
        const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o[r]), 0x138, 0xf, 0xf, false));   // wave_shr:1
        const float wrap = r > 0 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, o[r - 1]), 63)) : e[0];   // sample 0 is its own left neighbour (Kaldi)
        prev[r] = lane == 0 ? wrap : up;
      }
      for (int r = 0; r < 4; ++r) {
        const float ye = e[r] - p.preemph * prev[r], yo = o[r] - p.preemph * e[r];
        v[r] = {ye * we[r], yo * wo[r]};
      }
    }
    if (p.use_energy && !p.raw_energy) {
      float q = 0.0f;
      for (int r = 0; r < 4; ++r) q += v[r].x * v[r].x + v[r].y * v[r].y;
      log_energy = logf(fmaxf(wave_sum(q), 1.1920928955078125e-07f));
    }
    // pass 1 (Ns = 1, no twiddles)
    dft4(v);
    for (int r = 0; r < 4; ++r) my[pad1(4 * lane + r)] = make_float2(v[r].x, v[r].y);
    wave_lds_sync();
    // pass 2 (Ns = 4)
    for (int r = 0; r < 4; ++r) { const float2 t = my[pad1(lane + 64 * r)]; v[r] = {t.x, t.y}; }
    for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], t2[r - 1]);
    dft4(v);
    wave_lds_sync();
    for (int r = 0; r < 4; ++r) my[pad2(o2 + 4 * r)] = make_float2(v[r].x, v[r].y);
    wave_lds_sync();
    // pass 3 (Ns = 16)
    for (int r = 0; r < 4; ++r) { const float2 t = my[pad2(lane + 64 * r)]; v[r] = {t.x, t.y}; }
    for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], t3[r - 1]);
    dft4(v);
    wave_lds_sync();
    for (int r = 0; r < 4; ++r) my[pad3(o3 + 16 * r)] = make_float2(v[r].x, v[r].y);
    wave_lds_sync();
    // pass 4 (Ns = 64): bins lane + 64 r stay in registers
    for (int r = 0; r < 4; ++r) { const float2 t = my[pad3(lane + 64 * r)]; v[r] = {t.x, t.y}; }
    for (int r = 1; r < 4; ++r) v[r] = cmul(v[r], t4[r - 1]);
    dft4(v);
    wave_lds_sync();
    // split step: Z[256 - k] of bin k = lane + 64 r sits in lane (64 - lane) & 63 at 3 - r (lane 0: itself at (4 - r) & 3)
    for (int r = 0; r < 4; ++r) {
      const float qx = __shfl(v[3 - r].x, partner), qy = __shfl(v[3 - r].y, partner);
      const cplx self = v[(4 - r) & 3];
      const cplx zp = lane == 0 ? cplx{self.x, -self.y} : cplx{qx, -qy};    // conj Z[256 - k]
      const cplx ev = {0.5f * (v[r].x + zp.x), 0.5f * (v[r].y + zp.y)};
      const cplx d = {0.5f * (v[r].x - zp.x), 0.5f * (v[r].y - zp.y)};
      const cplx od = cmul(cplx{d.y, -d.x}, ts[r]);                          // -i d W512^k
      const float xr = ev.x + od.x, xi = ev.y + od.y;
      const float pwr = xr * xr + xi * xi;
      pw[lane + 64 * r] = (MFCC || p.use_power) ? pwr : sqrtf(pwr);                    // overwrites pass-4 inputs, already consumed
    }
    if (lane < 8) pw[256 + lane] = 0.0f;
    wave_lds_sync();
    // mel filters: one lane per segment of <= 8 FFT bins, partial sums to LDS, then one lane per mel bin adds its segments
    float *part = pw + 256 + 8;                                              // pw[256 .. 263] is read (times 0) by the last segments
    for (int sg = lane; sg < p.n_seg; sg += 64) {
      const float4 w0 = segw[0][sg], w1 = segw[1][sg];
      const float *q = pw + segk[sg];
      float acc = q[0] * w0.x;
      acc = fmaf(q[1], w0.y, acc); acc = fmaf(q[2], w0.z, acc); acc = fmaf(q[3], w0.w, acc);
      acc = fmaf(q[4], w1.x, acc); acc = fmaf(q[5], w1.y, acc); acc = fmaf(q[6], w1.z, acc); acc = fmaf(q[7], w1.w, acc);
      part[sg] = acc;
    }
    wave_lds_sync();
    float *dst = p.out + (size_t)frame * dim;
    for (int b = lane; b < p.num_bins; b += 64) {
      const int s_lo = p.bin_seg[b], s_hi = p.bin_seg[b + 1];
      float acc = part[s_lo];
      for (int sg = s_lo + 1; sg < s_hi; ++sg) acc += part[sg];
      if (take_log) acc = logf(fmaxf(acc, 1.1920928955078125e-07f));
      if (MFCC) pw[b] = acc;                                       // |X|^2 is consumed; num_bins <= 256 here
      else dst[b + ((p.use_energy && !p.htk_compat) ? 1 : 0)] = acc;
    }
    if (MFCC) {
      wave_lds_sync();
      write_cepstra(p, pw, log_energy, dst, lane);
    } else if (p.use_energy && lane == 0) dst[p.htk_compat ? p.num_bins : 0] = fmaxf(log_energy, p.log_energy_floor);
    wave_lds_sync();                                                         // the next frame's pass 1 overwrites pw
  }
}

// Per-utterance mean (and optionally variance) normalisation of every feature column, in place: reference
// kaldi_features.py:11-66 InputSequenceNormalization (torch.mean / unbiased torch.std over the frames, std floored at eps).
// One workgroup per (utterance, 64-column block); 4 row lanes x 64 columns, sums in f64.
__global__ __launch_bounds__(256) void cmvn_kernel(float *feats, const long long *frame_off, int dim, int mean_norm, int std_norm, float eps) {
  __shared__ double red[2][4][64];
  const int u = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), r = threadIdx.x >> 6;
  const long long f0 = frame_off[u], n = frame_off[u + 1] - f0;
  if (n <= 0) return;
  float *base = feats + (size_t)f0 * dim;
  double s = 0.0;
  if (c < dim) for (long long t = r; t < n; t += 4) s += (double)base[(size_t)t * dim + c];
  red[0][r][threadIdx.x & 63] = s;
  __syncthreads();
  const double mean = (red[0][0][threadIdx.x & 63] + red[0][1][threadIdx.x & 63] + red[0][2][threadIdx.x & 63] + red[0][3][threadIdx.x & 63]) / (double)n;
  double inv = 1.0;
  if (std_norm) {
    double q = 0.0;
    if (c < dim) for (long long t = r; t < n; t += 4) { const double d = (double)base[(size_t)t * dim + c] - mean; q += d * d; }
    red[1][r][threadIdx.x & 63] = q;
    __syncthreads();
    const double var = (red[1][0][threadIdx.x & 63] + red[1][1][threadIdx.x & 63] + red[1][2][threadIdx.x & 63] + red[1][3][threadIdx.x & 63]) / (double)(n > 1 ? n - 1 : 1);
    inv = 1.0 / fmax(sqrt(var), (double)eps);
  }
  const double m = mean_norm ? mean : 0.0;
  if (c < dim) for (long long t = r; t < n; t += 4) {
    float *v = base + (size_t)t * dim + c;
    *v = (float)(((double)*v - m) * inv);
  }
}


// Kaldi's sliding-window mean (/ variance) normalisation: the `apply-cmvn-sliding --cmn-window=300 --center=true` stage of
// the reference's offline pipeline (pytorch/pipeline/extract_xvectors_for_pytorch.sh:105-118).  Window rule of Kaldi's
// SlidingWindowCmnInternal (feat/feature-functions.cc; Kaldi itself is not vendored in the reference): centred windows
// [t - W/2, t - W/2 + W), causal ones [t - W, t + 1) grown to min_window, shifted to stay inside the utterance; f64 sums.
// One thread per (segment of 32 frames, column): the first window is summed directly, the following ones slide.
__device__ __forceinline__ void cmn_window_of(int t, int n, int window, int min_window, int center, int *ws, int *we) {
  int a, b;
  if (center) { a = t - window / 2; b = a + window; } else { a = t - window; b = t + 1; }
  if (a < 0) { b -= a; a = 0; }
  if (!center && b > t) b = max(t + 1, min_window);
  if (b > n) { a -= b - n; b = n; if (a < 0) a = 0; }
  *ws = a; *we = b;
}

constexpr int kSlideSeg = 32;
__global__ __launch_bounds__(256) void cmvn_sliding_kernel(const float *in, float *out, const long long *frame_off, int dim, int window,
                                                           int min_window, int center, int norm_vars) {
  const int u = blockIdx.x, c = blockIdx.z * 64 + (threadIdx.x & 63);
  const long long f0 = frame_off[u];
  const int n = (int)(frame_off[u + 1] - f0);
  const int seg = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int t0 = seg * kSlideSeg;
  if (t0 >= n || c >= dim) return;
  const float *x = in + (size_t)f0 * dim + c;
  float *y = out + (size_t)f0 * dim + c;
  int ws, we;
  cmn_window_of(t0, n, window, min_window, center, &ws, &we);
  double sum = 0.0, sumsq = 0.0;
  for (int k = ws; k < we; ++k) { const double v = x[(size_t)k * dim]; sum += v; sumsq += v * v; }
  const int t1 = min(n, t0 + kSlideSeg);
  for (int t = t0; t < t1; ++t) {
    int a, b;
    cmn_window_of(t, n, window, min_window, center, &a, &b);
    while (ws < a) { const double v = x[(size_t)ws * dim]; sum -= v; sumsq -= v * v; ++ws; }
    while (we < b) { const double v = x[(size_t)we * dim]; sum += v; sumsq += v * v; ++we; }
    const double frames = (double)(we - ws);
    double v = (double)x[(size_t)t * dim] - sum / frames;
    if (norm_vars) {
      if (we - ws == 1) v = 0.0;
      else {
        double var = sumsq / frames - (sum / frames) * (sum / frames);
        var = fmax(var, 1.0e-10);
        v *= 1.0 / sqrt(var);
      }
    }
    y[(size_t)t * dim] = (float)v;
  }
}

// Energy-based voice activity decision, reference runtime/extractor/torch_asv_extractor.cc:14-62 (= Kaldi
// compute-vad-decision): column 0 is the log energy; threshold = vad_energy_threshold + mean_scale * mean(log energy);
// a frame is voiced when at least proportion_threshold of the frames within +-context of it are above the threshold.
// One workgroup per utterance; the per-utterance voiced count goes to counts[u].
__global__ __launch_bounds__(256) void vad_energy_kernel(const float *feats, const long long *frame_off, int dim, float threshold, float mean_scale,
                                                         int context, float proportion, unsigned char *voiced, long long *counts) {
  __shared__ double red[256];
  __shared__ int cnt[256];
  const int u = blockIdx.x;
  const long long f0 = frame_off[u];
  const int n = (int)(frame_off[u + 1] - f0);
  const float *e = feats + (size_t)f0 * dim;
  double s = 0.0;
  for (int t = threadIdx.x; t < n; t += 256) s += (double)e[(size_t)t * dim];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  float thr = threshold;
  if (mean_scale != 0.0f && n > 0) thr += mean_scale * (float)red[0] / (float)n;
  int mine = 0;
  for (int t = threadIdx.x; t < n; t += 256) {
    int num = 0, den = 0;
    for (int t2 = t - context; t2 <= t + context; ++t2)
      if (t2 >= 0 && t2 < n) { ++den; num += e[(size_t)t2 * dim] > thr ? 1 : 0; }
    const unsigned char v = (float)num >= (float)den * proportion ? 1 : 0;
    voiced[f0 + t] = v;
    mine += v;
  }
  cnt[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) cnt[threadIdx.x] += cnt[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) counts[u] = cnt[0];
}

// Source row of every kept row (select-voiced-frames / index_select(nonzero(vad)), torch_asv_extractor.cc:104-108): one
// workgroup per utterance walks its flags 256 at a time, ranks by ballot + popcount, in frame order.
__global__ __launch_bounds__(256) void voiced_rows_kernel(const unsigned char *voiced, const long long *frame_off, const long long *out_off, long long *src_row) {
  __shared__ int wave_cnt[4];
  __shared__ int base_sh;
  const int u = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long f0 = frame_off[u];
  const int n = (int)(frame_off[u + 1] - f0);
  long long base = out_off[u];
  for (int t0 = 0; t0 < n; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const bool v = t < n && voiced[f0 + t] != 0;
    const unsigned long long m = __ballot(v);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wave_cnt[w];
    if (v) src_row[base + before + __popcll(m & ((1ull << lane) - 1))] = f0 + t;
    if (threadIdx.x == 0) base_sh = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
    base += base_sh;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float *in, const long long *src_row, long long rows, int dim, float *out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * dim) return;
  const long long r = i / dim;
  out[i] = in[src_row[r] * dim + (i - r * dim)];
}

float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

thread_local std::vector<void *> g_front_dev;            // device tables of the last option set (tiny; rebuilt on change)
thread_local asv_fbank_opts_t g_front_opts;
thread_local bool g_front_valid = false;
thread_local int g_front_device = -1;                     // the device the cached tables live on
// Host offset arrays are small and usually the same from call to call (fixed batch shapes): the last upload is kept per
// calling thread and reused when the contents are unchanged, which also removes the stream synchronisation a fresh upload
// needs (the source is a caller-owned / local array).  Calls of one thread are expected on one stream.
struct OffsetCache {
  std::vector<long long> host;
  long long *dev = nullptr;
  size_t cap = 0;
  int device = -1;
  int get(const long long *src, size_t count, hipStream_t s, const long long **out) {
    int cur = 0;
    ASV_HIP_CHECK(hipGetDevice(&cur));
    if (dev && device == cur && host.size() == count && memcmp(host.data(), src, count * 8) == 0) { *out = dev; return ASV_OK; }
    if (count > cap || device != cur) {
      if (dev) ASV_HIP_CHECK(hipFree(dev));
      dev = nullptr; cap = 0; host.clear();
      ASV_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dev), count * 16));
      cap = count * 2;
      device = cur;
    }
    host.assign(src, src + count);
    ASV_HIP_CHECK(hipMemcpyAsync(dev, host.data(), count * 8, hipMemcpyHostToDevice, s));
    ASV_HIP_CHECK(hipStreamSynchronize(s));
    *out = dev;
    return ASV_OK;
  }
};
thread_local OffsetCache g_fbank_offs, g_cmvn_offs;

struct FrontTables { float *dct; float *window; float2 *twiddle; float *mel_w; int *mel_first, *mel_count; int mel_stride; float *seg_w; int *seg_first, *bin_seg; int n_seg; };
thread_local FrontTables g_tab = {};

int window_size(const asv_fbank_opts_t &o, float ms) { return (int)(o.sample_rate * 0.001f * ms); }

}  // namespace
}  // namespace asv

using namespace asv;

extern "C" {

long long asv_fbank_num_frames(const asv_fbank_opts_t *o, long long num_samples) {
  if (!o || o->struct_size != sizeof(asv_fbank_opts_t)) return -1;
  const long long length = window_size(*o, o->frame_length_ms), shift = window_size(*o, o->frame_shift_ms);
  if (length <= 0 || shift <= 0) return -1;
  if (o->snip_edges) return num_samples < length ? 0 : 1 + (num_samples - length) / shift;
  return (num_samples + shift / 2) / shift;
}

static int fbank_entry(const asv_fbank_opts_t *o, const void *wave, bool pcm16, const long long *sample_offsets, int n_utts, float *feats, void *stream) {
  ASV_REQUIRE(o && o->struct_size == sizeof(asv_fbank_opts_t), "asv_fbank: struct_size mismatch");
  ASV_REQUIRE(wave && sample_offsets && feats && n_utts >= 1, "asv_fbank: bad argument");
  ASV_ON_OWNER(feats, "asv_fbank");
  ASV_REQUIRE(o->num_bins >= 3 && o->num_bins <= 512, "asv_fbank: num_bins %d", o->num_bins);
  ASV_REQUIRE(o->num_ceps >= 0 && o->num_ceps <= o->num_bins, "asv_fbank: num_ceps %d must not exceed num_bins %d", o->num_ceps, o->num_bins);
  ASV_REQUIRE(o->window_type >= ASV_WINDOW_POVEY && o->window_type <= ASV_WINDOW_BLACKMAN, "asv_fbank: window type %d", o->window_type);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int length = window_size(*o, o->frame_length_ms), shift = window_size(*o, o->frame_shift_ms);
  ASV_REQUIRE(length >= 2 && shift >= 1, "asv_fbank: frame length %d / shift %d samples", length, shift);
  int padded = length, log2n = 0;
  { int pw = 1; while (pw < length) { pw *= 2; } padded = pw; }
  ASV_REQUIRE(padded <= kMaxFft, "asv_fbank: windows of more than %d samples are not supported", kMaxFft);
  ASV_REQUIRE(o->round_to_power_of_two || padded == length, "asv_fbank: round_to_power_of_two=false needs a power-of-two window (got %d samples)", length);
  while ((1 << log2n) < padded) ++log2n;
  // ---- tables (host arithmetic in the reference's types: float mel scale, double window phase)
  int cur_dev = 0;
  ASV_HIP_CHECK(hipGetDevice(&cur_dev));                   // = the owner of `feats` (guard above)
  const double blackman = o->blackman_coeff != 0.0f ? (double)o->blackman_coeff : 0.42;      // 0 (a zero-filled struct) = Kaldi's default
  if (!g_front_valid || g_front_device != cur_dev || memcmp(&g_front_opts, o, sizeof(*o)) != 0) {
    for (void *ptr : g_front_dev) (void)hipFree(ptr);
    g_front_dev.clear();
    g_front_valid = false;
    std::vector<float> win(length);
    const double a = 6.283185307179586476925286766559005 / (length - 1);
    for (int i = 0; i < length; ++i) {
      const double x = (double)i;
      switch (o->window_type) {
        case ASV_WINDOW_HANNING: win[i] = (float)(0.5 - 0.5 * cos(a * x)); break;
        case ASV_WINDOW_SINE: win[i] = (float)sin(0.5 * a * x); break;
        case ASV_WINDOW_HAMMING: win[i] = (float)(0.54 - 0.46 * cos(a * x)); break;
        case ASV_WINDOW_RECTANGULAR: win[i] = 1.0f; break;
        case ASV_WINDOW_BLACKMAN: win[i] = (float)(blackman - 0.5 * cos(a * x) + (0.5 - blackman) * cos(2 * a * x)); break;
        default: win[i] = (float)pow(0.5 - 0.5 * cos(a * x), 0.85); break;          // povey
      }
    }
    std::vector<float2> tw(padded / 2);
    for (int k = 0; k < padded / 2; ++k) {
      const double ang = -6.283185307179586476925286766559005 * k / padded;
      tw[k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    // mel filters, mel-computations.cc:91-200; VTLN: the bin edges go through the piecewise-linear warp of :20-89
    const float nyquist = 0.5f * o->sample_rate;
    const float high = o->high_freq > 0.0f ? o->high_freq : nyquist + o->high_freq;
    ASV_REQUIRE(!(o->low_freq < 0.0f || o->low_freq >= nyquist || high <= 0.0f || high > nyquist || high <= o->low_freq),
                "asv_fbank: bad low-freq %g / high-freq %g vs nyquist %g", o->low_freq, high, nyquist);
    const float warp = o->vtln_warp == 0.0f ? 1.0f : o->vtln_warp;       // a zero-filled options block means "no warp"
    const float vt_low = o->vtln_low, vt_high = o->vtln_high < 0.0f ? o->vtln_high + nyquist : o->vtln_high;
    ASV_REQUIRE(warp == 1.0f || !(vt_low < 0.0f || vt_low <= o->low_freq || vt_low >= high || vt_high <= 0.0f || vt_high >= high || vt_high <= vt_low),
                "asv_fbank: bad vtln-low %g / vtln-high %g vs low-freq %g / high-freq %g", vt_low, vt_high, o->low_freq, high);
    auto warp_mel = [&](float mel) -> float {
      if (warp == 1.0f) return mel;
      const float freq = 700.0f * (expf(mel / 1127.0f) - 1.0f);
      float out = freq;
      if (!(freq < o->low_freq || freq > high)) {
        const float l = vt_low * std::max(1.0f, warp), h = vt_high * std::min(1.0f, warp), scale = 1.0f / warp;
        const float Fl = scale * l, Fh = scale * h;
        const float scale_left = (Fl - o->low_freq) / (l - o->low_freq), scale_right = (high - Fh) / (high - h);
        if (freq < l) out = o->low_freq + scale_left * (freq - o->low_freq);
        else if (freq < h) out = scale * freq;
        else out = high + scale_right * (freq - high);
      }
      return mel_scale(out);
    };
    const int n_fft = padded / 2;
    const float width = o->sample_rate / padded;
    const float mel_low = mel_scale(o->low_freq), mel_high = mel_scale(high);
    const float delta = (mel_high - mel_low) / (o->num_bins + 1);
    std::vector<int> first(o->num_bins, -1), count(o->num_bins, 0);
    std::vector<std::vector<float>> rows(o->num_bins);
    int stride = 1;
    for (int b = 0; b < o->num_bins; ++b) {
      const float left = warp_mel(mel_low + b * delta), center = warp_mel(mel_low + (b + 1) * delta), right = warp_mel(mel_low + (b + 2) * delta);
      int last = -1;
      std::vector<float> dense(n_fft, 0.0f);
      for (int i = 0; i < n_fft; ++i) {
        const float mel = mel_scale(width * i);
        if (mel > left && mel < right) {
          dense[i] = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
          if (first[b] < 0) first[b] = i;
          last = i;
        }
      }
      ASV_REQUIRE(first[b] >= 0, "asv_fbank: num_bins %d is too large for %d FFT bins", o->num_bins, n_fft);
      count[b] = last - first[b] + 1;
      rows[b].assign(dense.begin() + first[b], dense.begin() + last + 1);
      stride = std::max(stride, count[b]);
    }
    std::vector<float> melw((size_t)o->num_bins * stride, 0.0f);
    for (int b = 0; b < o->num_bins; ++b) memcpy(&melw[(size_t)b * stride], rows[b].data(), rows[b].size() * 4);
    auto up = [&](const void *h, size_t bytes, void **d) -> int {
      ASV_HIP_CHECK(hipMalloc(d, bytes));
      g_front_dev.push_back(*d);
      ASV_HIP_CHECK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
      return ASV_OK;
    };
    int rc;
    if ((rc = up(win.data(), win.size() * 4, reinterpret_cast<void **>(&g_tab.window)))) return rc;
    if ((rc = up(tw.data(), tw.size() * 8, reinterpret_cast<void **>(&g_tab.twiddle)))) return rc;
    if ((rc = up(melw.data(), melw.size() * 4, reinterpret_cast<void **>(&g_tab.mel_w)))) return rc;
    if ((rc = up(first.data(), first.size() * 4, reinterpret_cast<void **>(&g_tab.mel_first)))) return rc;
    if ((rc = up(count.data(), count.size() * 4, reinterpret_cast<void **>(&g_tab.mel_count)))) return rc;
    g_tab.mel_stride = stride;
    std::vector<float> segw;
    std::vector<int> segk, binseg(1, 0);
    for (int b = 0; b < o->num_bins; ++b) {
      for (int k = 0; k < count[b]; k += 8) {
        segk.push_back(first[b] + k);
        for (int t = 0; t < 8; ++t) segw.push_back(k + t < count[b] ? rows[b][k + t] : 0.0f);
      }
      binseg.push_back((int)segk.size());
    }
    if ((rc = up(segw.data(), segw.size() * 4, reinterpret_cast<void **>(&g_tab.seg_w)))) return rc;
    if ((rc = up(segk.data(), segk.size() * 4, reinterpret_cast<void **>(&g_tab.seg_first)))) return rc;
    if ((rc = up(binseg.data(), binseg.size() * 4, reinterpret_cast<void **>(&g_tab.bin_seg)))) return rc;
    g_tab.n_seg = (int)segk.size();
    g_tab.dct = nullptr;
    if (o->num_ceps > 0) {
      // matrix-functions.cc:15-43 (DCT-II rows, float normalisers, double cosine) x mel-computations.cc:203-212 (lifter)
      const int nb = o->num_bins;
      std::vector<float> dct((size_t)o->num_ceps * nb);
      for (int c = 0; c < o->num_ceps; ++c) {
        const float lift = o->cepstral_lifter != 0.0f ? (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * c / o->cepstral_lifter)) : 1.0f;
        for (int b = 0; b < nb; ++b) {
          const float v = c == 0 ? sqrtf(1.0f / nb) : sqrtf(2.0f / nb) * (float)cos(M_PI / nb * (b + 0.5) * c);
          dct[(size_t)c * nb + b] = v * lift;
        }
      }
      if ((rc = up(dct.data(), dct.size() * 4, reinterpret_cast<void **>(&g_tab.dct)))) return rc;
    }
    g_front_opts = *o;
    g_front_device = cur_dev;
    g_front_valid = true;
  }
  // ---- offsets
  std::vector<long long> offs(2 * (size_t)(n_utts + 1));
  long long total = 0;
  ASV_REQUIRE(sample_offsets[0] == 0, "asv_fbank: sample_offsets[0] must be 0");
  for (int u = 0; u <= n_utts; ++u) {
    offs[u] = sample_offsets[u];
    if (u > 0) ASV_REQUIRE(sample_offsets[u] >= sample_offsets[u - 1], "asv_fbank: sample offsets must not decrease");
    offs[n_utts + 1 + u] = total;
    if (u < n_utts) {
      const long long fr = asv_fbank_num_frames(o, sample_offsets[u + 1] - sample_offsets[u]);
      ASV_REQUIRE(!(fr > 0 && !o->snip_edges && sample_offsets[u + 1] - sample_offsets[u] < length), "asv_fbank: utterance %d is shorter than one window", u);
      total += fr;
    }
  }
  if (total == 0) return ASV_OK;
  const long long *d_offs = nullptr;
  { const int rc = g_fbank_offs.get(offs.data(), offs.size(), s, &d_offs); if (rc) return rc; }
  FbankKernelParams p;
  memset(&p, 0, sizeof(p));
  p.wave = wave; p.sample_off = d_offs; p.frame_off = d_offs + (n_utts + 1); p.n_utts = n_utts; p.total_frames = total;
  p.length = length; p.shift = shift; p.padded = padded; p.log2n = log2n; p.preemph = o->preemph;
  p.remove_dc = o->remove_dc_offset; p.snip_edges = o->snip_edges; p.use_energy = o->use_energy; p.raw_energy = o->raw_energy;
  p.htk_compat = o->htk_compat; p.use_log = o->use_log_fbank || o->num_ceps > 0; p.use_power = o->use_power || o->num_ceps > 0;
  p.log_energy_floor = o->energy_floor > 0.0f ? logf(o->energy_floor) : -INFINITY;
  p.num_bins = o->num_bins; p.window = g_tab.window; p.twiddle = g_tab.twiddle; p.mel_w = g_tab.mel_w;
  p.mel_first = g_tab.mel_first; p.mel_count = g_tab.mel_count; p.mel_stride = g_tab.mel_stride; p.out = feats;
  p.num_ceps = o->num_ceps; p.dct = g_tab.dct; p.c0_scale = (o->htk_compat && !o->use_energy) ? (float)M_SQRT2 : 1.0f;
  p.seg_w = g_tab.seg_w; p.seg_first = g_tab.seg_first; p.bin_seg = g_tab.bin_seg; p.n_seg = g_tab.n_seg;
  static const bool generic_only = getenv("ASV_AMD_FBANK_GENERIC") != nullptr;
  if (padded == 512 && g_tab.n_seg <= kFastMaxSeg && o->num_bins <= 256 && !generic_only) {
    // every wave gets the same number of consecutive frames: enough workgroups for 4 per CU, at least 4 frames each so
    // that the per-wave set-up (lane constants, utterance search) is shared
    static const int cus = [] {
      int dev = 0, n = 0;
      return (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }();
    const long long waves = (long long)cus * 4 * 4;
    p.fpw = (int)std::max<long long>(4, (total + waves - 1) / waves);
    const long long per_wg = 4LL * p.fpw;
    const dim3 grid((unsigned)((total + per_wg - 1) / per_wg));
    if (o->num_ceps > 0 && pcm16) hipLaunchKernelGGL((fbank512_kernel<true, short>), grid, dim3(256), 0, s, p);
    else if (o->num_ceps > 0) hipLaunchKernelGGL((fbank512_kernel<true, float>), grid, dim3(256), 0, s, p);
    else if (pcm16) hipLaunchKernelGGL((fbank512_kernel<false, short>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((fbank512_kernel<false, float>), grid, dim3(256), 0, s, p);
    ASV_HIP_CHECK(hipGetLastError());
    return ASV_OK;
  }
  const size_t lds_bytes = (size_t)4 * 2 * padded * 4;
  if (pcm16) hipLaunchKernelGGL(fbank_kernel<short>, dim3((unsigned)((total + 3) / 4)), dim3(256), lds_bytes, s, p);
  else hipLaunchKernelGGL(fbank_kernel<float>, dim3((unsigned)((total + 3) / 4)), dim3(256), lds_bytes, s, p);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}


int asv_fbank(const asv_fbank_opts_t *o, const float *wave, const long long *sample_offsets, int n_utts, float *feats, void *stream) {
  return fbank_entry(o, wave, false, sample_offsets, n_utts, feats, stream);
}

int asv_fbank_pcm16(const asv_fbank_opts_t *o, const short *wave, const long long *sample_offsets, int n_utts, float *feats, void *stream) {
  return fbank_entry(o, wave, true, sample_offsets, n_utts, feats, stream);
}

// uploads a host int64 array for the duration of one call
static int upload_offsets(const long long *host, size_t count, hipStream_t s, long long **dev) {
  ASV_HIP_CHECK(hipMallocAsync(reinterpret_cast<void **>(dev), count * 8, s));
  ASV_HIP_CHECK(hipMemcpyAsync(*dev, host, count * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));                  // the caller's array may be a temporary
  return ASV_OK;
}

int asv_cmvn_sliding(const float *feats, float *out, const long long *frame_offsets, int n_utts, int dim, int cmn_window, int min_window,
                     int center, int norm_vars, void *stream) {
  ASV_REQUIRE(feats && out && feats != out && frame_offsets && n_utts >= 1 && dim >= 1, "asv_cmvn_sliding: bad argument (in-place is not supported)");
  ASV_REQUIRE(cmn_window > 0 && (center || (min_window > 0 && min_window <= cmn_window)), "asv_cmvn_sliding: cmn_window %d / min_window %d", cmn_window, min_window);
  ASV_REQUIRE(frame_offsets[0] == 0, "asv_cmvn_sliding: frame_offsets[0] must be 0");
  ASV_ON_OWNER(feats, "asv_cmvn_sliding");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long long longest = 0;
  for (int u = 0; u < n_utts; ++u) longest = std::max(longest, frame_offsets[u + 1] - frame_offsets[u]);
  if (longest == 0) return ASV_OK;
  ASV_REQUIRE(longest < (1LL << 31), "asv_cmvn_sliding: utterance too long");
  long long *d = nullptr;
  int rc = upload_offsets(frame_offsets, (size_t)n_utts + 1, s, &d);
  if (rc) return rc;
  const unsigned segs = (unsigned)((longest + 4 * kSlideSeg - 1) / (4 * kSlideSeg));
  hipLaunchKernelGGL(cmvn_sliding_kernel, dim3((unsigned)n_utts, segs, (unsigned)((dim + 63) / 64)), dim3(256), 0, s, feats, out, d, dim, cmn_window,
                     min_window, center, norm_vars);
  ASV_HIP_CHECK(hipGetLastError());
  ASV_HIP_CHECK(hipFreeAsync(d, s));
  return ASV_OK;
}

int asv_vad_energy(const float *feats, const long long *frame_offsets, int n_utts, int dim, float energy_threshold, float energy_mean_scale,
                   int frames_context, float proportion_threshold, unsigned char *voiced, long long *voiced_counts, void *stream) {
  ASV_REQUIRE(feats && frame_offsets && voiced && voiced_counts && n_utts >= 1 && dim >= 1, "asv_vad_energy: bad argument");
  ASV_REQUIRE(energy_mean_scale >= 0.0f && frames_context >= 0 && proportion_threshold > 0.0f && proportion_threshold < 1.0f,
              "asv_vad_energy: mean scale %g / context %d / proportion %g", energy_mean_scale, frames_context, proportion_threshold);
  ASV_REQUIRE(frame_offsets[0] == 0, "asv_vad_energy: frame_offsets[0] must be 0");
  ASV_ON_OWNER(feats, "asv_vad_energy");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  long long *d = nullptr;
  ASV_HIP_CHECK(hipMallocAsync(reinterpret_cast<void **>(&d), ((size_t)2 * n_utts + 1) * 8, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d, frame_offsets, ((size_t)n_utts + 1) * 8, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(vad_energy_kernel, dim3((unsigned)n_utts), dim3(256), 0, s, feats, d, dim, energy_threshold, energy_mean_scale, frames_context,
                     proportion_threshold, voiced, d + n_utts + 1);
  ASV_HIP_CHECK(hipGetLastError());
  ASV_HIP_CHECK(hipMemcpyAsync(voiced_counts, d + n_utts + 1, (size_t)n_utts * 8, hipMemcpyDeviceToHost, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  ASV_HIP_CHECK(hipFreeAsync(d, s));
  return ASV_OK;
}

int asv_select_frames(const float *feats, const unsigned char *voiced, const long long *frame_offsets, const long long *out_offsets, int n_utts,
                      int dim, float *out, void *stream) {
  ASV_REQUIRE(feats && voiced && frame_offsets && out_offsets && out && n_utts >= 1 && dim >= 1, "asv_select_frames: bad argument");
  ASV_REQUIRE(frame_offsets[0] == 0 && out_offsets[0] == 0, "asv_select_frames: offsets must start at 0");
  ASV_ON_OWNER(feats, "asv_select_frames");
  const long long rows = out_offsets[n_utts];
  if (rows == 0) return ASV_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  std::vector<long long> both(2 * ((size_t)n_utts + 1));
  memcpy(both.data(), frame_offsets, ((size_t)n_utts + 1) * 8);
  memcpy(both.data() + n_utts + 1, out_offsets, ((size_t)n_utts + 1) * 8);
  long long *d = nullptr;
  ASV_HIP_CHECK(hipMallocAsync(reinterpret_cast<void **>(&d), (both.size() + (size_t)rows) * 8, s));
  ASV_HIP_CHECK(hipMemcpyAsync(d, both.data(), both.size() * 8, hipMemcpyHostToDevice, s));
  ASV_HIP_CHECK(hipStreamSynchronize(s));
  long long *src_row = d + both.size();
  hipLaunchKernelGGL(voiced_rows_kernel, dim3((unsigned)n_utts), dim3(256), 0, s, voiced, d, d + n_utts + 1, src_row);
  const long long elems = rows * dim;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, feats, src_row, rows, dim, out);
  ASV_HIP_CHECK(hipGetLastError());
  ASV_HIP_CHECK(hipFreeAsync(d, s));
  return ASV_OK;
}

int asv_cmvn(float *feats, const long long *frame_offsets, int n_utts, int dim, int mean_norm, int std_norm, float eps, void *stream) {
  ASV_REQUIRE(feats && frame_offsets && n_utts >= 1 && dim >= 1, "asv_cmvn: bad argument");
  ASV_REQUIRE(frame_offsets[0] == 0, "asv_cmvn: frame_offsets[0] must be 0");
  ASV_ON_OWNER(feats, "asv_cmvn");
  if (!mean_norm && !std_norm) return ASV_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long *d = nullptr;
  { const int rc = g_cmvn_offs.get(frame_offsets, (size_t)n_utts + 1, s, &d); if (rc) return rc; }
  hipLaunchKernelGGL(cmvn_kernel, dim3((unsigned)n_utts, (unsigned)((dim + 63) / 64)), dim3(256), 0, s, feats, d, dim, mean_norm, std_norm, eps);
  ASV_HIP_CHECK(hipGetLastError());
  return ASV_OK;
}

}  // extern "C"
