// C wrapper around the reference's own feature code (runtime/kaldifeat/csrc, compiled in place by oracle/Makefile.ref
// into oracle/_ref/libkaldifeat_ref.so - TEST INFRASTRUCTURE, build container only).  It exposes the fbank computation
// that runtime/extractor feeds the model with (feature_pipeline -> kaldifeat::Fbank), so the numpy restatement in
// oracle/fbank_oracle.py and the HIP front-end can be pinned to outputs of the reference itself.
#include <cstdint>
#include <cstring>

#include "kaldifeat/csrc/feature-fbank.h"
#include "kaldifeat/csrc/feature-mfcc.h"

namespace {
// options that came later (blackman coefficient, VTLN): set before a call, reset to the reference's defaults afterwards
struct Extras { float blackman_coeff = 0.42f, vtln_warp = 1.0f, vtln_low = 100.0f, vtln_high = -500.0f; };
Extras g_extras;
}  // namespace

extern "C" {

void kaldifeat_ref_set_extras(float blackman_coeff, float vtln_warp, float vtln_low, float vtln_high) {
  g_extras.blackman_coeff = blackman_coeff; g_extras.vtln_warp = vtln_warp; g_extras.vtln_low = vtln_low; g_extras.vtln_high = vtln_high;
}

// wave: n samples (float, in the int16 range like Kaldi's WaveData); out: [frames][num_bins] row-major, capacity cap_frames.
// Returns the number of frames, or -1 if out is too small.
int kaldifeat_ref_fbank(const float *wave, int n, float sample_rate, float frame_length_ms, float frame_shift_ms, float dither,
                        float preemph, int remove_dc_offset, const char *window_type, int round_to_power_of_two, int snip_edges,
                        int num_bins, float low_freq, float high_freq, int use_energy, float energy_floor, int raw_energy,
                        int htk_compat, int use_log_fbank, int use_power, float *out, int cap_frames) {
  kaldifeat::FbankOptions opts;
  opts.frame_opts.samp_freq = sample_rate;
  opts.frame_opts.frame_length_ms = frame_length_ms;
  opts.frame_opts.frame_shift_ms = frame_shift_ms;
  opts.frame_opts.dither = dither;
  opts.frame_opts.preemph_coeff = preemph;
  opts.frame_opts.remove_dc_offset = remove_dc_offset != 0;
  opts.frame_opts.window_type = window_type;
  opts.frame_opts.round_to_power_of_two = round_to_power_of_two != 0;
  opts.frame_opts.snip_edges = snip_edges != 0;
  opts.mel_opts.num_bins = num_bins;
  opts.mel_opts.low_freq = low_freq;
  opts.mel_opts.high_freq = high_freq;
  opts.frame_opts.blackman_coeff = g_extras.blackman_coeff;
  opts.mel_opts.vtln_low = g_extras.vtln_low;
  opts.mel_opts.vtln_high = g_extras.vtln_high;
  opts.use_energy = use_energy != 0;
  opts.energy_floor = energy_floor;
  opts.raw_energy = raw_energy != 0;
  opts.htk_compat = htk_compat != 0;
  opts.use_log_fbank = use_log_fbank != 0;
  opts.use_power = use_power != 0;
  kaldifeat::Fbank fbank(opts);
  torch::Tensor w = torch::from_blob(const_cast<float *>(wave), {n}, torch::kFloat).clone();
  torch::Tensor feats = fbank.ComputeFeatures(w, g_extras.vtln_warp).contiguous();
  const int frames = (int)feats.size(0), dim = (int)feats.size(1);
  if (frames > cap_frames) return -1;
  std::memcpy(out, feats.data_ptr<float>(), sizeof(float) * (size_t)frames * dim);
  return frames;
}


// kaldifeat::Mfcc with the same calling convention; out: [frames][num_ceps].
int kaldifeat_ref_mfcc(const float *wave, int n, float sample_rate, float frame_length_ms, float frame_shift_ms, float preemph,
                       int remove_dc_offset, const char *window_type, int round_to_power_of_two, int snip_edges, int num_bins,
                       float low_freq, float high_freq, int num_ceps, float cepstral_lifter, int use_energy, float energy_floor,
                       int raw_energy, int htk_compat, float *out, int cap_frames) {
  kaldifeat::MfccOptions opts;
  opts.frame_opts.samp_freq = sample_rate;
  opts.frame_opts.frame_length_ms = frame_length_ms;
  opts.frame_opts.frame_shift_ms = frame_shift_ms;
  opts.frame_opts.dither = 0.0f;
  opts.frame_opts.preemph_coeff = preemph;
  opts.frame_opts.remove_dc_offset = remove_dc_offset != 0;
  opts.frame_opts.window_type = window_type;
  opts.frame_opts.round_to_power_of_two = round_to_power_of_two != 0;
  opts.frame_opts.snip_edges = snip_edges != 0;
  opts.mel_opts.num_bins = num_bins;
  opts.mel_opts.low_freq = low_freq;
  opts.mel_opts.high_freq = high_freq;
  opts.frame_opts.blackman_coeff = g_extras.blackman_coeff;
  opts.mel_opts.vtln_low = g_extras.vtln_low;
  opts.mel_opts.vtln_high = g_extras.vtln_high;
  opts.num_ceps = num_ceps;
  opts.cepstral_lifter = cepstral_lifter;
  opts.use_energy = use_energy != 0;
  opts.energy_floor = energy_floor;
  opts.raw_energy = raw_energy != 0;
  opts.htk_compat = htk_compat != 0;
  kaldifeat::Mfcc mfcc(opts);
  torch::Tensor w = torch::from_blob(const_cast<float *>(wave), {n}, torch::kFloat).clone();
  torch::Tensor feats = mfcc.ComputeFeatures(w, g_extras.vtln_warp).contiguous();
  const int frames = (int)feats.size(0), dim = (int)feats.size(1);
  if (frames > cap_frames) return -1;
  std::memcpy(out, feats.data_ptr<float>(), sizeof(float) * (size_t)frames * dim);
  return frames;
}

}  // extern "C"
