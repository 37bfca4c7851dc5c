"""numpy restatement of the Kaldi-compatible log-mel filterbank front-end (TEST INFRASTRUCTURE - see oracle/__init__.py).

SURVEY.md section 8(f) rank 2: the step in front of the extraction path.  The reference computes it with
torchaudio.compliance.kaldi.fbank in Python (pytorch/libs/egs/kaldi_features.py:72-137; torchaudio is absent here) and
with its C++ copy of kaldifeat in the runtime (runtime/kaldifeat/csrc).  This file follows the C++ sources, which ARE
in the reference tree and are compiled in place into oracle/_ref/libkaldifeat_ref.so (oracle/Makefile.ref) to pin it:
  framing / snip_edges                runtime/kaldifeat/csrc/feature-window.cc:60-133
  dc removal, raw log energy          runtime/kaldifeat/csrc/feature-common-inl.h:32-47
  pre-emphasis                        runtime/kaldifeat/csrc/feature-window.cc:150-170
  window functions (povey, ...)       runtime/kaldifeat/csrc/feature-window.cc:21-51
  |rfft|^2 without the Nyquist bin    runtime/kaldifeat/csrc/feature-fbank.cc:63-72
  mel filterbank matrix               runtime/kaldifeat/csrc/mel-computations.cc:60-141 (MelScale = 1127 ln(1 + f / 700))
  log with floor FLT_EPSILON          runtime/kaldifeat/csrc/feature-fbank.cc:74-77
Dither must be 0 (the only random step).
"""

import numpy as np

FLT_EPS = np.float32(1.1920928955078125e-07)


def window_size(sample_rate, frame_length_ms):
    return int(sample_rate * 0.001 * frame_length_ms)


def padded_size(n, round_to_power_of_two=True):
    if not round_to_power_of_two:
        return n
    p = 1
    while p < n:
        p *= 2
    return p


def num_frames(num_samples, sample_rate=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, snip_edges=True):
    length, shift = window_size(sample_rate, frame_length_ms), window_size(sample_rate, frame_shift_ms)
    if snip_edges:
        return 0 if num_samples < length else 1 + (num_samples - length) // shift
    return (num_samples + shift // 2) // shift


def window_function(n, kind="povey", blackman_coeff=0.42):
    a = 2.0 * np.pi / (n - 1)
    i = np.arange(n, dtype=np.float64)
    if kind == "hanning":
        w = 0.5 - 0.5 * np.cos(a * i)
    elif kind == "sine":
        w = np.sin(0.5 * a * i)
    elif kind == "hamming":
        w = 0.54 - 0.46 * np.cos(a * i)
    elif kind == "povey":
        w = np.power(0.5 - 0.5 * np.cos(a * i), 0.85)
    elif kind == "rectangular":
        w = np.ones(n)
    elif kind == "blackman":
        w = blackman_coeff - 0.5 * np.cos(a * i) + (0.5 - blackman_coeff) * np.cos(2 * a * i)
    else:
        raise ValueError("Invalid window type " + kind)
    return w.astype(np.float32)


def mel_scale(f):
    return np.float32(1127.0) * np.log(np.float32(1.0) + np.asarray(f, dtype=np.float32) / np.float32(700.0), dtype=np.float32)


def inverse_mel_scale(m):
    return np.float32(700.0) * (np.exp(np.float32(m) / np.float32(1127.0), dtype=np.float32) - np.float32(1.0))


def vtln_warp_freq(vtln_low, vtln_high, low_freq, high_freq, warp, freq):
    """mel-computations.cc:20-79: continuous piecewise-linear warp with F(low) = low, F(high) = high, slope 1/warp between
    the inflection points l = vtln_low * max(1, warp) and h = vtln_high * min(1, warp); float arithmetic like the reference."""
    f32 = np.float32
    freq, low_freq, high_freq, warp = f32(freq), f32(low_freq), f32(high_freq), f32(warp)
    if freq < low_freq or freq > high_freq:
        return freq
    l = f32(vtln_low) * max(f32(1.0), warp)
    h = f32(vtln_high) * min(f32(1.0), warp)
    scale = f32(1.0) / warp
    Fl, Fh = scale * l, scale * h
    scale_left = (Fl - low_freq) / (l - low_freq)
    scale_right = (high_freq - Fh) / (high_freq - h)
    if freq < l:
        return low_freq + scale_left * (freq - low_freq)
    if freq < h:
        return scale * freq
    return high_freq + scale_right * (freq - high_freq)


def mel_banks(num_bins, padded, sample_rate=16000.0, low_freq=20.0, high_freq=0.0, vtln_warp=1.0, vtln_low=100.0, vtln_high=-500.0):
    """[padded / 2, num_bins] float32 triangular filters on the mel scale (mel-computations.cc:91-200); with vtln_warp != 1 the
    three edges of every bin go through the VTLN warp (:129-161)."""
    nyquist = 0.5 * sample_rate
    high = high_freq if high_freq > 0.0 else nyquist + high_freq
    if low_freq < 0.0 or low_freq >= nyquist or high <= 0.0 or high > nyquist or high <= low_freq:
        raise ValueError("Bad values in options: low-freq %s and high-freq %s vs. nyquist %s" % (low_freq, high, nyquist))
    vt_high = vtln_high + nyquist if vtln_high < 0.0 else vtln_high
    if vtln_warp != 1.0 and (vtln_low < 0.0 or vtln_low <= low_freq or vtln_low >= high or vt_high <= 0.0 or vt_high >= high or vt_high <= vtln_low):
        raise ValueError("Bad values in options: vtln-low %s and vtln-high %s, versus low-freq %s and high-freq %s" % (vtln_low, vt_high, low_freq, high))

    def warp_mel(m):
        if vtln_warp == 1.0:
            return np.float32(m)
        return mel_scale(vtln_warp_freq(vtln_low, vt_high, low_freq, high, vtln_warp, inverse_mel_scale(m)))
    n_fft = padded // 2
    width = np.float32(sample_rate / padded)
    mel_low, mel_high = mel_scale(low_freq), mel_scale(high)
    delta = np.float32((mel_high - mel_low) / np.float32(num_bins + 1))
    mel = mel_scale(width * np.arange(n_fft, dtype=np.float32))
    out = np.zeros((n_fft, num_bins), dtype=np.float32)
    for b in range(num_bins):
        left = warp_mel(np.float32(mel_low + np.float32(b) * delta))
        center = warp_mel(np.float32(mel_low + np.float32(b + 1) * delta))
        right = warp_mel(np.float32(mel_low + np.float32(b + 2) * delta))
        inside = (mel > left) & (mel < right)
        if not inside.any():
            raise ValueError("You may have set num_mel_bins too large.")
        up = (mel - left) / (center - left)
        down = (right - mel) / (right - center)
        out[:, b] = np.where(inside, np.where(mel <= center, up, down), np.float32(0.0))
    return out


def fbank(wave, sample_rate=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, remove_dc_offset=True, window_type="povey",
          round_to_power_of_two=True, snip_edges=True, num_bins=23, low_freq=20.0, high_freq=0.0, use_energy=False, energy_floor=0.0,
          raw_energy=True, htk_compat=False, use_log_fbank=True, use_power=True, blackman_coeff=0.42, vtln_warp=1.0, vtln_low=100.0,
          vtln_high=-500.0, dtype=np.float32):
    """wave: 1-D samples in the int16 value range (Kaldi WaveData convention).  Returns [frames, num_bins (+1)]."""
    wave = np.asarray(wave, dtype=np.float32)
    length, shift = window_size(sample_rate, frame_length_ms), window_size(sample_rate, frame_shift_ms)
    n = num_frames(len(wave), sample_rate, frame_length_ms, frame_shift_ms, snip_edges)
    dim = num_bins + (1 if use_energy else 0)
    if n == 0:
        return np.zeros((0, dim), dtype=np.float32)
    if not snip_edges:
        pad = (n - 1) * shift + length - len(wave)
        left = (length - shift) // 2
        right = pad - left
        wave = np.concatenate([wave[:left][::-1], wave, wave[len(wave) - right:][::-1] if right > 0 else wave[:0]])
    idx = np.arange(n)[:, None] * shift + np.arange(length)[None, :]
    frames = wave[idx].astype(dtype)
    if remove_dc_offset:
        frames = frames - frames.mean(axis=1, keepdims=True, dtype=dtype)
    log_energy = None
    if use_energy and raw_energy:
        log_energy = np.log(np.maximum((frames * frames).sum(axis=1, dtype=dtype), FLT_EPS))
    if preemph != 0.0:
        out = np.empty_like(frames)
        out[:, 1:] = frames[:, 1:] - dtype(preemph) * frames[:, :-1]
        out[:, 0] = frames[:, 0] * dtype(1.0 - np.float32(preemph)) if dtype == np.float32 else frames[:, 0] * (1.0 - preemph)
        frames = out
    frames = frames * window_function(length, window_type, blackman_coeff).astype(dtype)[None, :]
    padded = padded_size(length, round_to_power_of_two)
    if padded > length:
        frames = np.concatenate([frames, np.zeros((n, padded - length), dtype=dtype)], axis=1)
    if use_energy and not raw_energy:
        log_energy = np.log(np.maximum((frames * frames).sum(axis=1, dtype=dtype), FLT_EPS))
    spec = np.abs(np.fft.rfft(frames.astype(np.float64), axis=1))[:, :-1]
    if use_power:
        spec = spec * spec
    mel = spec.astype(dtype) @ mel_banks(num_bins, padded, sample_rate, low_freq, high_freq, vtln_warp, vtln_low, vtln_high).astype(dtype)
    if use_log_fbank:
        mel = np.log(np.maximum(mel, FLT_EPS))
    if use_energy:
        if energy_floor > 0.0:
            log_energy = np.maximum(log_energy, np.log(np.float32(energy_floor)))
        mel = np.concatenate([mel, log_energy[:, None]] if htk_compat else [log_energy[:, None], mel], axis=1)
    return mel.astype(np.float32)


def dct_matrix(num_ceps, num_bins):
    """Rows 0..num_ceps-1 of the orthonormal DCT-II (reference matrix-functions.cc:15-43: float normalisers, double cosine)."""
    out = np.empty((num_ceps, num_bins), dtype=np.float32)
    out[0, :] = np.sqrt(np.float32(1.0) / np.float32(num_bins))
    norm = np.sqrt(np.float32(2.0) / np.float32(num_bins))
    for r in range(1, num_ceps):
        out[r, :] = norm * np.cos(np.pi / num_bins * (np.arange(num_bins) + 0.5) * r).astype(np.float32)
    return out


def lifter_coeffs(q, n):
    """1 + Q/2 sin(pi i / Q) (reference mel-computations.cc:203-212)."""
    return (1.0 + 0.5 * q * np.sin(np.pi * np.arange(n) / q)).astype(np.float32)


def mfcc(wave, num_ceps=13, cepstral_lifter=22.0, use_energy=True, energy_floor=0.0, raw_energy=True, htk_compat=False, num_bins=23, **kw):
    """Reference feature-mfcc.cc:78-150: log mel energies of the power spectrum -> DCT -> lifter; C0 replaced by the log
    energy (use_energy); htk_compat rolls C0 / the energy to the last column (C0 times sqrt 2 when it is not the energy)."""
    both = fbank(wave, num_bins=num_bins, use_energy=True, energy_floor=energy_floor, raw_energy=raw_energy, htk_compat=False,
                 use_log_fbank=True, use_power=True, **kw)
    log_energy, logmel = both[:, 0], both[:, 1:]
    feats = logmel @ dct_matrix(num_ceps, num_bins).T
    if cepstral_lifter != 0.0:
        feats = feats * lifter_coeffs(cepstral_lifter, num_ceps)[None, :]
    if use_energy:
        feats[:, 0] = log_energy
    if htk_compat:
        feats = np.roll(feats, -1, axis=1)
        if not use_energy:
            feats[:, -1] *= np.float32(np.sqrt(2.0))
    return feats.astype(np.float32)


def sliding_cmn(x, cmn_window=600, min_window=100, center=False, norm_vars=False):
    """Kaldi SlidingWindowCmnInternal (feat/feature-functions.cc of Kaldi 5.5; Kaldi is NOT vendored in /root/reference, whose
    pipeline only calls the binary: extract_xvectors_for_pytorch.sh:105-118) - PARITY UNPINNED: restated from the published
    algorithm, no reference-side fixture exists.  Double precision like Kaldi."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    out = np.empty_like(x)
    for t in range(n):
        if center:
            ws = t - cmn_window // 2
            we = ws + cmn_window
        else:
            ws, we = t - cmn_window, t + 1
        if ws < 0:
            we -= ws
            ws = 0
        if not center and we > t:
            we = max(t + 1, min_window)
        if we > n:
            ws -= we - n
            we = n
            ws = max(ws, 0)
        w = x[ws:we]
        v = x[t] - w.mean(axis=0)
        if norm_vars:
            if we - ws == 1:
                v = np.zeros_like(v)
            else:
                var = np.maximum((w * w).mean(axis=0) - w.mean(axis=0) ** 2, 1.0e-10)
                v = v / np.sqrt(var)
        out[t] = v
    return out.astype(np.float32)


def vad_energy(feats, vad_energy_threshold=5.0, vad_energy_mean_scale=0.5, vad_frames_context=2, vad_proportion_threshold=0.12):
    """Reference runtime/extractor/torch_asv_extractor.cc:14-62, restated.  Pinned to the reference's own decisions:
    tests/golden/vad.npz comes from that file compiled in place (oracle/Makefile.ref, oracle/gen_vad_golden.py).
    Returns a 0/1 vector."""
    e = np.asarray(feats, dtype=np.float32)[:, 0]
    n = len(e)
    thr = np.float32(vad_energy_threshold)
    if vad_energy_mean_scale != 0.0:
        thr = np.float32(thr + np.float32(vad_energy_mean_scale) * np.float32(e.sum(dtype=np.float64)) / np.float32(n))
    out = np.zeros(n, dtype=np.uint8)
    above = e > thr
    for t in range(n):
        lo, hi = max(0, t - vad_frames_context), min(n, t + vad_frames_context + 1)
        out[t] = 1 if above[lo:hi].sum() >= np.float32(hi - lo) * np.float32(vad_proportion_threshold) else 0
    return out
