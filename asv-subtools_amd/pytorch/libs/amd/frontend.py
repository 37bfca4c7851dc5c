# -*- coding:utf-8 -*-
"""Acoustic front-end on the MI355X: Kaldi-compatible log-mel filterbank features (+ per-utterance mean/variance
normalisation) of a batch of waveforms in one launch, through asv_fbank / asv_cmvn of libasv_amd.so.

Replaces the per-utterance torchaudio.compliance.kaldi.fbank loop of reference
pytorch/libs/egs/kaldi_features.py:107-137 (and kaldifeat::Fbank of runtime/kaldifeat/csrc); option names and defaults
below are torchaudio's, so a `kaldi_featset` dict of a reference config passes through unchanged.
"""

import ctypes as C

import numpy as np

from . import capi

# torchaudio.compliance.kaldi.fbank keyword -> (asv_fbank_opts_t field, default)
_DEFAULTS = dict(
    blackman_coeff=0.42, channel=-1, dither=0.0, energy_floor=1.0, frame_length=25.0, frame_shift=10.0, high_freq=0.0,
    htk_compat=False, low_freq=20.0, min_duration=0.0, num_mel_bins=23, preemphasis_coefficient=0.97, raw_energy=True,
    remove_dc_offset=True, round_to_power_of_two=True, sample_frequency=16000.0, snip_edges=True, subtract_mean=False,
    use_energy=False, use_log_fbank=True, use_power=True, vtln_high=-500.0, vtln_low=100.0, vtln_warp=1.0, window_type="povey")


# torchaudio.compliance.kaldi.mfcc: the same framing / mel options, no use_log_fbank / use_power, plus the cepstral ones
_MFCC_DEFAULTS = dict({k: v for k, v in _DEFAULTS.items() if k not in ("use_log_fbank", "use_power")}, num_ceps=13, cepstral_lifter=22.0)


def fbank_options(_kind="fbank", **kw):
    """A filled asv_fbank_opts_t from torchaudio-style keywords; refuses what the device path does not compute."""
    defaults = _DEFAULTS if _kind == "fbank" else _MFCC_DEFAULTS
    unknown = set(kw) - set(defaults)
    if unknown:
        raise TypeError("%s: unknown option(s) %s" % (_kind, sorted(unknown)))
    o = dict(defaults, **kw)
    o.setdefault("use_log_fbank", True)
    o.setdefault("use_power", True)
    o.setdefault("num_ceps", 0)
    o.setdefault("cepstral_lifter", 0.0)
    o["dim"] = o["num_ceps"] if _kind == "mfcc" else o["num_mel_bins"] + int(o["use_energy"])
    if _kind == "mfcc" and not 1 <= o["num_ceps"] <= o["num_mel_bins"]:
        raise ValueError("mfcc: num_ceps %d must be in [1, num_mel_bins = %d]" % (o["num_ceps"], o["num_mel_bins"]))
    if o["dither"] != 0.0:
        raise ValueError("fbank: dither is random noise and is not offered on the device path; set dither=0.0 (extraction configs do)")
    if o["window_type"] not in capi.WINDOW_TYPES:
        raise ValueError("fbank: unknown window_type %r" % (o["window_type"],))
    if o["channel"] not in (-1, 0):
        raise ValueError("fbank: pass one channel per utterance")
    opts = capi.FbankOpts()
    opts.struct_size = C.sizeof(capi.FbankOpts)
    opts.sample_rate = o["sample_frequency"]
    opts.frame_length_ms, opts.frame_shift_ms = o["frame_length"], o["frame_shift"]
    opts.preemph = o["preemphasis_coefficient"]
    opts.remove_dc_offset = int(o["remove_dc_offset"])
    opts.window_type = capi.WINDOW_TYPES[o["window_type"]]
    opts.round_to_power_of_two = int(o["round_to_power_of_two"])
    opts.snip_edges = int(o["snip_edges"])
    opts.num_bins = o["num_mel_bins"]
    opts.low_freq, opts.high_freq = o["low_freq"], o["high_freq"]
    opts.use_energy = int(o["use_energy"])
    opts.energy_floor = o["energy_floor"]
    opts.raw_energy = int(o["raw_energy"])
    opts.htk_compat = int(o["htk_compat"])
    opts.use_log_fbank = int(o["use_log_fbank"])
    opts.use_power = int(o["use_power"])
    opts.num_ceps = o["num_ceps"]
    opts.cepstral_lifter = o["cepstral_lifter"]
    opts.blackman_coeff = o["blackman_coeff"]
    opts.vtln_warp, opts.vtln_low, opts.vtln_high = o["vtln_warp"], o["vtln_low"], o["vtln_high"]
    return opts, o


def mfcc_options(**kw):
    return fbank_options("mfcc", **kw)


def num_frames(num_samples, **kw):
    opts, _ = fbank_options("fbank", **kw)
    n = capi.lib().asv_fbank_num_frames(C.byref(opts), int(num_samples))
    if n < 0:
        raise capi.AsvError("asv_fbank_num_frames: bad options")
    return int(n)


def _frame_counts(lens, o):
    """asv_fbank_num_frames for an array of lengths (feature-window.cc:71-114), vectorised."""
    lens = np.asarray(lens, dtype=np.int64)
    length = int(np.float32(o["sample_frequency"]) * np.float32(0.001) * np.float32(o["frame_length"]))
    shift = int(np.float32(o["sample_frequency"]) * np.float32(0.001) * np.float32(o["frame_shift"]))
    if o["snip_edges"]:
        return np.where(lens < length, 0, 1 + (lens - length) // shift)
    return (lens + shift // 2) // shift


def fbank_device(wave, sample_off, mean_norm=False, std_norm=False, eps=1e-10, kind="fbank", **kw):
    """wave: 1-D f32 or int16 (16-bit PCM) CUDA tensor holding all utterances back to back, sample_off: int64 numpy [n+1].
    Returns (feats [sum frames, dim] f32 CUDA tensor, frame_offsets int64 numpy [n+1]).  kind="mfcc": cepstra."""
    import torch
    opts, o = fbank_options(kind, **kw)
    lib = capi.lib()
    sample_off = np.ascontiguousarray(sample_off, dtype=np.int64)
    n = len(sample_off) - 1
    if n < 1 or wave.dim() != 1 or wave.dtype not in (torch.float32, torch.int16) or not wave.is_cuda or not wave.is_contiguous() or int(sample_off[-1]) != wave.shape[0]:
        raise ValueError("fbank_device: wave must be a contiguous 1-D f32 / int16 CUDA tensor of sample_off[-1] samples")
    entry = lib.asv_fbank if wave.dtype == torch.float32 else lib.asv_fbank_pcm16
    frame_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(_frame_counts(np.diff(sample_off), o), out=frame_off[1:])
    dim = o["dim"]
    feats = torch.empty((int(frame_off[-1]), dim), dtype=torch.float32, device=wave.device)
    stream = C.c_void_p(torch.cuda.current_stream(wave.device).cuda_stream)
    with torch.cuda.device(wave.device):
        capi.check(entry(C.byref(opts), C.c_void_p(wave.data_ptr()), sample_off.ctypes.data_as(C.POINTER(C.c_longlong)),
                         n, C.c_void_p(feats.data_ptr()), stream), "asv_fbank")
        if (o["subtract_mean"] or mean_norm or std_norm) and feats.shape[0] > 0:
            capi.check(lib.asv_cmvn(C.c_void_p(feats.data_ptr()), frame_off.ctypes.data_as(C.POINTER(C.c_longlong)), n, dim,
                                    int(bool(o["subtract_mean"] or mean_norm)), int(bool(std_norm)), float(eps), stream), "asv_cmvn")
    return feats, frame_off


def fbank_packed(waveforms, device=None, **kw):
    """waveforms: list of 1-D float arrays / tensors (host or device), samples in the int16 value range.
    Returns (feats [sum frames, dim] f32 CUDA tensor, frame_offsets int64 numpy [n+1]) - the packed layout
    asv_net_extract consumes, so features never visit the host."""
    import torch
    if len(waveforms) == 0:
        raise ValueError("fbank: empty batch")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    for w in waveforms:
        if w.ndim != 1:
            raise ValueError("fbank: every waveform must be 1-D (one channel), got shape %s" % (tuple(w.shape),))
    sample_off = np.zeros(len(waveforms) + 1, dtype=np.int64)
    np.cumsum([int(w.shape[0]) for w in waveforms], out=sample_off[1:])
    pcm16 = all((w.dtype == torch.int16) if isinstance(w, torch.Tensor) else (np.asarray(w).dtype == np.int16) for w in waveforms)
    dt = torch.int16 if pcm16 else torch.float32                      # 16-bit PCM is shipped and read as it is (half the ingest)
    if all(isinstance(w, torch.Tensor) and w.is_cuda for w in waveforms):
        wave = torch.cat([w.to(dt) for w in waveforms]) if len(waveforms) > 1 else waveforms[0].to(dt).contiguous()
    else:
        host = torch.empty(int(sample_off[-1]), dtype=dt).pin_memory()
        hv = host.numpy()
        for w, a, b in zip(waveforms, sample_off[:-1], sample_off[1:]):
            hv[a:b] = w.detach().cpu().numpy() if isinstance(w, torch.Tensor) else np.asarray(w)
        wave = host.to(dev, non_blocking=True)
    return fbank_device(wave, sample_off, **kw)


def fbank(waveforms, **kw):
    """List of per-utterance [frames, dim] CUDA tensors (views of one packed allocation)."""
    feats, off = fbank_packed(waveforms, **kw)
    return [feats[int(a):int(b)] for a, b in zip(off[:-1], off[1:])]


def mfcc(waveforms, **kw):
    """torchaudio.compliance.kaldi.mfcc for a batch: list of [frames, num_ceps] CUDA tensors."""
    return fbank(waveforms, kind="mfcc", **kw)


def _off(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_longlong))


def cmvn_sliding(feats, frame_off, cmn_window=600, min_window=100, center=False, norm_vars=False):
    """Kaldi apply-cmvn-sliding on packed device features (defaults are Kaldi's; the reference's extraction pipeline uses
    cmn_window=300, center=True: extract_xvectors_for_pytorch.sh:105-118).  Returns a new tensor."""
    import torch
    frame_off = _off(frame_off)
    out = torch.empty_like(feats)
    with torch.cuda.device(feats.device):
        capi.check(capi.lib().asv_cmvn_sliding(C.c_void_p(feats.data_ptr()), C.c_void_p(out.data_ptr()), _i64p(frame_off), len(frame_off) - 1,
                                               feats.shape[1], int(cmn_window), int(min_window), int(center), int(norm_vars),
                                               C.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)), "asv_cmvn_sliding")
    return out


def vad_energy(feats, frame_off, vad_energy_threshold=5.0, vad_energy_mean_scale=0.5, vad_frames_context=2, vad_proportion_threshold=0.12):
    """Energy VAD on column 0 (defaults: VadEnergyOptions of runtime/extractor/torch_asv_extractor.h:24-27).
    Returns (voiced uint8 CUDA tensor [frames], voiced_counts int64 numpy [n])."""
    import torch
    frame_off = _off(frame_off)
    n = len(frame_off) - 1
    voiced = torch.empty(feats.shape[0], dtype=torch.uint8, device=feats.device)
    counts = np.zeros(n, dtype=np.int64)
    with torch.cuda.device(feats.device):
        capi.check(capi.lib().asv_vad_energy(C.c_void_p(feats.data_ptr()), _i64p(frame_off), n, feats.shape[1], float(vad_energy_threshold),
                                             float(vad_energy_mean_scale), int(vad_frames_context), float(vad_proportion_threshold),
                                             C.c_void_p(voiced.data_ptr()), _i64p(counts),
                                             C.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)), "asv_vad_energy")
    return voiced, counts


def select_voiced(feats, voiced, frame_off, counts):
    """select-voiced-frames: (kept rows packed [sum counts, dim], new offsets int64 [n+1])."""
    import torch
    frame_off = _off(frame_off)
    out_off = np.zeros(len(frame_off), dtype=np.int64)
    np.cumsum(counts, out=out_off[1:])
    out = torch.empty((int(out_off[-1]), feats.shape[1]), dtype=torch.float32, device=feats.device)
    with torch.cuda.device(feats.device):
        capi.check(capi.lib().asv_select_frames(C.c_void_p(feats.data_ptr()), C.c_void_p(voiced.data_ptr()), _i64p(frame_off), _i64p(out_off),
                                                len(frame_off) - 1, feats.shape[1], C.c_void_p(out.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(feats.device).cuda_stream)), "asv_select_frames")
    return out, out_off
