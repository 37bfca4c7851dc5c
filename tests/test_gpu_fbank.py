"""asv_fbank / asv_cmvn on the MI355X against the reference's kaldifeat outputs (tests/golden/fbank.npz) and the numpy
oracle; batching, ragged lengths, mean/variance normalisation, and waveform -> embedding without a host round trip."""

import numpy as np
import pytest

import helpers
from oracle import fbank_oracle

pytestmark = pytest.mark.gpu

# Absolute, on log-mel values of ~10..25, i.e. max|a-b| / max|b| < 1e-4 (the north star's relative-fp32 bar).  Both
# sides run an f32 FFT, so a weak bin next to strong harmonics carries ~1e-7 * (strongest / weakest amplitude) of
# rounding noise on either side (the numpy oracle itself sits 2e-4 from the reference); the mean error is checked tighter.
LOG_TOL = 2e-3
LOG_MEAN_TOL = 2e-5


def test_fbank_matches_reference_kaldifeat_outputs():
    from libs.amd import frontend
    for name, wave, kw, ref in helpers.fbank_cases():
        got = frontend.fbank([wave], **helpers.fbank_torchaudio_kw(kw))[0].cpu().numpy()
        assert got.shape == ref.shape, name
        if kw.get("use_log_fbank", True):
            assert np.abs(got - ref).max() < LOG_TOL and np.abs(got - ref).mean() < LOG_MEAN_TOL, (name, np.abs(got - ref).max())
        else:
            assert helpers.rel_err(got, ref) < 2e-5, name


def test_mfcc_matches_reference_kaldifeat_outputs():
    from libs.amd import frontend
    for name, wave, kw, ref in helpers.mfcc_cases():
        full = dict(helpers.MFCC_REF_DEFAULTS, **kw)
        got = frontend.mfcc([wave], **{helpers.FBANK_KW.get(k, k): v for k, v in full.items()})[0].cpu().numpy()
        assert got.shape == ref.shape, name
        assert np.abs(got - ref).max() < LOG_TOL and np.abs(got - ref).mean() < 5e-5, (name, np.abs(got - ref).max())
    # the reference's class interface with its default feature type
    import torch
    from libs.egs.kaldi_features import KaldiFeature
    name, wave, kw, ref = helpers.mfcc_cases()[0]
    feats = KaldiFeature("mfcc", {"use_energy": True, "energy_floor": 0.0, "dither": 0.0})(torch.from_numpy(wave)[None, :])
    assert np.abs(feats[0].cpu().numpy() - ref).max() < LOG_TOL


def test_ragged_batch_equals_per_utterance_and_the_oracle():
    from libs.amd import frontend, synth
    lens = [16000, 400, 399, 48000, 1234, 31999, 0, 560]
    waves = [synth.synth_wave(n, 100 + i) for i, n in enumerate(lens)]
    feats, off = frontend.fbank_packed(waves, num_mel_bins=80, energy_floor=0.0)
    assert list(np.diff(off)) == [fbank_oracle.num_frames(n) for n in lens] and off[-1] == feats.shape[0]
    full = feats.cpu().numpy()
    for i in (0, 1, 3, 4, 7):
        want = fbank_oracle.fbank(waves[i], num_bins=80)
        assert np.abs(full[off[i]:off[i + 1]] - want).max() < LOG_TOL
        alone = frontend.fbank([waves[i]], num_mel_bins=80, energy_floor=0.0)[0].cpu().numpy()
        assert np.array_equal(alone, full[off[i]:off[i + 1]])            # a frame's arithmetic does not depend on its batch


def test_mean_variance_normalisation_matches_torch():
    import torch
    from libs.amd import frontend, synth
    from libs.egs.kaldi_features import KaldiFeature, InputSequenceNormalization
    waves = [synth.synth_wave(n, 300 + i) for i, n in enumerate([8000, 20000, 4000])]
    raw = frontend.fbank(waves, num_mel_bins=40)
    for std_norm in (False, True):
        got = frontend.fbank(waves, num_mel_bins=40, mean_norm=True, std_norm=std_norm)
        for r, g in zip(raw, got):
            r64 = r.double().cpu()
            want = r64 - r64.mean(0)
            if std_norm:
                want = want / torch.clamp(r64.std(0), min=1e-10)
            assert (g.cpu().double() - want).abs().max() < 2e-5
    # the reference's class interface: [batch, time] tensor + relative lengths
    T = 20000
    batch = torch.zeros(3, T)
    for i, w in enumerate(waves):
        batch[i, :len(w)] = torch.from_numpy(w)
    rel = torch.tensor([len(w) / T for w in waves])
    feats = KaldiFeature("fbank", {"num_mel_bins": 40, "dither": 0.0}, {"mean_norm": True, "std_norm": False})(batch, rel)
    want = frontend.fbank(waves, num_mel_bins=40, mean_norm=True)
    assert all(torch.equal(a, b) for a, b in zip(feats, want))
    one = InputSequenceNormalization(mean_norm=True, std_norm=True)(raw[1])
    assert (one - frontend.fbank(waves, num_mel_bins=40, mean_norm=True, std_norm=True)[1]).abs().max() < 1e-6


def test_other_sample_rates_and_window_sizes():
    from libs.amd import frontend, synth
    for sr, kw in ((8000.0, dict(num_bins=30)), (48000.0, dict(num_bins=40, frame_length_ms=25.0)), (16000.0, dict(num_bins=64, frame_length_ms=32.0, round_to_power_of_two=False))):
        wave = synth.synth_wave(int(sr * 0.7), 77, sr)
        want = fbank_oracle.fbank(wave, sample_rate=sr, **kw)
        got = frontend.fbank([wave], **helpers.fbank_torchaudio_kw(dict(kw, sample_rate=sr)))[0].cpu().numpy()
        assert got.shape == want.shape and np.abs(got - want).max() < LOG_TOL, (sr, np.abs(got - want).max())
    with pytest.raises(Exception):
        frontend.fbank([synth.synth_wave(16000, 1)], frame_length=25.0, round_to_power_of_two=False)     # 400 is not a power of two


def test_waveforms_to_embeddings_stay_on_the_device():
    """wav -> fbank(80) + CMN -> x-vector: features are handed to the extractor as device matrices."""
    from libs.amd import frontend, synth
    g, sd, model = helpers.golden_model("ecapa_c3")
    model.cuda()
    model.amd_precision = "f32"
    waves = [synth.synth_wave(n, 500 + i) for i, n in enumerate([32000, 48000, 20000])]
    feats = frontend.fbank(waves, num_mel_bins=80, mean_norm=True)
    emb = model.extract_embedding_batch(feats).numpy()
    host = model.extract_embedding_batch([f.cpu().numpy() for f in feats]).numpy()
    assert np.isfinite(emb).all() and np.array_equal(emb, host)
