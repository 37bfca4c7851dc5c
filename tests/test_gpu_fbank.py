"""asv_fbank / asv_cmvn on the MI355X against the reference's kaldifeat outputs (tests/golden/fbank.npz) and the numpy
oracle; batching, ragged lengths, mean/variance normalisation, and waveform -> embedding without a host round trip."""

import os

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


def test_pcm16_input_is_bit_identical_to_float_input():
    import torch
    from libs.amd import frontend, synth
    waves = [synth.synth_wave(n, 600 + i) for i, n in enumerate([16000, 4001, 9999])]
    as_f32 = frontend.fbank(waves, num_mel_bins=80, snip_edges=False)
    as_i16 = frontend.fbank([w.astype(np.int16) for w in waves], num_mel_bins=80, snip_edges=False)
    assert all(torch.equal(a, b) for a, b in zip(as_f32, as_i16))
    m32 = frontend.mfcc(waves, sample_frequency=8000.0)
    m16 = frontend.mfcc([torch.from_numpy(w.astype(np.int16)).cuda() for w in waves], sample_frequency=8000.0)
    assert all(torch.equal(a, b) for a, b in zip(m32, m16))


def test_sliding_cmn_matches_the_kaldi_restatement():
    import torch
    from libs.amd import synth
    from libs.amd import frontend
    lens = [650, 299, 300, 301, 1, 37, 0, 1200]
    mats = [synth.synth_feats(T, 30, 40 + i) * 3.0 + 1.5 for i, T in enumerate(lens)]
    off = np.concatenate([[0], np.cumsum(lens)])
    packed = torch.from_numpy(np.concatenate(mats, axis=0)).cuda()
    for kw in (dict(cmn_window=300, center=True), dict(), dict(cmn_window=100, min_window=20, norm_vars=True), dict(cmn_window=64, center=True, norm_vars=True)):
        got = frontend.cmvn_sliding(packed, off, **kw).cpu().numpy()
        for i, m in enumerate(mats):
            if len(m):
                want = fbank_oracle.sliding_cmn(m, **kw)
                assert np.abs(got[off[i]:off[i + 1]] - want).max() < 2e-5, (kw, lens[i])


def test_energy_vad_and_voiced_frame_selection():
    import torch
    from libs.amd import frontend, synth
    lens = [32000, 8000, 16000, 400, 24000]
    waves = []
    for i, n in enumerate(lens):
        w = synth.synth_wave(n, 700 + i)
        gate = (np.sin(np.arange(n) / 16000.0 * 2 * np.pi * 1.3 + i) > -0.2).astype(np.float32)       # silences of a few hundred ms
        waves.append(np.rint(w * gate + 3.0 * np.random.RandomState(i).standard_normal(n)).astype(np.float32))
    feats, off = frontend.fbank_packed(waves, kind="mfcc", use_energy=True, energy_floor=0.0, num_ceps=20, num_mel_bins=30)
    host = feats.cpu().numpy()
    for kw in (dict(), dict(vad_energy_threshold=5.5, vad_frames_context=0, vad_proportion_threshold=0.6), dict(vad_energy_mean_scale=0.0, vad_energy_threshold=9.0)):
        voiced, counts = frontend.vad_energy(feats, off, **kw)
        v = voiced.cpu().numpy()
        for i in range(len(lens)):
            want = fbank_oracle.vad_energy(host[off[i]:off[i + 1]], **kw)
            assert np.array_equal(v[off[i]:off[i + 1]], want), (kw, i)
            assert counts[i] == want.sum()
        assert 0 < counts.sum() < len(v)                                     # the gate really produced both classes
        kept, koff = frontend.select_voiced(feats, voiced, off, counts)
        assert np.array_equal(kept.cpu().numpy(), host[v.astype(bool)]) and list(np.diff(koff)) == list(counts)


def test_energy_vad_matches_the_compiled_reference():
    """The device VAD against the decisions of the reference's own ComputeVadEnergy (tests/golden/vad.npz,
    runtime/extractor/torch_asv_extractor.cc:14-62 compiled in place in the build container): all cases of one option set in
    one packed launch."""
    import json
    import torch
    from libs.amd import frontend
    g = np.load(os.path.join(helpers.REPO, "tests", "golden", "vad.npz"))
    meta = json.loads(str(g["meta"]))
    by_opts = {}
    for case in meta:
        by_opts.setdefault(json.dumps(case["options"], sort_keys=True), []).append(case)
    n_checked = 0
    for key, cases in by_opts.items():
        cols = [g[c["name"] + "/energy"] for c in cases]
        off = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int64)
        feats = np.zeros((int(off[-1]), 4), dtype=np.float32)
        feats[:, 0] = np.concatenate(cols)
        feats[:, 1:] = np.random.RandomState(0).standard_normal((len(feats), 3))
        voiced, counts = frontend.vad_energy(torch.from_numpy(feats).cuda(), off, **json.loads(key))
        v = voiced.cpu().numpy()
        for i, c in enumerate(cases):
            want = g[c["name"] + "/voiced"]
            assert np.array_equal(v[off[i]:off[i + 1]], want), c["name"]
            assert counts[i] == want.sum()
            n_checked += 1
    assert n_checked == len(meta)


def test_full_size_properties_shift_and_batch_invariance():
    """BASELINE-sized batch (512 utterances x 200 frames, too slow for the numpy oracle): properties the domain offers.
    Dropping the first frame shift of samples drops exactly the first frame (snip_edges on), whatever the batch around it;
    an utterance's frames do not depend on its neighbours or on its sample alignment in the packed buffer."""
    import torch
    from libs.amd import frontend, synth
    n = 400 + 199 * 160
    base = [synth.synth_wave(n + 160 + (i % 3), 4000 + i) for i in range(16)]            # odd lengths: odd packed offsets
    waves = [base[i % 16] for i in range(512)]
    feats, off = frontend.fbank_packed(waves, num_mel_bins=80)
    assert feats.shape == (int(off[-1]), 80) and off[-1] == 512 * 201 and torch.isfinite(feats).all()
    shifted, off2 = frontend.fbank_packed([w[160:] for w in waves], num_mel_bins=80)
    assert list(np.diff(off2)) == [200] * 512
    a = feats.view(512, 201, 80)[:, 1:, :]
    assert torch.equal(a, shifted.view(512, 200, 80))
    for i in (0, 17, 511):                                                                # copies of one utterance at different places
        assert torch.equal(feats[off[i]:off[i + 1]], feats[off[i % 16]:off[i % 16 + 1]])
    want = fbank_oracle.fbank(base[5], num_bins=80)
    assert np.abs(feats[off[5]:off[6]].cpu().numpy() - want).max() < LOG_TOL
