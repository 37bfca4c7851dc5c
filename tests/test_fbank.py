"""Front-end, CPU side: the numpy restatement of the Kaldi fbank computation is pinned to outputs
of the reference's own kaldifeat code (tests/golden/fbank.npz, made by oracle/gen_fbank_golden.py from
oracle/_ref/libkaldifeat_ref.so), and the host half of the C ABI (frame counts, option checking) agrees with it."""

import numpy as np
import pytest

import helpers
from oracle import fbank_oracle


def test_oracle_matches_reference_kaldifeat_outputs():
    cases = helpers.fbank_cases()
    assert len(cases) >= 9
    for name, wave, kw, ref in cases:
        got = fbank_oracle.fbank(wave, **kw)
        assert got.shape == ref.shape, name
        if kw.get("use_log_fbank", True):
            assert np.abs(got - ref).max() < 5e-4, (name, np.abs(got - ref).max())        # log-mel values are ~10..25
        else:
            assert helpers.rel_err(got, ref) < 1e-5, name


def test_mfcc_oracle_matches_reference_kaldifeat_outputs():
    cases = helpers.mfcc_cases()
    assert len(cases) >= 5
    for name, wave, kw, ref in cases:
        got = fbank_oracle.mfcc(wave, **kw)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 5e-4, (name, np.abs(got - ref).max())


def test_frame_counts_of_the_c_abi_match_the_oracle():
    from libs.amd import frontend
    for n in (0, 1, 399, 400, 401, 559, 560, 561, 16000, 123457):
        for snip in (True, False):
            assert frontend.num_frames(n, snip_edges=snip) == fbank_oracle.num_frames(n, snip_edges=snip), (n, snip)
    assert frontend.num_frames(8000, sample_frequency=8000.0, frame_length=20.0, frame_shift=7.5) == fbank_oracle.num_frames(8000, 8000.0, 20.0, 7.5)


def test_unsupported_options_are_refused_up_front():
    from libs.amd import frontend
    from libs.egs.kaldi_features import KaldiFeature
    with pytest.raises(ValueError):
        frontend.fbank_options(dither=1.0)
    with pytest.raises(ValueError):
        frontend.fbank_options(window_type="blackman")
    with pytest.raises(TypeError):
        frontend.fbank_options(num_bins=80)                      # torchaudio spells it num_mel_bins
    with pytest.raises(ValueError):
        KaldiFeature("mfcc", {"num_ceps": 30})                   # > num_mel_bins (23)
    with pytest.raises(TypeError):
        KaldiFeature("mfcc", {"use_power": False})               # torchaudio's mfcc has no such option
    with pytest.raises(ValueError):
        KaldiFeature("fbank", {"dither": 0.5})
