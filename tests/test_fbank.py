"""Front-end, CPU side: the numpy restatement of the Kaldi fbank computation is pinned to outputs
of the reference's own kaldifeat code (tests/golden/fbank.npz, made by oracle/gen_fbank_golden.py from
oracle/_ref/libkaldifeat_ref.so), and the host half of the C ABI (frame counts, option checking) agrees with it."""

import glob
import os

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
        frontend.fbank_options(window_type="kaiser")
    opts, _ = frontend.fbank_options(window_type="blackman", blackman_coeff=0.4, vtln_warp=1.1)     # offered since round 2
    assert opts.window_type == 5 and abs(opts.blackman_coeff - 0.4) < 1e-7 and abs(opts.vtln_warp - 1.1) < 1e-7
    with pytest.raises(TypeError):
        frontend.fbank_options(num_bins=80)                      # torchaudio spells it num_mel_bins
    with pytest.raises(ValueError):
        KaldiFeature("mfcc", {"num_ceps": 30})                   # > num_mel_bins (23)
    with pytest.raises(TypeError):
        KaldiFeature("mfcc", {"use_power": False})               # torchaudio's mfcc has no such option
    with pytest.raises(ValueError):
        KaldiFeature("fbank", {"dither": 0.5})


def test_sliding_cmn_and_vad_restatements_known_answers():
    """Kaldi / runtime algorithms with no reference-side fixture (parity unpinned): hand-checkable cases."""
    rng = np.random.RandomState(3)
    x = rng.randn(50, 4).astype(np.float32) + 2.0
    # a window covering the whole utterance is plain mean subtraction, centred or not
    for center in (True, False):
        got = fbank_oracle.sliding_cmn(x, cmn_window=600, min_window=100, center=center)
        assert np.abs(got - (x - x.mean(0))).max() < 1e-5
    # centred window of 4 on a ramp: interior frames see [t-2, t+2) -> mean t-0.5; the edges see the shifted window
    ramp = np.arange(10, dtype=np.float32)[:, None]
    got = fbank_oracle.sliding_cmn(ramp, cmn_window=4, center=True)[:, 0]
    assert np.allclose(got, [-1.5, -0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 1.5])
    # causal window of 3 with min_window 2: frame 0 uses frames [0,2), then [0,2), [0,3), [1,4) ...
    got = fbank_oracle.sliding_cmn(ramp, cmn_window=3, min_window=2, center=False)[:, 0]
    assert np.allclose(got[:5], [-0.5, 0.5, 1.0, 1.5, 1.5])
    # single-frame windows give zeros under variance normalisation
    assert np.all(fbank_oracle.sliding_cmn(ramp[:1], cmn_window=4, center=True, norm_vars=True) == 0)
    # VAD: threshold 5 + 0.5 * mean; context 1, proportion 0.6 -> a frame needs 2 of its 3 neighbours (edges: 2 of 2)
    e = np.array([10, 10, 0, 10, 0, 0, 0, 10, 10, 10], dtype=np.float32)[:, None]
    v = fbank_oracle.vad_energy(e, vad_energy_threshold=5.0, vad_energy_mean_scale=0.0, vad_frames_context=1, vad_proportion_threshold=0.6)
    assert list(v) == [1, 1, 1, 0, 0, 0, 0, 1, 1, 1]
    v = fbank_oracle.vad_energy(e, vad_energy_threshold=1.0, vad_energy_mean_scale=0.5, vad_frames_context=0, vad_proportion_threshold=0.5)
    assert list(v) == [1, 1, 0, 1, 0, 0, 0, 1, 1, 1]                        # threshold 1 + 0.5 * 6 = 4


REF_WAVS = "/root/reference/runtime/test/wav"


@pytest.mark.skipif(not (os.path.isdir(REF_WAVS) and os.path.exists(os.path.join(helpers.REPO, "oracle", "_ref", "libkaldifeat_ref.so"))),
                    reason="build container only: needs the reference's test wavs and oracle/_ref (make -C oracle -f Makefile.ref)")
def test_oracle_matches_the_compiled_reference_on_the_reference_test_wavs():
    """The reference's own 16 kHz recordings (runtime/test/wav), fbank-80 as its runtime config computes it, straight through
    the compiled kaldifeat code vs the numpy restatement - real speech instead of synthetic harmonics."""
    import ctypes as C
    import wave
    from oracle import gen_fbank_golden as G
    lib = C.CDLL(os.path.join(helpers.REPO, "oracle", "_ref", "libkaldifeat_ref.so"))
    lib.kaldifeat_ref_fbank.restype = C.c_int
    paths = sorted(glob.glob(os.path.join(REF_WAVS, "*.wav")))[:4]
    assert paths
    for path in paths:
        with wave.open(path, "rb") as f:
            assert f.getsampwidth() == 2
            sr = f.getframerate()
            x = np.frombuffer(f.readframes(min(f.getnframes(), 5 * sr)), dtype="<i2").reshape(-1, f.getnchannels())[:, 0].astype(np.float32)
        kw = dict(num_bins=80, sample_rate=float(sr))
        ref = G.reference_fbank(lib, x, **kw)
        got = fbank_oracle.fbank(x, **kw)
        assert got.shape == ref.shape and len(ref) > 100
        assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 2e-5, (path, np.abs(got - ref).max())
        lib.kaldifeat_ref_mfcc.restype = C.c_int
        mkw = dict(num_ceps=20, num_bins=30, sample_rate=float(sr))
        ref = G.reference_mfcc(lib, x, **mkw)
        got = fbank_oracle.mfcc(x, **mkw)
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).mean() < 5e-5, path


def test_vad_restatement_matches_the_compiled_reference():
    """tests/golden/vad.npz holds the decisions of the reference's own ComputeVadEnergy (runtime/extractor/torch_asv_extractor.cc:14-62,
    compiled in place by oracle/Makefile.ref; oracle/gen_vad_golden.py): thresholds with ties, contexts wider than the
    utterance, one- and two-frame inputs, nothing voiced."""
    import json
    g = np.load(os.path.join(helpers.REPO, "tests", "golden", "vad.npz"))
    meta = json.loads(str(g["meta"]))
    assert len(meta) >= 10
    for case in meta:
        e = g[case["name"] + "/energy"]
        feats = np.stack([e, np.zeros_like(e)], axis=1)
        got = fbank_oracle.vad_energy(feats, **case["options"])
        assert np.array_equal(got, g[case["name"] + "/voiced"]), case["name"]


def test_vad_reference_library_reproduces_the_fixture():
    """Build container only: the compiled reference itself still yields the committed fixture."""
    lib_path = os.path.join(helpers.REPO, "oracle", "_ref", "libextractor_ref.so")
    if not os.path.exists(lib_path) or not os.path.isdir("/root/reference"):
        pytest.skip("oracle/_ref/libextractor_ref.so is built in the build container only (make -C oracle -f Makefile.ref)")
    import ctypes as C
    import json
    lib = C.CDLL(lib_path)
    g = np.load(os.path.join(helpers.REPO, "tests", "golden", "vad.npz"))
    for case in json.loads(str(g["meta"])):
        e = np.ascontiguousarray(g[case["name"] + "/energy"].reshape(-1, 1))
        o = case["options"]
        voiced = np.zeros(len(e), dtype=np.float32)
        rc = lib.extractor_ref_vad_energy(e.ctypes.data_as(C.c_void_p), len(e), 1, C.c_float(o["vad_energy_threshold"]), C.c_float(o["vad_energy_mean_scale"]),
                                          int(o["vad_frames_context"]), C.c_float(o["vad_proportion_threshold"]), voiced.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert np.array_equal(voiced.astype(np.uint8), g[case["name"] + "/voiced"]), case["name"]
