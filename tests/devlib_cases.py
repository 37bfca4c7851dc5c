"""Cases that need the DEVELOPER build of the library (`make -C asv-subtools_amd/csrc dev` -> libasv_amd_dev.so): kernel variants
that are not part of the product - the chain kernel's first pooling epilogue (ASV_AMD_CHAIN_POOLV=0) and the four-wave chain
kernel (csrc/tools/kernels_tdnn_chain4.hip, ASV_AMD_CHAIN_WAVES=4).  Not collected by default (the file name has no test_
prefix): tests/test_gpu_devlib.py runs this file in a subprocess with ASV_AMD_LIB pointing at the developer library and
ASV_AMD_LIVE_TUNE=1 (the switches are then read at every launch), and skips when that library has not been built."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _gpu_model(name, precision):
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = precision
    return g, sd, model


def test_developer_library_is_the_one_running():
    from libs.amd import capi
    capi.lib()
    with open("/proc/self/maps") as f:
        assert "libasv_amd_dev.so" in f.read()


def test_both_pooling_epilogues_of_the_chain_kernel(monkeypatch):
    from libs.amd import synth
    g2, sd2, model2 = _gpu_model("xvector_near_ragged", "bf16")
    mats2 = [synth.synth_feats(200, 80, 7000 + i) for i in range(300)]
    monkeypatch.setenv("ASV_AMD_NO_CHAIN", "1")
    b = model2.extract_embedding_batch(mats2).numpy()
    lens = [1, 2, 3, 5, 4, 7, 1, 9, 13, 21, 2, 34, 6, 55, 3, 89, 11, 144, 1, 1, 8, 233, 17, 2, 40, 31, 32, 33, 64, 63, 65, 12] * 6
    mats3 = [synth.synth_feats(T, 80, 9000 + i) for i, T in enumerate(lens)]
    ref = model2.extract_embedding_batch(mats3).numpy()                  # per-layer kernels
    monkeypatch.delenv("ASV_AMD_NO_CHAIN")
    outs = {}
    for v in ("0", "1"):
        monkeypatch.setenv("ASV_AMD_CHAIN_POOLV", v)
        outs[v] = model2.extract_embedding_batch(mats3).numpy()
        assert np.isfinite(outs[v]).all()
        assert rel_err(outs[v], ref) < 6e-3, (v, rel_err(outs[v], ref))
        a200 = model2.extract_embedding_batch(mats2).numpy()
        assert rel_err(a200, b) < 6e-3, (v, rel_err(a200, b))
    assert rel_err(outs["0"], outs["1"]) < 1e-4, rel_err(outs["0"], outs["1"])   # same moments, other summation order


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_four_wave_chain_kernel_matches_eight_wave(monkeypatch, precision):
    """ASV_AMD_CHAIN_WAVES=4 runs the chain as four waves of 512 registers with the pooling arithmetic inside the next unit's K
    loop (kernels_tdnn_chain4.hip: measured, not the default).  Same operands and the same rounding of the intermediate tiles as
    the 8-wave kernel; only the order of the f32 sums of the pooled moments differs.  Ragged lengths >= 32 frames (seams, gap
    rows and whole fragments inside 128-row tiles) at the C2 shape; a batch with a shorter utterance falls back to 8 waves."""
    from libs.amd import synth
    g, sd, model = _gpu_model("xvector_near_ragged", precision)
    rng = np.random.RandomState(3)
    lens = [200] * 40 + [int(x) for x in rng.randint(32, 420, size=90)] + [32, 33, 63, 64, 65, 127, 128, 129, 191, 193]
    mats = [synth.synth_feats(t, 80, 7000 + i) for i, t in enumerate(lens)]
    monkeypatch.setenv("ASV_AMD_CHAIN_WAVES", "8")
    eight = model.extract_embedding_batch(mats).numpy()
    monkeypatch.setenv("ASV_AMD_CHAIN_WAVES", "4")
    four = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(four).all()
    assert not np.array_equal(four, eight)                       # another kernel did run
    assert rel_err(four, eight) < 2e-5, rel_err(four, eight)
    short = mats + [synth.synth_feats(20, 80, 1)]                # 20 frames: three utterances could meet in one fragment -> 8 waves
    with_four = model.extract_embedding_batch(short).numpy()
    monkeypatch.setenv("ASV_AMD_CHAIN_WAVES", "8")
    assert np.array_equal(model.extract_embedding_batch(short).numpy(), with_four)
