"""Shared test helpers: golden fixtures, synthetic weights, blueprint construction."""

import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
MODEL_DIR = os.path.join(REPO, "asv-subtools_amd", "pytorch", "model")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    shapes = {}
    for k, v in zip(g["shape_keys"], g["shape_vals"]):
        k, v = str(k), str(v)
        shapes[k] = tuple(int(d) for d in v.split(",")) if v else ()
    return g, shapes


def golden_state_dict(name):
    from libs.amd import synth
    g, shapes = load_golden(name)
    return g, synth.synth_state_dict(shapes, int(g["wseed"]))


def golden_feats(g):
    from libs.amd import synth
    return [synth.synth_feats(int(T), int(g["dim"]), int(seed)) for T, seed in g["utts"]]


def build_model(blueprint, creation, sd=None):
    """Blueprint from THIS repo's model/ directory + optional numpy state_dict."""
    import torch
    import libs.support.utils as utils
    model = utils.create_model_from_py(os.path.join(MODEL_DIR, blueprint), creation)
    if sd is not None:
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    return model


def golden_model(name):
    g, sd = golden_state_dict(name)
    return g, sd, build_model(str(g["blueprint"]), str(g["creation"]), sd)


def rel_err(a, b):
    """max |a-b| / max |b|  (the 'relative fp32' metric of BASELINE.json's north_star)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def fbank_cases():
    """tests/golden/fbank.npz: outputs of the reference's kaldifeat code (oracle/gen_fbank_golden.py) ->
    [(name, waveform regenerated from its seed, option dict, reference features)]."""
    import json
    from libs.amd import synth
    g = np.load(os.path.join(GOLDEN, "fbank.npz"))
    return [(name, synth.synth_wave(samples, seed, kw.get("sample_rate", 16000.0)), kw, g[name]) for name, samples, seed, kw in json.loads(str(g["cases"]))]


def mfcc_cases():
    import json
    from libs.amd import synth
    g = np.load(os.path.join(GOLDEN, "fbank.npz"))
    return [(name, synth.synth_wave(samples, seed, kw.get("sample_rate", 16000.0)), kw, g[name]) for name, samples, seed, kw in json.loads(str(g["mfcc_cases"]))]


# asv_fbank_opts_t / oracle names -> torchaudio.compliance.kaldi.fbank keywords (what libs.amd.frontend takes)
FBANK_KW = dict(sample_rate="sample_frequency", frame_length_ms="frame_length", frame_shift_ms="frame_shift", preemph="preemphasis_coefficient",
                num_bins="num_mel_bins")
MFCC_REF_DEFAULTS = dict(use_energy=True, energy_floor=0.0)           # kaldifeat's MfccOptions (torchaudio: False, 1.0)


def fbank_torchaudio_kw(kw, energy_floor_default=0.0):
    out = {FBANK_KW.get(k, k): v for k, v in kw.items()}
    out.setdefault("energy_floor", energy_floor_default)          # kaldifeat's default (torchaudio's is 1.0)
    return out


def plda_ragged_set(g, synth=None):
    """Training set of tests/golden/plda_ragged.npz: planted speaker embeddings, a different number of examples per speaker."""
    if synth is None:
        from libs.amd import synth
    x, labels = synth.synth_speaker_embeddings(int(g["n_spk"]), int(g["per_spk_max"]), int(g["dim"]), seed=int(g["seed"]), within=1.0, between=0.7)
    keep = np.zeros(len(labels), dtype=bool)
    sizes = np.random.RandomState(int(g["seed"]) + 1).randint(2, int(g["per_spk_max"]) + 1, size=int(g["n_spk"]))
    for spk, n in enumerate(sizes):
        keep[np.flatnonzero(labels == spk)[:n]] = True
    return x[keep], labels[keep]
