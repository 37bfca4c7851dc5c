# -*- coding:utf-8 -*-
"""Deterministic synthetic weights and feature matrices.

There are no trained checkpoints and no datasets on the build or GPU machines, so every
parity test, golden fixture and bench run uses the recipe below.
It depends on numpy's legacy MT19937 `RandomState` only, so the same seed gives the same
bytes on every machine - that is what lets the 24 MB x-vector state_dict stay out of the
repo while the golden *embeddings* (a few hundred KB) are committed.

The recipe is keyed on the state_dict key names of the reference blueprints
(/root/reference/pytorch/model/xvector.py:28-35, ecapa_tdnn_xvector.py:263-339,
libs/nnet/resnet.py:212-371) so the reference model and this package load identical
parameters.
"""

import zlib

import numpy as np


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode("utf-8")) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(key, shape, seed=0):
    """One state_dict entry. Batch-norm statistics are deliberately non-trivial
    (default running_mean=0 / running_var=1 would hide BN bugs)."""
    shape = tuple(int(s) for s in shape)
    r = _rng(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "t":                                    # attention temperatures (pooling.py:242-252): positive, not tiny
        return r.uniform(1.0, 5.0, shape).astype(np.float32)
    if leaf in ("prior_mean", "prior_logprec"):        # xi-vector prior (pooling.py:178-179): zeros at init, make them matter
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "s" and len(shape) == 1:               # LDE scale (pooling.py:143): small, so that the soft assignment stays soft
        return r.uniform(0.02, 0.06, shape).astype(np.float32)
    if leaf == "running_mean":
        return (0.5 * r.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return r.uniform(0.5, 2.0, shape).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:           # BN gamma
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "bias":
        return (0.2 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) >= 2:                                # conv / linear weight
        fan_in = shape[1]
        for k in shape[2:]:
            fan_in *= min(int(k), 3)                   # at most 3 active taps per axis
        std = (2.0 / max(fan_in, 1)) ** 0.5
        return (std * r.standard_normal(shape)).astype(np.float32)
    return r.standard_normal(shape).astype(np.float32)


def synth_state_dict(shapes, seed=0):
    """shapes: {key: shape}. Returns {key: np.ndarray} in the same order."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_feats(num_frames, feat_dim, seed):
    """A CMN-like Kaldi feature matrix [T, D] float32."""
    return np.random.RandomState(int(seed)).randn(int(num_frames), int(feat_dim)).astype(np.float32)


def synth_wave(num_samples, seed, sample_rate=16000.0):
    """A speech-like waveform: a few drifting harmonics under a slow envelope plus noise, rounded to int16 values and
    returned as float32 (Kaldi WaveData convention).  Includes a DC offset so remove_dc_offset matters."""
    r = np.random.RandomState(int(seed))
    t = np.arange(int(num_samples), dtype=np.float64) / sample_rate
    f0 = r.uniform(90.0, 220.0)
    x = np.zeros_like(t)
    for h in range(1, 12):
        x += r.uniform(0.2, 1.0) / h * np.sin(2 * np.pi * h * f0 * t * (1.0 + 0.02 * np.sin(2 * np.pi * 3.0 * t)) + r.uniform(0, 6.28))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * r.uniform(2.0, 5.0) * t + r.uniform(0, 6.28))
    x = 3000.0 * env * x + 400.0 * r.standard_normal(t.shape) + r.uniform(-50.0, 50.0)
    return np.clip(np.rint(x), -32768, 32767).astype(np.float32)


def synth_lengths(n, lo, hi, seed):
    """Utterance lengths ~ randint(lo, hi] (config C4/C5 stand-ins)."""
    return np.random.RandomState(int(seed)).randint(int(lo), int(hi) + 1, size=int(n)).astype(np.int64)


def synth_speaker_embeddings(n_spk, per_spk, dim, seed, within=1.0, between=1.0):
    """Planted-speaker embedding set for scoring tests: x = between*m_spk + within*e."""
    r = np.random.RandomState(int(seed))
    means = between * r.standard_normal((n_spk, dim))
    x = means[:, None, :] + within * r.standard_normal((n_spk, per_spk, dim))
    labels = np.repeat(np.arange(n_spk), per_spk)
    return x.reshape(n_spk * per_spk, dim).astype(np.float32), labels


def synth_trials(labels, n_trials, seed, target_frac=0.5):
    """Random (enroll_idx, test_idx, is_target) trial list over a labelled set."""
    r = np.random.RandomState(int(seed))
    labels = np.asarray(labels)
    n = len(labels)
    by_spk = {}
    for i, l in enumerate(labels):
        by_spk.setdefault(int(l), []).append(i)
    enroll = np.empty(n_trials, dtype=np.int64)
    test = np.empty(n_trials, dtype=np.int64)
    tgt = np.empty(n_trials, dtype=np.int64)
    for t in range(n_trials):
        a = r.randint(n)
        if r.rand() < target_frac:
            peers = by_spk[int(labels[a])]
            b = peers[r.randint(len(peers))]
            if b == a:
                b = peers[(peers.index(a) + 1) % len(peers)]
        else:
            b = r.randint(n)
            while labels[b] == labels[a]:
                b = r.randint(n)
        enroll[t], test[t], tgt[t] = a, b, int(labels[a] == labels[b])
    return enroll, test, tgt


def synth_planted_utts(n_spk, per_spk, dim, t_lo, t_hi, noise, seed=3):
    """Planted-speaker feature matrices for the EER gates (tests/test_gpu_eer_gate.py, bench.py's `parity.eer` leg): speaker s owns a
    fixed random feature track, an utterance is its first T ~ randint[t_lo, t_hi] frames plus white noise.  Returns
    (list of [T, dim] float32 matrices, int labels)."""
    r = np.random.RandomState(int(seed))
    mats, labels = [], []
    for s in range(int(n_spk)):
        base = synth_feats(t_hi, dim, 500_000 + s)
        for _ in range(int(per_spk)):
            T = int(r.randint(t_lo, t_hi + 1))
            mats.append((base[:T] + noise * r.standard_normal((T, dim)).astype(np.float32)).astype(np.float32))
            labels.append(s)
    return mats, np.asarray(labels)
