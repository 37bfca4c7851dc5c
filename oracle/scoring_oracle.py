"""numpy restatement of the scoring back-end (TEST INFRASTRUCTURE - see oracle/__init__.py).

Cosine path = the Kaldi binaries the reference shells out to; Kaldi itself is not vendored
under /root/reference (unpinned `git clone` of kaldi-asr/kaldi master, README.md:195-197), so
the arithmetic follows Kaldi's published definitions, anchored on the reference's call sites:
  ivector-mean / ivector-subtract-global-mean    score/process.sh:156-192
  ivector-normalize-length --scaleup=false       score/process.sh:194-203   (x / ||x||_2)
  ivector-compute-dot-products                   score/score.sh:82-97
PLDA = the reference's own Python restatement of Kaldi's ivector/plda.cc,
score/pyplda/plda_base.py (stats 37-81, EM 227-300, diagonalisation 302-335, transform 93-107,
normalisation 165-172, LLR 109-136); two-covariance scorer score/pyplda/gaussian-plda-scoring.py:23-50.
EER = computeEER-like-Bosaris.py:50-91.  PLDA / EER are pinned to the reference code itself by
tests/golden/scoring_plda.npz (oracle/gen_golden.py); the Kaldi-binary cosine path has no
reference-side fixture and is "parity unpinned" beyond its definition.
"""

import math

import numpy as np

M_LOG_2PI = 1.8378770664093454835606594728112


# ------------------------------------------------------------------------------- cosine

def global_mean(x):
    return np.asarray(x, dtype=np.float64).mean(axis=0)


def length_normalize(x, mean=None):
    """subtract-global-mean (optional) then x / ||x||  (--scaleup=false)."""
    x = np.asarray(x, dtype=np.float64)
    if mean is not None:
        x = x - mean
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def dot_trials(enroll, test, ei, ti):
    return np.einsum("ij,ij->i", np.asarray(enroll, dtype=np.float64)[ei], np.asarray(test, dtype=np.float64)[ti])


# --------------------------------------------------------------------------------- PLDA

class PldaStats(object):
    """plda_base.py:37-81."""

    def __init__(self, dim):
        self.dim = dim
        self.num_example = 0
        self.num_classes = 0
        self.class_weight = 0.0
        self.example_weight = 0.0
        self.sum = np.zeros((dim, 1))
        self.offset_scatter = np.zeros((dim, dim))
        self.classinfo = []                  # (weight, n, mean[dim,1])

    def add_samples(self, weight, group):
        n = group.shape[0]
        mean = np.mean(group, axis=0).reshape(-1, 1)
        self.offset_scatter += weight * group.T.dot(group)
        self.offset_scatter += -n * weight * mean.dot(mean.T)
        self.classinfo.append((weight, n, mean))
        self.num_example += n
        self.num_classes += 1
        self.class_weight += weight
        self.example_weight += weight * n
        self.sum += weight * mean


def plda_em(stats, num_iters=10):
    """plda_base.py:248-300: returns (mean[dim,1], within_var, between_var)."""
    dim = stats.dim
    between, within = np.eye(dim), np.eye(dim)
    for _ in range(num_iters):
        w_stats = np.zeros((dim, dim)); b_stats = np.zeros((dim, dim))
        w_count = 0.0; b_count = 0.0
        w_stats += stats.offset_scatter
        w_count += stats.example_weight - stats.class_weight
        w_inv, b_inv = np.linalg.inv(within), np.linalg.inv(between)
        gmean = stats.sum / stats.class_weight
        for weight, n, cmean in stats.classinfo:
            if not n:
                continue
            mix = np.linalg.inv(b_inv + n * w_inv)
            m = (cmean - gmean).reshape(-1, 1)
            w = mix.dot(n * w_inv.dot(m)).reshape(-1, 1)
            mw = m - w
            b_stats += weight * mix + weight * w.dot(w.T)
            b_count += weight
            w_stats += weight * n * mix + weight * n * mw.dot(mw.T)
            w_count += weight
        within = (1.0 / w_count) * w_stats
        between = (1.0 / b_count) * b_stats
    return (1.0 / stats.class_weight) * stats.sum, within, between


def plda_diagonalise(within, between):
    """plda_base.py:302-335: transform that whitens within_var and diagonalises between_var.
    Eigenvalues are left in numpy.linalg.eigh's ascending order, like PldaEstimation.get_output."""
    t1 = np.linalg.inv(np.linalg.cholesky(within))
    proj = t1.dot(between).dot(t1.T)
    s, U = np.linalg.eigh(proj)
    assert s.min() > 0
    return U.T.dot(t1), s


def plda_transform(x, mean, transform, psi, num_examples=1, normalize_length=True, simple_length_norm=False):
    """plda_base.py:93-107 + 165-172 for one vector x[dim]."""
    dim = x.shape[-1]
    y = transform.dot(x) - transform.dot(mean.reshape(-1))
    if simple_length_norm:
        factor = math.sqrt(dim) / np.linalg.norm(y)
    else:
        factor = math.sqrt(dim / np.dot(1.0 / (psi + 1.0 / num_examples), y ** 2))
    return factor * y if normalize_length else y


def plda_llr(train, n, test, psi):
    """plda_base.py:109-136."""
    dim = train.shape[0]
    mean = n * psi / (n * psi + 1.0) * train
    var = 1.0 + psi / (n * psi + 1.0)
    given = -0.5 * (np.sum(np.log(var)) + M_LOG_2PI * dim + np.sum((test - mean) ** 2 / var))
    var0 = psi + 1.0
    without = -0.5 * (np.sum(np.log(var0)) + M_LOG_2PI * dim + np.sum(test ** 2 / var0))
    return given - without


def two_cov_terms(between, within, mean):
    """gaussian-plda-scoring.py:31-50 (k = 0 as in the reference)."""
    tot_inv = np.linalg.inv(between + within)
    w2b_inv = np.linalg.inv(within + 2 * between)
    w_inv = np.linalg.inv(within)
    gamma = (-1 / 4) * (w2b_inv + w_inv) + (1 / 2) * tot_inv
    lam = (-1 / 4) * (w2b_inv - w_inv)
    c = (w2b_inv - tot_inv).dot(mean.reshape(-1, 1))
    return gamma, lam, c


def two_cov_score(e, t, gamma, lam, c):
    """gaussian-plda-scoring.py:23-29."""
    e, t = e.reshape(-1, 1), t.reshape(-1, 1)
    return float((e.T.dot(lam).dot(t) + t.T.dot(lam).dot(e) + e.T.dot(gamma).dot(e) + t.T.dot(gamma).dot(t) + (e + t).T.dot(c)).item())


# ---------------------------------------------------------------------------------- EER

def compute_eer(scores, labels):
    """computeEER-like-Bosaris.py:50-91: ascending sort of [score, label]; first point with
    FAR <= FRR; EER = mean of the two rates at whichever of {that point, the previous one} has
    the smaller |FAR - FRR|.  Returns (eer in [0,1], threshold)."""
    pairs = sorted([[float(s), int(l)] for s, l in zip(scores, labels)])
    num_p = sum(l for _, l in pairs)
    num_n = len(pairs) - num_p
    num_fa, num_fr = num_n, 0
    memory = None
    for score, label in pairs:
        if label == 1:
            num_fr += 1
        else:
            num_fa -= 1
        far, frr = num_fa / num_n, num_fr / num_p
        if far <= frr:
            if memory is None or abs(far - frr) <= abs(memory[0] - memory[1]):
                return (far + frr) / 2, score
            return (memory[0] + memory[1]) / 2, memory[2]
        memory = (far, frr, score)
    raise ValueError("FAR never drops to FRR")


# ------------------------------------------------------------------- score normalisation

def score_norm(enroll_cohort, test_cohort, ei, ti, scores, top_n=0, cross_select=False):
    """S-norm (top_n <= 0) / AS-norm of score/ScoreNormalization.py:70-179 on dense score matrices
    enroll_cohort [E, C], test_cohort [T, C] (the reference works on <key, key, score> text rows; the recipe
    scores every enrol / test vector against every cohort vector, gather_results_from_epochs.sh:103-183).
    pandas semantics: mean and *sample* std (ddof = 1) of the selected cohort scores, float64 (FINITE scores: pandas skips a NaN
    cohort score - and so does the device selection -, numpy's mean here would propagate it; no caller produces one);
    normed = 0.5 * ((s - mu_e) / sd_e + (s - mu_t) / sd_t)                       (lines 104-105, 171-175).
    AS-norm: the top_n largest cohort scores per row (124-126, 150-151); cross_select: the enrol-side statistics
    of trial (e, t) use the cohort vectors that are top_n for t and vice versa (139-148)."""
    ec = np.asarray(enroll_cohort, dtype=np.float64)
    tc = np.asarray(test_cohort, dtype=np.float64)
    s = np.asarray(scores, dtype=np.float64)
    n_c = ec.shape[1]
    n = n_c if top_n <= 0 else min(int(top_n), n_c)

    def top_idx(m):
        # stable descending order: among equal scores the lower cohort index first
        return np.argsort(-m, axis=1, kind="stable")[:, :n]

    def stats(v):
        with np.errstate(invalid="ignore", divide="ignore"):
            return v.mean(axis=-1), v.std(axis=-1, ddof=1)

    ie, it = top_idx(ec), top_idx(tc)
    if not cross_select:
        mu_e, sd_e = stats(np.take_along_axis(ec, ie, axis=1))
        mu_t, sd_t = stats(np.take_along_axis(tc, it, axis=1))
        return 0.5 * ((s - mu_e[ei]) / sd_e[ei] + (s - mu_t[ti]) / sd_t[ti])
    mu_e, sd_e = stats(np.take_along_axis(ec[ei], it[ti], axis=1))       # enrol scores at the test's top cohort
    mu_t, sd_t = stats(np.take_along_axis(tc[ti], ie[ei], axis=1))
    return 0.5 * ((s - mu_e) / sd_e + (s - mu_t) / sd_t)
