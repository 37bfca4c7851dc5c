# -*- coding:utf-8 -*-
"""Scoring back-end on the MI355X: cosine (sub-mean + length-norm + dot products), Kaldi-style
PLDA (transform + log-likelihood ratio) and EER, through the C ABI of libasv_amd.so.

Replaces the process chain the reference runs per scoring job (bash -> Kaldi binaries):
    score/process.sh:156-203   ivector-mean, ivector-subtract-global-mean, ivector-normalize-length
    score/score.sh:82-121      ivector-compute-dot-products, ivector-plda-scoring
    computeEER.sh / computeEER-like-Bosaris.py:50-91
and the numpy PLDA of score/pyplda/plda_base.py: statistics + EM training (asv_plda_train, float64 on the device -
SURVEY.md 8(f) rank 4) and scoring; only the final D x D diagonalisation (Cholesky + eigh) is host numpy.

All functions take host numpy arrays or CUDA torch tensors; results stay on the device unless
`.cpu()` is called by the caller.
"""

import ctypes as C
import math
import os

import numpy as np

from . import capi


def _dev(x, dtype=None, device=None):
    import torch
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    if not t.is_cuda:
        t = t.cuda() if device is None else t.to(device)
    return t.contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _debug_checks():
    return os.environ.get("ASV_AMD_DEBUG_CHECKS", "0") not in ("0", "", "false")


def _trial_indices(enroll_idx, n_enroll, test_idx, n_test, device, what):
    """Trial indices address device rows directly (`enroll + ei * dim`): anything outside [0, n) is rejected BEFORE a kernel
    can read out of bounds.  A trial list is host data in every caller of the reference chain (it is parsed from the trials
    file): it is validated here on the host - numpy, no device work, no synchronisation - and then uploaded, so scoring
    stays asynchronous on the stream.  Lists that already live on the device are checked with ONE combined reduction and ONE
    read-back for both of them."""
    import torch
    pairs = ((enroll_idx, n_enroll, "enrol"), (test_idx, n_test, "test"))
    on_device = [isinstance(i, torch.Tensor) and i.is_cuda for i, _, _ in pairs]
    for (idx, n, side), dev_side in zip(pairs, on_device):
        if dev_side:
            continue
        a = idx.numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
        if a.size and (int(a.min()) < 0 or int(a.max()) >= n):
            raise ValueError("%s (%s): trial index range [%d, %d] outside the %d vectors" % (what, side, int(a.min()), int(a.max()), n))
    ei, ti = _dev(enroll_idx, torch.int32, device), _dev(test_idx, torch.int32, device)
    dev_lists = [(t, n, side) for t, (_, n, side), d in zip((ei, ti), pairs, on_device) if d and t.numel()]
    if dev_lists:
        stats = torch.stack([x for t, _, _ in dev_lists for x in (t.min(), t.max())]).cpu().tolist()      # one read-back
        for k, (_, n, side) in enumerate(dev_lists):
            lo, hi = int(stats[2 * k]), int(stats[2 * k + 1])
            if lo < 0 or hi >= n:
                raise ValueError("%s (%s): trial index range [%d, %d] outside the %d vectors" % (what, side, lo, hi, n))
    return ei, ti


def mean_vector(x):
    """ivector-mean over all vectors (score/process.sh:177)."""
    import torch
    x = _dev(x, torch.float32)
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    capi.check(capi.lib().asv_mean_vec(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _stream(x)), "asv_mean_vec")
    return out


def length_normalize(x, mean=None, normalize=True):
    """[x - mean] then x / ||x|| (ivector-subtract-global-mean + ivector-normalize-length
    --scaleup=false); returns a new device tensor."""
    import torch
    x = _dev(x, torch.float32).clone()
    m = _dev(mean, torch.float32, x.device) if mean is not None else None
    capi.check(capi.lib().asv_length_norm(_ptr(x), x.shape[0], x.shape[1], _ptr(m), int(normalize), _stream(x)), "asv_length_norm")
    return x


def score_matrix(enroll, test):
    """All-pairs dot products S[i, j] = <enroll_i, test_j> (exact-f32 MFMA GEMM)."""
    import torch
    e, t = _dev(enroll, torch.float32), _dev(test, torch.float32)
    assert e.shape[1] == t.shape[1]
    out = torch.empty((e.shape[0], t.shape[0]), dtype=torch.float32, device=e.device)
    capi.check(capi.lib().asv_dot_score_matrix(_ptr(e), e.shape[0], _ptr(t), t.shape[0], e.shape[1], _ptr(out), _stream(e)), "asv_dot_score_matrix")
    return out


def score_trials(enroll, test, enroll_idx, test_idx):
    """Dot product per trial (ivector-compute-dot-products over a trial list)."""
    import torch
    e, t = _dev(enroll, torch.float32), _dev(test, torch.float32)
    ei, ti = _trial_indices(enroll_idx, e.shape[0], test_idx, t.shape[0], e.device, "score_trials")
    out = torch.empty(ei.shape[0], dtype=torch.float32, device=e.device)
    capi.check(capi.lib().asv_dot_score_trials(_ptr(e), _ptr(t), e.shape[1], _ptr(ei), _ptr(ti), ei.shape[0], _ptr(out), _stream(e)), "asv_dot_score_trials")
    return out


def speaker_mean(vectors, groups):
    """Speaker-level enrolment vectors: `ivector-mean ark:spk2utt ...` of score/process.sh:156-167.  groups: one sequence of
    row indices per speaker (the spk2utt map resolved to rows of `vectors`).  Returns (means [G, dim] f32 device tensor,
    num_utts [G] int32 device tensor - what score.sh:99-121 hands to ivector-plda-scoring as --num-utts and what
    Plda.transform_vectors(num_examples=...) / llr_trials(enroll_num_utts=...) take)."""
    import torch
    x = _dev(vectors, torch.float32)
    groups = [np.asarray(g, dtype=np.int64).reshape(-1) for g in groups]
    for g in groups:
        if g.size == 0:
            raise ValueError("speaker_mean: a speaker without utterances (ivector-mean skips it with a warning; drop it from spk2utt)")
        if g.min() < 0 or g.max() >= x.shape[0]:
            raise ValueError("speaker_mean: utterance index outside the %d vectors" % x.shape[0])
    offsets = np.zeros(len(groups) + 1, dtype=np.int32)
    np.cumsum([g.size for g in groups], out=offsets[1:])
    order = np.concatenate(groups).astype(np.int32) if groups else np.zeros(0, dtype=np.int32)
    od, of = _dev(order, torch.int32, x.device), _dev(offsets, torch.int32, x.device)
    means = torch.empty((len(groups), x.shape[1]), dtype=torch.float32, device=x.device)
    counts = torch.empty(len(groups), dtype=torch.int32, device=x.device)
    capi.check(capi.lib().asv_group_mean(_ptr(x), x.shape[0], x.shape[1], _ptr(od), _ptr(of), len(groups), _ptr(means), _ptr(counts), _stream(x)),
               "asv_group_mean")
    return means, counts


def read_spk2utt(path, utt_keys):
    """Kaldi spk2utt text ('spk utt1 utt2 ...') -> (speaker ids, groups of row indices into `utt_keys`)."""
    row = {k: i for i, k in enumerate(utt_keys)}
    spks, groups = [], []
    with open(path) as f:
        for line in f:
            toks = line.split()
            if len(toks) >= 2:
                spks.append(toks[0])
                groups.append([row[u] for u in toks[1:]])
    return spks, groups


def cosine_trials(enroll, test, enroll_idx, test_idx, submean=None):
    """The reference's cosine recipe: [submean] -> norm -> dot (scoreSets.sh "submean-norm" +
    score.sh cosine).  `submean`: vector to subtract from both sides, or None."""
    e = length_normalize(enroll, submean)
    t = length_normalize(test, submean)
    return score_trials(e, t, enroll_idx, test_idx)


def eer(scores, labels):
    """Equal error rate in percent + threshold (computeEER-like-Bosaris.py semantics)."""
    import torch
    s = _dev(scores, torch.float32)
    l = _dev(labels, torch.int32, s.device)
    e, thr = C.c_float(0), C.c_float(0)
    capi.check(capi.lib().asv_eer(_ptr(s), _ptr(l), s.shape[0], C.byref(e), C.byref(thr), _stream(s)), "asv_eer")
    return float(e.value), float(thr.value)


def score_normalize(scores, enroll_cohort, test_cohort, enroll_idx, test_idx, top_n=300, cross_select=False):
    """S-norm / AS-norm of trial scores (reference score/ScoreNormalization.py:70-179, there a pandas groupby +
    per-trial loop over text score files).  enroll_cohort [E, C] / test_cohort [T, C]: scores of every enrolment /
    test vector against the cohort (e.g. `score_matrix(enroll, cohort)`); top_n <= 0 selects S-norm (all cohort
    scores), otherwise AS-norm over the top_n largest; cross_select as in the reference's --cross-select."""
    import torch
    s = _dev(scores, torch.float32)
    ec, tc = _dev(enroll_cohort, torch.float32, s.device), _dev(test_cohort, torch.float32, s.device)
    assert ec.dim() == 2 and tc.dim() == 2 and ec.shape[1] == tc.shape[1], "cohort score matrices must share the cohort axis"
    ei, ti = _trial_indices(enroll_idx, ec.shape[0], test_idx, tc.shape[0], s.device, "score_normalize")
    assert ei.shape[0] == s.shape[0] == ti.shape[0]
    # two full passes + two blocking read-backs: debug switch only (ASV_AMD_DEBUG_CHECKS=1); cohort scores are dot products of
    # finite, length-normalised vectors in every caller of this module.  (Without the check a NaN cohort score is SKIPPED by the
    # device selection - which is also what the reference's pandas groupby().mean() / .std() / sort_values().head() do with it:
    # skipna, NaN sorted last.)
    if _debug_checks() and (bool(torch.isnan(ec).any().item()) or bool(torch.isnan(tc).any().item())):
        raise ValueError("score_normalize: NaN in the cohort scores (the device selection orders keys and skips it, as the reference's "
                         "pandas groupby().mean() / .std() / sort_values().head() do - silently in both; ASV_AMD_DEBUG_CHECKS=1 makes it loud)")
    out = torch.empty_like(s)
    capi.check(capi.lib().asv_score_norm(_ptr(ec), ec.shape[0], _ptr(tc), tc.shape[0], ec.shape[1], _ptr(ei), _ptr(ti), _ptr(s), s.shape[0],
                                         int(top_n), int(bool(cross_select)), _ptr(out), _stream(s)), "asv_score_norm")
    return out


def cosine_asnorm_trials(enroll, test, cohort, enroll_idx, test_idx, submean=None, top_n=300, cross_select=False):
    """The published-EER protocol (recipe/voxcelebSRC/gather_results_from_epochs.sh:103-183): submean -> norm ->
    cosine for the trials and for enrol x cohort / test x cohort, then AS-norm, all on the device."""
    e, t, c = length_normalize(enroll, submean), length_normalize(test, submean), length_normalize(cohort, submean)
    raw = score_trials(e, t, enroll_idx, test_idx)
    return score_normalize(raw, score_matrix(e, c), score_matrix(t, c), enroll_idx, test_idx, top_n=top_n, cross_select=cross_select)


# -------------------------------------------------------------------------------------- PLDA

class Plda(object):
    """Diagonalised PLDA model: y = transform (x - mean), between-class variances `psi`
    (plda_base.py PLDA class, Kaldi ivector/plda.h)."""

    def __init__(self, mean, transform, psi):
        self.mean = np.asarray(mean, dtype=np.float64).reshape(-1)
        self.transform = np.asarray(transform, dtype=np.float64)
        self.psi = np.asarray(psi, dtype=np.float64).reshape(-1)
        self.dim = self.mean.shape[0]

    @classmethod
    def from_covariances(cls, mean, within_var, between_var):
        """Simultaneous diagonalisation (plda_base.py:302-335): inv(chol(W)), eigh."""
        within_var = np.asarray(within_var, dtype=np.float64)
        t1 = np.linalg.inv(np.linalg.cholesky(within_var))
        s, U = np.linalg.eigh(t1.dot(np.asarray(between_var, dtype=np.float64)).dot(t1.T))
        if s.min() <= 0:
            raise ValueError("between-class covariance is not positive definite after whitening")
        return cls(mean, U.T.dot(t1), s)

    @classmethod
    def read_stats_ark(cls, path):
        """The 'mean' / 'within_var' / 'between_var' ark PldaEstimation.plda_write produces (337-342)."""
        from libs.support import kaldi_io
        parts = {k: np.array(v, dtype=np.float64) for k, v in kaldi_io.read_vec_flt_ark(path)}
        dim = parts["mean"].shape[0]
        return cls.from_covariances(parts["mean"], parts["within_var"].reshape(dim, dim), parts["between_var"].reshape(dim, dim))

    def write_kaldi_text(self, path):
        """Kaldi text <Plda> (plda_base.py:216-225) - what ivector-copy-plda / ivector-plda-scoring read."""
        with open(path, "w") as f:
            f.write("<Plda>  [ " + " ".join(map(str, self.mean)) + " ]\n [")
            for row in self.transform:
                f.write("\n  " + " ".join(map(str, row)))
            f.write(" ]\n [ " + " ".join(map(str, self.psi)) + " ]\n</Plda> ")

    def adapt_unsupervised(self, vectors, mean_diff_scale=1.0, within_covar_scale=0.3, between_covar_scale=0.7):
        """Kaldi-style unsupervised domain adaptation, plda_base.py:344-485 (PldaUnsupervisedAdaptor.add_stats + update_plda):
        the statistics of the adaptation vectors come from the device (asv_scatter_f64), the dim x dim algebra follows the
        reference step by step in float64 (its np.linalg.eig calls on symmetric matrices are eigh here: same spectrum, ordered).
        Returns a new Plda; eigenvalue order / eigenvector signs are free, scores are not affected."""
        total, xtx = second_moments(vectors)
        n = float(vectors.shape[0])
        dim = self.dim
        mean = total / n
        variance = xtx / n - np.outer(mean, mean)
        mean_diff = mean - self.mean
        variance = variance + mean_diff_scale * np.outer(mean_diff, mean_diff)
        transform_mod = self.transform / np.sqrt(1.0 + self.psi)[:, None]
        variance_proj = transform_mod.dot(variance).dot(transform_mod.T)
        s, P = np.linalg.eigh(0.5 * (variance_proj + variance_proj.T))
        W = np.diag(1.0 / (1.0 + self.psi))
        B = np.diag(self.psi / (1.0 + self.psi))
        Wp, Bp = P.T.dot(W).dot(P), P.T.dot(B).dot(P)
        excess = np.clip(s - 1.0, 0.0, None)
        Wp[np.arange(dim), np.arange(dim)] += excess * within_covar_scale
        Bp[np.arange(dim), np.arange(dim)] += excess * between_covar_scale
        inv = np.linalg.inv(P.T.dot(transform_mod))
        Wmod, Bmod = inv.dot(Wp).dot(inv.T), inv.dot(Bp).dot(inv.T)
        c_inv = np.linalg.inv(np.linalg.cholesky(0.5 * (Wmod + Wmod.T)))
        bproj = c_inv.dot(Bmod).dot(c_inv.T)
        psi_new, Q = np.linalg.eigh(0.5 * (bproj + bproj.T))
        return Plda(mean, Q.T.dot(c_inv), psi_new)

    def transform_vectors(self, x, num_examples=None, normalize_length=True, simple_length_norm=False):
        """plda_base.py:93-107 for a whole set at once; returns a device tensor [n, dim]."""
        import torch
        x = _dev(x, torch.float32)
        dev = x.device
        mean = _dev(self.mean.astype(np.float32), device=dev)
        tr = _dev(self.transform.astype(np.float32), device=dev)
        psi = _dev(self.psi.astype(np.float32), device=dev)
        ne = _dev(num_examples, torch.int32, dev) if num_examples is not None else None
        mode = capi.PLDA_NORM_NONE if not normalize_length else (capi.PLDA_NORM_SIMPLE if simple_length_norm else capi.PLDA_NORM_PSI)
        out = torch.empty((x.shape[0], self.dim), dtype=torch.float32, device=dev)
        capi.check(capi.lib().asv_plda_transform(_ptr(x), x.shape[0], self.dim, _ptr(mean), _ptr(tr), _ptr(psi), _ptr(ne), mode, _ptr(out), _stream(x)),
                   "asv_plda_transform")
        return out

    def llr_trials(self, enroll_t, test_t, enroll_idx, test_idx, enroll_num_utts=None):
        """plda_base.py:109-136 per trial on already-transformed vectors."""
        import torch
        e, t = _dev(enroll_t, torch.float32), _dev(test_t, torch.float32)
        dev = e.device
        psi = _dev(self.psi.astype(np.float32), device=dev)
        ei, ti = _trial_indices(enroll_idx, e.shape[0], test_idx, t.shape[0], dev, "llr_trials")
        en = _dev(enroll_num_utts, torch.int32, dev) if enroll_num_utts is not None else None
        if en is not None and en.shape[0] != e.shape[0]:
            raise ValueError("llr_trials: %d enrolment vectors but %d num_utts entries" % (e.shape[0], en.shape[0]))
        out = torch.empty(ei.shape[0], dtype=torch.float32, device=dev)
        capi.check(capi.lib().asv_plda_llr_trials(_ptr(e), _ptr(t), self.dim, _ptr(psi), _ptr(en), _ptr(ei), _ptr(ti), ei.shape[0], _ptr(out), _stream(e)),
                   "asv_plda_llr_trials")
        return out


class TwoCovPlda(object):
    """Two-covariance PLDA scorer of score/pyplda/gaussian-plda-scoring.py: Gamma / Lambda / c from (mean, within, between)
    as CalculateVar does (31-50; main() adds 5e-5 I to within_var first, 66) on the host in float64 - D x D, once per model -
    and the per-trial form (23-29) on the device in float64 (asv_two_cov_trials)."""

    def __init__(self, mean, within_var, between_var, within_ridge=5e-5):
        mean = np.asarray(mean, dtype=np.float64).reshape(-1)
        within = np.asarray(within_var, dtype=np.float64) + within_ridge * np.eye(mean.shape[0])
        between = np.asarray(between_var, dtype=np.float64)
        tot_inv = np.linalg.inv(between + within)
        w2b_inv = np.linalg.inv(within + 2 * between)
        w_inv = np.linalg.inv(within)
        self.dim = mean.shape[0]
        self.gamma = np.ascontiguousarray((-1 / 4) * (w2b_inv + w_inv) + (1 / 2) * tot_inv)
        self.lam = np.ascontiguousarray((-1 / 4) * (w2b_inv - w_inv))
        self.c = np.ascontiguousarray((w2b_inv - tot_inv).dot(mean))

    @classmethod
    def read_stats_ark(cls, path, within_ridge=5e-5):
        from libs.support import kaldi_io
        parts = {k: np.array(v, dtype=np.float64) for k, v in kaldi_io.read_vec_flt_ark(path)}
        dim = parts["mean"].shape[0]
        return cls(parts["mean"], parts["within_var"].reshape(dim, dim), parts["between_var"].reshape(dim, dim), within_ridge)

    def score_trials(self, enroll, test, enroll_idx, test_idx):
        """float64 device tensor [n_trials]."""
        import torch
        e, t = _dev(enroll, torch.float32), _dev(test, torch.float32)
        ei, ti = _trial_indices(enroll_idx, e.shape[0], test_idx, t.shape[0], e.device, "two_cov")
        if e.shape[1] != self.dim or t.shape[1] != self.dim:
            raise ValueError("two_cov: %d-dimensional model, vectors of %d / %d" % (self.dim, e.shape[1], t.shape[1]))
        out = torch.empty(ei.shape[0], dtype=torch.float64, device=e.device)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        with torch.cuda.device(e.device):
            capi.check(capi.lib().asv_two_cov_trials(_ptr(e), e.shape[0], _ptr(t), t.shape[0], self.dim, dp(self.gamma), dp(self.lam), dp(self.c),
                                                     _ptr(ei), _ptr(ti), ei.shape[0], _ptr(out), _stream(e)), "asv_two_cov_trials")
        return out


def second_moments(vectors):
    """(sum [dim], X^T X [dim, dim]) of a set of vectors, accumulated in float64 on the device (asv_scatter_f64)."""
    import torch
    x = _dev(vectors, torch.float32)
    dim = x.shape[1]
    total, xtx = np.zeros(dim), np.zeros((dim, dim))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with torch.cuda.device(x.device):
        capi.check(capi.lib().asv_scatter_f64(_ptr(x), x.stride(0), x.shape[0], dim, dp(total), dp(xtx), _stream(x)), "asv_scatter_f64")
    return total, xtx


def train_lda(vectors, labels, lda_dim, total_covariance_factor=0.0, covariance_floor=1.0e-6):
    """Kaldi ivector-compute-lda (score/process.sh:218-229 runs it with --total-covariance-factor=0.1; Kaldi itself is not
    vendored in the reference: restated from ivectorbin/ivector-compute-lda.cc, PARITY UNPINNED).  Class statistics come from
    the device (asv_class_scatter_f64); the dim x dim algebra is host float64: subtract the global mean, whiten
    factor * total + (1 - factor) * within covariance (eigenvalues floored at floor * largest), diagonalise the projected
    between-class covariance, keep the lda_dim strongest directions.  Returns the [lda_dim, dim + 1] affine matrix
    ivector-transform applies (last column = offset -A mean); apply_affine(x, mat) applies it on the device."""
    import torch
    x = _dev(vectors, torch.float32)
    order, offsets = group_rows_by_class(labels)
    dim, n = x.shape[1], x.shape[0]
    total, xtx, cls = np.zeros(dim), np.zeros((dim, dim)), np.zeros((dim, dim))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with torch.cuda.device(x.device):
        capi.check(capi.lib().asv_class_scatter_f64(_ptr(x), x.stride(0), n, dim, order.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    offsets.ctypes.data_as(C.POINTER(C.c_longlong)), len(offsets) - 1, dp(total), dp(xtx), dp(cls),
                                                    _stream(x)), "asv_class_scatter_f64")
    mean = total / n
    mm = n * np.outer(mean, mean)
    total_covar = (xtx - mm) / n
    between = (cls - mm) / n
    within = total_covar - between
    mat = total_covariance_factor * total_covar + (1.0 - total_covariance_factor) * within
    s, U = np.linalg.eigh(0.5 * (mat + mat.T))
    s = np.maximum(s, covariance_floor * s.max())
    T = (U / np.sqrt(s)).T                                        # diag(s^-1/2) U^T: T mat T^T = I
    proj = T.dot(between).dot(T.T)
    e, V = np.linalg.eigh(0.5 * (proj + proj.T))
    keep = np.argsort(-e)[:lda_dim]
    A = V[:, keep].T.dot(T)
    return np.concatenate([A, -A.dot(mean)[:, None]], axis=1)


def apply_affine(x, mat):
    """ivector-transform with a [rows, dim + 1] matrix (last column = offset) or a [rows, dim] linear one, rows <= dim, on the
    device; returns [n, rows]."""
    mat = np.asarray(mat, dtype=np.float64)
    dim = np.asarray(x.shape)[1]
    rows = mat.shape[0]
    if mat.shape[1] not in (dim, dim + 1) or rows > dim:
        raise ValueError("apply_affine: matrix %s does not fit %d-dimensional vectors" % (mat.shape, dim))
    A = np.zeros((dim, dim))
    A[:rows] = mat[:, :dim]
    shift = None
    if mat.shape[1] == dim + 1:                                   # A x + b = A (x - m) with A m = -b (b lies in the row space image of A)
        shift = np.linalg.lstsq(mat[:, :dim], -mat[:, dim], rcond=None)[0]
    return linear_transform(x, A, shift)[:, :rows]


def zca_whitening(vectors, regularization=1e-6, center=True):
    """ZCA whitening matrices of score/whiten/*.py (class ZCA.fit): covariance X^T X / (n - 1) of the [centred] vectors from
    the device statistics, eigen-decomposition of the dim x dim covariance on the host.  train_ZCA_Whitening.py does not
    centre (center=False), do_ZCA_Whitening.py does.  Returns (mean or zeros, whiten, dewhiten); apply with
    linear_transform(x, whiten, mean)."""
    total, xtx = second_moments(vectors)
    n = np.asarray(vectors).shape[0] if not hasattr(vectors, "shape") else vectors.shape[0]
    mean = total / n if center else np.zeros_like(total)
    cov = (xtx - n * np.outer(mean, mean)) / (n - 1) if center else xtx / (n - 1)
    cov = 0.5 * (cov + cov.T)
    S, U = np.linalg.eigh(cov)                     # symmetric PSD: the SVD of the reference (scipy.linalg.svd) up to ordering / signs
    s = np.sqrt(np.clip(S, regularization, None))
    return mean, (U / s).dot(U.T), (U * s).dot(U.T)


def linear_transform(x, matrix, mean=None):
    """y = matrix (x - mean) for every row, on the device (the transform kernel of the PLDA back-end without normalisation)."""
    import torch
    x = _dev(x, torch.float32)
    dev = x.device
    matrix = np.asarray(matrix, dtype=np.float32)
    if matrix.shape[0] != matrix.shape[1] or matrix.shape[1] != x.shape[1]:
        raise ValueError("linear_transform: a square [dim, dim] matrix is expected (dimension-reducing transforms: slice the result)")
    m = _dev(np.zeros(x.shape[1], dtype=np.float32) if mean is None else np.asarray(mean, dtype=np.float32), device=dev)
    tr = _dev(matrix, device=dev)
    psi = _dev(np.ones(x.shape[1], dtype=np.float32), device=dev)
    out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=dev)
    capi.check(capi.lib().asv_plda_transform(_ptr(x), x.shape[0], x.shape[1], _ptr(m), _ptr(tr), _ptr(psi), None, capi.PLDA_NORM_NONE, _ptr(out), _stream(x)),
               "asv_plda_transform")
    return out


def group_rows_by_class(labels):
    """Host bookkeeping of train_plda: (order int32 [N]: row indices grouped by class, the classes in ascending size as the
    reference's sorted PldaStats wants them (plda_base.py:68-81, 236-240); offsets int64 [K+1] into `order`)."""
    labels = np.asarray(labels)
    classes, inv, counts = np.unique(labels, return_inverse=True, return_counts=True)
    by_size = np.argsort(counts, kind="stable")                  # classes in ascending size
    rank = np.empty(len(classes), dtype=np.int64)
    rank[by_size] = np.arange(len(classes))
    order = np.argsort(rank[inv], kind="stable").astype(np.int32)
    offsets = np.zeros(len(classes) + 1, dtype=np.int64)
    np.cumsum(counts[by_size], out=offsets[1:])
    return order, offsets


def train_plda(vectors, labels, num_iters=10):
    """PLDA statistics + EM (plda_base.py:37-81, 227-300) in float64 on the device (asv_plda_train): vectors [N, dim]
    (host array or CUDA tensor, f32 like the extracted embeddings), labels [N] (any hashable ids).  The host only groups
    row indices by class and orders the classes by size, as the reference's sorted PldaStats does.
    Returns (mean [dim], within_var [dim, dim], between_var [dim, dim]) float64 numpy arrays."""
    import torch
    x = _dev(vectors, torch.float32)
    labels = np.asarray(labels)
    if labels.shape[0] != x.shape[0]:
        raise ValueError("train_plda: %d vectors but %d labels" % (x.shape[0], labels.shape[0]))
    order, offsets = group_rows_by_class(labels)
    dim = x.shape[1]
    mean, within, between = np.zeros(dim), np.zeros((dim, dim)), np.zeros((dim, dim))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    with torch.cuda.device(x.device):
        capi.check(capi.lib().asv_plda_train(_ptr(x), x.stride(0), x.shape[0], dim, order.ctypes.data_as(C.POINTER(C.c_int32)),
                                             offsets.ctypes.data_as(C.POINTER(C.c_longlong)), len(offsets) - 1, int(num_iters),
                                             dp(mean), dp(within), dp(between), _stream(x)), "asv_plda_train")
    return mean, within, between
