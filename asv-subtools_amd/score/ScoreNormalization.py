# -*- coding:utf-8 -*-
"""S-norm / AS-norm of trial scores on an MI355X - command-line compatible with the reference's
score/ScoreNormalization.py (options 27-56 there; called from recipe/voxcelebSRC/gather_results_from_epochs.sh:
103-183 as `python3 subtools/score/ScoreNormalization.py --top-n=300 --cross-select=true trials.score
enroll_cohort.score test_cohort.score out.score`).

Score files are text rows `<key1> <key2> <score>`.  The reference groups them with pandas and walks the trials in
a Python loop; here the cohort scores become two dense matrices (every enrolment / test key must be scored against
the same cohort keys, which is what the recipe produces) and libasv_amd.so does the selection and normalisation
(asv_score_norm).  There is no CPU path: without a ROCm device or the library this exits non-zero.
"""

import argparse
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch"))


def get_args():
    parser = argparse.ArgumentParser(description="Score Normalization.")
    parser.add_argument("--method", default="asnorm", type=str, choices=["snorm", "asnorm"], help="Choices to select a score normalization.")
    parser.add_argument("--top-n", type=int, default=300, help="Used in AS-Norm.")
    parser.add_argument("--second-cohort", type=str, default="true", choices=["true", "false"],
                        help="If true, get cohort key from the second field of score.")
    parser.add_argument("--cross-select", type=str, default="false", choices=["true", "false"],
                        help="Used in AS-Norm. If true, select top n enroll/test keys by test/enroll_cohort scores.")
    parser.add_argument("input_score", metavar="enroll-test-score", type=str, help="Original score path for <enroll, test>.")
    parser.add_argument("enroll_cohort_score", metavar="enroll-cohort-score", type=str, help="Score file path for <enroll, cohort>.")
    parser.add_argument("test_cohort_score", metavar="test-cohort-score", type=str, help="Score file path for <test, cohort>.")
    parser.add_argument("output_score", metavar="output-score-path", type=str, help="Output score path for <enroll, test> after score normalization.")
    return parser.parse_args()


def read_score_file(path):
    """-> (keys1, keys2, scores float64) of a `<key1> <key2> <score>` text file."""
    k1, k2, sc = [], [], []
    with open(path) as f:
        for n, line in enumerate(f, 1):
            parts = line.split()
            if not parts:
                continue
            if len(parts) != 3:
                raise ValueError("%s:%d: expected '<key> <key> <score>', got %r" % (path, n, line.rstrip("\n")))
            k1.append(parts[0]); k2.append(parts[1]); sc.append(float(parts[2]))
    return k1, k2, np.asarray(sc, dtype=np.float64)


def dense_cohort_matrix(path, cohort_second):
    """Rows = the non-cohort keys in order of first appearance, columns = cohort keys; every pair must be present once."""
    a, b, sc = read_score_file(path)
    keys, cohort = (a, b) if cohort_second else (b, a)
    row_of, col_of = {}, {}
    for k in keys:
        row_of.setdefault(k, len(row_of))
    for c in cohort:
        col_of.setdefault(c, len(col_of))
    m = np.full((len(row_of), len(col_of)), np.nan, dtype=np.float32)
    r = np.fromiter((row_of[k] for k in keys), dtype=np.int64, count=len(keys))
    c = np.fromiter((col_of[k] for k in cohort), dtype=np.int64, count=len(cohort))
    m[r, c] = sc
    if len(sc) != m.size or np.isnan(m).any():
        raise ValueError("%s: %d rows for %d x %d key pairs - the MI355X path needs every key scored against every cohort key exactly once"
                         % (path, len(sc), m.shape[0], m.shape[1]))
    return m, row_of, col_of


def main():
    print(" ".join(sys.argv))
    args = get_args()
    try:
        from libs.amd import scoring
        second = args.second_cohort == "true"
        ec, enroll_row, cohort_e = dense_cohort_matrix(args.enroll_cohort_score, second)
        tc, test_row, cohort_t = dense_cohort_matrix(args.test_cohort_score, second)
        if cohort_e.keys() != cohort_t.keys():
            raise ValueError("enroll-cohort and test-cohort files use different cohort keys")
        if list(cohort_e) != list(cohort_t):                       # same set, other column order: align test to enrol
            tc = tc[:, [cohort_t[k] for k in cohort_e]]
        e_keys, t_keys, sc = read_score_file(args.input_score)
        try:
            ei = np.fromiter((enroll_row[k] for k in e_keys), dtype=np.int32, count=len(e_keys))
            ti = np.fromiter((test_row[k] for k in t_keys), dtype=np.int32, count=len(t_keys))
        except KeyError as e:
            raise ValueError("trial key %s has no cohort scores" % e)
        top_n = 0 if args.method == "snorm" else args.top_n
        out = scoring.score_normalize(sc.astype(np.float32), ec, tc, ei, ti, top_n=top_n,
                                      cross_select=(args.method == "asnorm" and args.cross_select == "true")).cpu().numpy()
        with open(args.output_score, "w") as f:
            for a, b, v in zip(e_keys, t_keys, out):
                f.write("%s %s %s\n" % (a, b, repr(float(v))))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
