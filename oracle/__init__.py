"""CPU oracle for the embedding-extraction + scoring hot path of Snowdar/asv-subtools.

TEST INFRASTRUCTURE ONLY.  Nothing under `asv-subtools_amd/` may import this package.  The
only legitimate callers are `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` - and there only as the checker / the timed CPU baseline, never as the thing
shipped.  The product path fails loudly when the HIP library is missing.

Parity pinning: the reference has no golden vectors or known-answer tests for this path
(SURVEY.md section 4 / 8(c)), so every function here is pinned against outputs of the
reference implementation itself, run in the build container by `oracle/gen_golden.py`
(which imports /root/reference read-only) and committed under `tests/golden/`.
`tests/test_oracle_golden.py` re-checks the oracle against those fixtures on every run.
"""
