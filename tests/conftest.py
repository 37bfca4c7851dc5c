import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "asv-subtools_amd", "pytorch")
for p in (REPO, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
# ASV_AMD_LIVE_TUNE is NOT set here: the suite exercises the library as production runs it (developer switches read once per
# process; the product library has no ablation code at all).  The kernel variants of the developer build are tested by
# tests/test_gpu_devlib.py in a subprocess of its own.


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


@pytest.fixture(scope="session")
def repo_root():
    return REPO
