import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "asv-subtools_amd", "pytorch")
for p in (REPO, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
# the library reads its developer switches (kernel variants for A/B) once per process; the tests flip them between launches
os.environ.setdefault("ASV_AMD_LIVE_TUNE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


@pytest.fixture(scope="session")
def repo_root():
    return REPO
