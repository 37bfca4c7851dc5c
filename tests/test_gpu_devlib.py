"""Runs tests/devlib_cases.py against the developer build of the library (libasv_amd_dev.so: ablation instantiations, the chain
kernel's first pooling epilogue, the four-wave chain kernel - none of which is in libasv_amd.so) in a subprocess; skipped when
that build does not exist (`make -C asv-subtools_amd/csrc dev`)."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVLIB = os.path.join(REPO, "asv-subtools_amd", "libasv_amd_dev.so")


@pytest.mark.skipif(not os.path.exists(DEVLIB), reason="developer library not built (make -C asv-subtools_amd/csrc dev)")
def test_developer_build_variants():
    env = dict(os.environ, ASV_AMD_LIB=DEVLIB, ASV_AMD_LIVE_TUNE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "devlib_cases.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
