"""The developer build of the library (libasv_amd_dev.so, `make -C asv-subtools_amd/csrc dev`), when present, must resolve every symbol
at load time (RTLD_NOW) and export the C ABI of the product library: hipcc once dropped the host stub of one ablation instantiation
and the library only failed at dlopen on the GPU box."""

import ctypes
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEVLIB = os.path.join(REPO, "asv-subtools_amd", "libasv_amd_dev.so")


@pytest.mark.skipif(not os.path.exists(DEVLIB), reason="developer library not built")
def test_developer_library_resolves_all_symbols():
    from libs.amd import capi
    lib = ctypes.CDLL(DEVLIB, mode=os.RTLD_NOW)
    for name in capi.SYMBOLS:
        assert hasattr(lib, name), name
