# -*- coding:utf-8 -*-
"""ctypes face of libasv_io.so (include/asv_io.h, csrc/host_io.c): batched positioned reads on native threads - what the sharded
extraction path fills its packed batch buffers with.  Host I/O only: nothing here touches the device or the arithmetic."""

import ctypes as C
import os

_LIB = None


def library_path():
    env = os.environ.get("ASV_AMD_IO_LIB")
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    # .../asv-subtools_amd/pytorch/libs/support -> .../asv-subtools_amd/libasv_io.so
    return os.path.normpath(os.path.join(here, "..", "..", "..", "libasv_io.so"))


def lib():
    """Loads libasv_io.so once; None when it has not been built (callers then read with os.preadv from Python - same bytes,
    one utterance per call)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            _LIB = False
        else:
            L = C.CDLL(path)
            L.asv_io_version.restype = C.c_int
            L.asv_io_last_errno.restype = C.c_int
            L.asv_io_pread_batch.restype = C.c_int
            L.asv_io_pread_batch.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_int]
            _LIB = L
    return _LIB or None


def pread_batch(fds, offsets, nbytes, base_address, dst_offsets, threads=4):
    """Reads nbytes[i] bytes of descriptor fds[i] at file offset offsets[i] to address base_address + dst_offsets[i], for every i, on
    `threads` native threads (the GIL is released for the whole call).  Raises OSError naming the first read that failed."""
    import numpy as np
    L = lib()
    n = len(fds)
    fd = np.ascontiguousarray(fds, dtype=np.int32)
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    nb = np.ascontiguousarray(nbytes, dtype=np.int64)
    dst = (np.ascontiguousarray(dst_offsets, dtype=np.int64) + int(base_address)).astype(np.uint64)
    rc = L.asv_io_pread_batch(n, fd.ctypes.data_as(C.POINTER(C.c_int32)), off.ctypes.data_as(C.POINTER(C.c_int64)), nb.ctypes.data_as(C.POINTER(C.c_int64)),
                              dst.ctypes.data_as(C.POINTER(C.c_void_p)), int(threads))
    if rc != 0:
        err = L.asv_io_last_errno()
        raise OSError(err, "read %d of the batch failed (%s)" % (-rc - 1, os.strerror(err) if err else "the file ends inside the matrix"), -rc - 1)
