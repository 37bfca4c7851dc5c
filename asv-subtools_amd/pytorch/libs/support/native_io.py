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
            L.asv_io_scan_ark.restype = C.c_int64
            L.asv_io_scan_ark.argtypes = [C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
            if L.asv_io_version() >= 2:
                L.asv_io_pack_vec_ark.restype = C.c_int64
                L.asv_io_pack_vec_ark.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
            if L.asv_io_version() >= 3:
                L.asv_io_parse_scp.restype = C.c_int64
                L.asv_io_parse_scp.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
            _LIB = L
    return _LIB or None


def parse_scp(data, path_cap=4096):
    """One native pass over the bytes of a Kaldi scp table (asv_io_parse_scp): dict of arrays key_off / key_len / rx_off / rx_len (spans in
    `data`), path_id (-1: not of the plain 'path:offset' form), offset, and `paths` (the distinct path strings); None when the library
    is too old or a capacity is exceeded (the caller parses in Python)."""
    import numpy as np
    L = lib()
    if L is None or L.asv_io_version() < 3:
        return None
    cap = data.count(b"\n") + 1
    key_off, rx_off, offset = (np.empty(cap, dtype=np.int64) for _ in range(3))
    key_len, rx_len, path_id = (np.empty(cap, dtype=np.int32) for _ in range(3))
    path_off, path_len = np.empty(path_cap, dtype=np.int64), np.empty(path_cap, dtype=np.int32)
    n_paths = C.c_int32(0)
    n = int(L.asv_io_parse_scp(data, len(data), cap, key_off.ctypes.data, key_len.ctypes.data, rx_off.ctypes.data, rx_len.ctypes.data, path_id.ctypes.data,
                               offset.ctypes.data, path_cap, path_off.ctypes.data, path_len.ctypes.data, C.byref(n_paths)))
    if n < 0:
        return None
    paths = [data[int(path_off[k]):int(path_off[k]) + int(path_len[k])].decode("latin1") for k in range(n_paths.value)]
    return {"n": n, "key_off": key_off[:n], "key_len": key_len[:n], "rx_off": rx_off[:n], "rx_len": rx_len[:n], "path_id": path_id[:n], "offset": offset[:n],
            "paths": paths}


def pack_vec_ark(keys, vectors, as_array=False):
    """bytes of the binary ark entries of float32 row vectors [n, dim] (asv_io_pack_vec_ark; kaldi_io.vec_flt_ark_bytes uses it when
    the library is there - 0.05 ms instead of 0.4 ms of Python per batch of 327 vectors)."""
    import numpy as np
    L = lib()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = v.shape
    blob = "\n".join(keys).encode("latin1")
    cap = len(blob) - max(n - 1, 0) + n * (11 + 4 * dim)
    out = np.empty(cap, dtype=np.uint8)
    used = L.asv_io_pack_vec_ark(n, dim, blob, v.ctypes.data, dim, out.ctypes.data, cap)
    if used != cap:
        raise ValueError("asv_io_pack_vec_ark wrote %d of %d bytes (a key with a newline?)" % (used, cap))
    return out if as_array else out.tobytes()                    # (as_array: the uint8 array itself - a file's write() takes it, one copy fewer)


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


def scan_ark(fd, start, cap=8192, keys_cap=None):
    """Index of up to `cap` plain float32 entries of the ark file behind descriptor `fd` from byte `start` (asv_io_scan_ark):
    (keys, payload offsets int64, rows int32, cols int32, next offset, stopped) - stopped: 0 end of file, 1 cap reached, 2 an entry
    of another kind at `next`, 3 malformed entry at `next`; a read error raises OSError."""
    import numpy as np
    L = lib()
    keys_cap = int(keys_cap or cap * 48)
    off = np.empty(cap, dtype=np.int64)
    rows = np.empty(cap, dtype=np.int32)
    cols = np.empty(cap, dtype=np.int32)
    keys = C.create_string_buffer(keys_cap)
    nxt, stopped = C.c_int64(0), C.c_int32(0)
    n = int(L.asv_io_scan_ark(int(fd), int(start), int(cap), off.ctypes.data_as(C.POINTER(C.c_int64)), rows.ctypes.data_as(C.POINTER(C.c_int32)),
                              cols.ctypes.data_as(C.POINTER(C.c_int32)), keys, keys_cap, C.byref(nxt), C.byref(stopped)))
    if stopped.value == 4:
        err = L.asv_io_last_errno()
        raise OSError(err, os.strerror(err))
    names = [k.decode("latin1") for k in keys.raw.split(b"\n", n)[:n]]
    return names, off[:n], rows[:n], cols[:n], int(nxt.value), int(stopped.value)
