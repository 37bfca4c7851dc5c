# -*- coding:utf-8 -*-
"""Kaldi table I/O (ark / scp / pipes / gzip) for the extraction and scoring path.

Wire-format contract = what the reference reads and writes through its vendored kaldi-io
(/root/reference/pytorch/libs/support/kaldi_io.py): the `open_or_fd` rspecifier rules
(43-73), `read_key` (148-163), float vectors (329-399) and float / compressed matrices
(449-608).  This is an independent implementation with the same function names and
argument meaning, plus bulk helpers (`read_mat_ark_batched`) that the per-utterance loop of
the reference does not have.

Binary layouts (little endian):
    vector : key SP \\0 B  'FV '|'DV '  \\4 <int32 dim>            <dim   x f32|f64>
    matrix : key SP \\0 B  'FM '|'DM '  \\4 <int32 rows> \\4 <int32 cols>  <rows*cols x f32|f64> (row major)
    matrix : key SP \\0 B  'CM '  <f32 min><f32 range><int32 rows><int32 cols>
                            cols x 4 x uint16 percentiles, then cols*rows uint8 column-major
"""

import gzip
import io
import os
import re
import struct
import subprocess
import sys
import threading

import numpy as np


class UnsupportedDataType(Exception):
    pass


class UnknownVectorHeader(Exception):
    pass


class UnknownMatrixHeader(Exception):
    pass


class BadSampleSize(Exception):
    pass


class BadInputFormat(Exception):
    pass


class SubprocessFailed(Exception):
    pass


_SPECIFIER = re.compile(r"^(ark|scp)(,scp|,b|,t|,n?f|,n?p|,b?o|,n?s|,n?cs)*:")
_OFFSET = re.compile(r":[0-9]+$")


def _split_specifier(name):
    """'ark,t:foo.ark:123' -> ('ark', 'foo.ark', 123)."""
    prefix = None
    if _SPECIFIER.search(name):
        prefix, name = name.split(":", 1)
        prefix = prefix.split(",")[0]
    offset = None
    if _OFFSET.search(name):
        name, off = name.rsplit(":", 1)
        offset = int(off)
    return prefix, name, offset


class _PipeWriter(io.RawIOBase):
    """Write end of a `| cmd` pipe.  close() closes the child's stdin, WAITS for the child and raises
    SubprocessFailed on a non-zero exit status: with the standard wspecifier `ark:| copy-vector ark:- ark,scp:...`
    (pipeline/extract_xvectors_for_pytorch.sh:120) the ark/scp pair is complete on disk when close() returns, and a
    failing child is an exception in the caller instead of a lost status."""

    mode = "wb"

    def __init__(self, proc, cmd):
        io.RawIOBase.__init__(self)
        self._proc, self._cmd, self._stream = proc, cmd, proc.stdin

    def write(self, data):
        return self._stream.write(data)

    def flush(self):
        if not self._stream.closed:
            self._stream.flush()

    def fileno(self):
        return self._stream.fileno()

    def writable(self):
        return True

    def close(self):
        if self.closed:
            return
        # Whatever the flush does (BrokenPipeError once the child has died), the child's stdin is closed BEFORE waiting for
        # it - a live child would otherwise never see end-of-file and wait() would hang - and a failed child is reported as
        # SubprocessFailed with its exit status, the write error chained behind it.
        err = None
        try:
            io.RawIOBase.close(self)                  # flushes, marks closed
        except OSError as e:
            err = e
        finally:
            try:
                self._stream.close()
            except OSError as e:
                err = err or e
            finally:
                ret = self._proc.wait()
        if ret != 0:
            raise SubprocessFailed("cmd %s returned %d !" % (self._cmd, ret)) from err
        if err is not None:
            raise err


def popen(cmd, mode="rb"):
    """Shell pipe as a file object.  As in the reference (kaldi_io.py:76-113) a NON-daemon watcher thread waits for
    the child, so the interpreter does not exit (and run.pl does not return) while e.g. copy-vector is still writing,
    and a non-zero exit raises SubprocessFailed in that thread; the write end additionally waits and raises in
    close() (see _PipeWriter)."""
    if not isinstance(cmd, str):
        raise TypeError("invalid cmd type (%s, expected string)" % type(cmd))
    if mode not in ("r", "w", "rb", "wb"):
        raise ValueError("invalid mode %s" % mode)
    reading = mode[0] == "r"
    proc = subprocess.Popen(cmd, shell=True, stderr=sys.stderr,
                            stdout=subprocess.PIPE if reading else None,
                            stdin=None if reading else subprocess.PIPE)

    def _watch():
        ret = proc.wait()
        if ret > 0 and reading:                       # the write end reports through close()
            raise SubprocessFailed("cmd %s returned %d !" % (cmd, ret))

    threading.Thread(target=_watch, daemon=False).start()
    if reading:
        return io.TextIOWrapper(proc.stdout) if "b" not in mode else proc.stdout
    w = _PipeWriter(proc, cmd)
    return io.TextIOWrapper(w) if "b" not in mode else w


def open_or_fd(file, mode="rb"):
    """Opens a file / gzip / 'cmd |' / '| cmd' pipe, or passes an open descriptor through.
    A trailing ':offset' seeks; a leading 'ark:' / 'scp:' specifier is stripped."""
    if not isinstance(file, str):
        return file
    _, name, offset = _split_specifier(file)
    if name.endswith("|"):
        fd = popen(name[:-1], "rb")
    elif name.startswith("|"):
        fd = popen(name[1:], "wb")
    elif name.rsplit(".", 1)[-1] == "gz":
        fd = gzip.open(name, mode)
    else:
        fd = open(name, mode)
    if offset is not None:
        fd.seek(offset)
    return fd


def open_with_prefix(file, mode="rb"):
    """Like open_or_fd but insists on an explicit 'ark:' / 'scp:' prefix and returns it."""
    if not isinstance(file, str):
        raise TypeError("{} is not a file.".format(file))
    prefix, _, _ = _split_specifier(file)
    if prefix not in ("ark", "scp"):
        raise TypeError("Specifier of {} is None, please specify this with scp or ark.".format(file))
    return prefix, open_or_fd(file, mode)


def _read_exact(fd, n):
    buf = fd.read(n)
    if len(buf) != n:
        raise BadInputFormat("unexpected end of stream (wanted %d bytes, got %d)" % (n, len(buf)))
    return buf


def read_key(fd):
    """Next utterance key of an ark stream, or None at end of file."""
    mode = getattr(fd, "mode", "rb")                    # (gzip.GzipFile.mode is an int: always binary - the reference's `'b' in fd.mode` raises TypeError on a .gz ark)
    assert not isinstance(mode, str) or "b" in mode, "Error: 'fd' was opened in text mode (in python3 use sys.stdin.buffer)"
    chars = []
    while True:
        ch = fd.read(1)
        if ch == b"" or ch == b" ":
            break
        chars.append(ch)
    key = b"".join(chars).decode("latin1").strip()
    if key == "":
        return None
    assert re.match(r"^\S+$", key) is not None
    return key


# ------------------------------------------------------------------------------- vectors

def _read_vec_flt_binary(fd):
    header = _read_exact(fd, 3).decode()
    if header == "FV ":
        dtype, size = np.float32, 4
    elif header == "DV ":
        dtype, size = np.float64, 8
    else:
        raise UnknownVectorHeader("The header contained '%s'" % header)
    assert _read_exact(fd, 1) == b"\4"
    dim = struct.unpack("<i", _read_exact(fd, 4))[0]
    if dim == 0:
        return np.array([], dtype="float32")
    return np.frombuffer(_read_exact(fd, dim * size), dtype=dtype)


def read_vec_flt(file_or_fd):
    """One float vector, binary or text ('[ 1 2 3 ]')."""
    fd = open_or_fd(file_or_fd)
    try:
        flag = fd.read(2)
        if flag == b"\0B":
            return _read_vec_flt_binary(fd)
        toks = (flag + fd.readline()).decode().strip().split()
        toks = [t for t in toks if t not in ("[", "]")]
        return np.array(toks, dtype=float)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_flt_ark(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        key = read_key(fd)
        while key:
            yield key, read_vec_flt(fd)
            key = read_key(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_flt_scp(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            key, rxfile = line.decode().strip().split(" ", 1)
            yield key, read_vec_flt(rxfile)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_flt_auto(file):
    """'scp:foo.scp' or 'ark:foo.ark' (the reference's 2020-05-31 addition, kaldi_io.py:273-287)."""
    prefix, fd = open_with_prefix(file)
    reader = read_vec_flt_scp if prefix == "scp" else read_vec_flt_ark
    for key, vec in reader(fd):
        yield key, vec


read_vec = read_vec_flt_auto     # score/pyplda calls kaldi_io.read_vec (shim 5)


def write_vec_flt(file_or_fd, v, key=""):
    """Binary float vector (32 or 64 bit); with `key` it is one ark entry."""
    assert isinstance(v, np.ndarray)
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if v.dtype == np.float32:
            tag = b"FV "
        elif v.dtype == np.float64:
            tag = b"DV "
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % v.dtype)
        head = (key + " ").encode("latin1") if key != "" else b""
        fd.write(head + b"\0B" + tag + b"\4" + struct.pack("<I", v.shape[0]) + v.tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_vec_int(file_or_fd):
    """One int32 vector (binary: '\\0B' \\4 dim then (\\4 int32)*dim; or text)."""
    fd = open_or_fd(file_or_fd)
    try:
        flag = fd.read(2)
        if flag == b"\0B":
            assert _read_exact(fd, 1) == b"\4"
            dim = struct.unpack("<i", _read_exact(fd, 4))[0]
            if dim == 0:
                return np.array([], dtype="int32")
            rec = np.frombuffer(_read_exact(fd, dim * 5), dtype=[("size", "int8"), ("value", "int32")], count=dim)
            assert rec[0]["size"] == 4
            return rec[:]["value"]
        toks = (flag + fd.readline()).decode().strip().split()
        toks = [t for t in toks if t not in ("[", "]")]
        return np.array(toks, dtype=int)
    finally:
        if fd is not file_or_fd:
            fd.close()


# ------------------------------------------------------------------------------ matrices

def _read_compressed_mat(fd, fmt, chunk=None):
    """Kaldi CompressedMatrix: 'CM ' (per-column 4-point piecewise-linear uint8 quantiser - the one format the reference's
    reader decodes, kaldi_io.py:527-569), and the two header-only formats of Kaldi's compressed-matrix.cc, which the reference
    refuses: 'CM2' (kTwoByte: row-major uint16, value = min + range * u / 65535) and 'CM3' (kOneByte: row-major uint8,
    value = min + range * u / 255).  All three share the 16-byte GlobalHeader behind the token."""
    if fmt in ("CM2", "CM3"):
        assert _read_exact(fd, 1) == b" "                 # the space that ends the 3-letter token
        gmin, grange, rows, cols = struct.unpack("<ffii", _read_exact(fd, 16))
        width, dt, top = (2, np.uint16, 65535.0) if fmt == "CM2" else (1, np.uint8, 255.0)
        q = np.frombuffer(_read_exact(fd, rows * cols * width), dtype=dt).reshape(rows, cols).astype(np.float32)
        mat = np.float32(gmin) + q * np.float32(np.float64(np.float32(grange)) * (1.0 / top))    # compressed-matrix.cc: float increment = range * (1.0 / 65535.0)
        if chunk is not None:
            mat = mat[int(chunk[0]):int(chunk[1]) + 1]
        return mat
    if fmt != "CM ":
        raise UnknownMatrixHeader("The header contained '%s'" % fmt)
    gmin, grange, rows, cols = struct.unpack("<ffii", _read_exact(fd, 16))
    pct = np.frombuffer(_read_exact(fd, cols * 8), dtype=np.uint16).reshape(cols, 4).astype(np.float32)
    pct = pct * np.float32(grange) * np.float32(1.52590218966964e-05) + np.float32(gmin)
    data = np.frombuffer(_read_exact(fd, cols * rows), dtype=np.uint8).reshape(cols, rows).astype(np.float32)
    p0, p25, p75, p100 = (pct[:, i:i + 1] for i in range(4))
    lo = p0 + (p25 - p0) / np.float32(64.0) * data
    mid = p25 + (p75 - p25) / np.float32(128.0) * (data - 64)
    hi = p75 + (p100 - p75) / np.float32(63.0) * (data - 192)
    mat = np.where(data <= 64, lo, np.where(data > 192, hi, mid)).astype(np.float32).T
    if chunk is not None:
        mat = mat[int(chunk[0]):int(chunk[1]) + 1]
    return mat


def _read_mat_binary(fd, chunk=None):
    header = _read_exact(fd, 3).decode()
    if header.startswith("CM"):
        return _read_compressed_mat(fd, header, chunk=chunk)
    if header == "FM ":
        dtype, size = np.float32, 4
    elif header == "DM ":
        dtype, size = np.float64, 8
    else:
        raise UnknownMatrixHeader("The header contained '%s'" % header)
    s1, rows, s2, cols = struct.unpack("<bibi", _read_exact(fd, 10))
    first, last = 0, rows - 1
    if chunk is not None:
        first, last = int(chunk[0]), int(chunk[1])
    if first > 0:
        if hasattr(fd, "seekable") and fd.seekable():
            fd.seek(first * cols * size, 1)
        else:
            _read_exact(fd, first * cols * size)
    n = last - first + 1
    buf = _read_exact(fd, n * cols * size)
    return np.frombuffer(buf, dtype=dtype).reshape(n, cols)


def _read_mat_ascii(fd, chunk=None):
    rows, count = [], 0
    first, last = (int(chunk[0]), int(chunk[1])) if chunk is not None else (0, -1)
    while True:
        line = fd.readline().decode()
        if len(line) == 0:
            raise BadInputFormat("end of stream inside a text matrix")
        toks = line.strip().split()
        if not toks:
            continue
        closing = toks[-1] == "]"
        if closing:
            toks = toks[:-1]
        if toks and count >= first and (last < 0 or count <= last):
            rows.append(np.array(toks, dtype="float32"))
        count += 1 if toks else 0
        if closing or (last >= 0 and count > last):
            return np.vstack(rows)


def read_mat(file_or_fd, chunk=None):
    """One float matrix [rows, cols]; `chunk=[first, last]` reads a row range (reference
    extension used by the egs readers)."""
    fd = open_or_fd(file_or_fd)
    try:
        flag = _read_exact(fd, 2)
        if flag == b"\0B":
            return _read_mat_binary(fd, chunk=chunk)
        assert flag == b" [", flag
        return _read_mat_ascii(fd, chunk=chunk)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_mat_ark(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        key = read_key(fd)
        while key:
            yield key, read_mat(fd)
            key = read_key(fd)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_mat_scp(file_or_fd):
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            key, rxfile = line.decode().strip().split(" ", 1)
            yield key, read_mat(rxfile)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_mat_ark_batched(file_or_fd, max_frames=65536, max_utts=1024):
    """Bulk reader for the batched extractor: yields (keys, feats [sum T, D] float32,
    offsets int32 [n+1]) groups of at most `max_frames` frames / `max_utts` utterances -
    the packed ragged layout asv_net_extract() consumes (include/asv_amd.h)."""
    keys, mats, frames = [], [], 0
    for key, mat in read_mat_ark(file_or_fd):
        if mats and (frames + mat.shape[0] > max_frames or len(mats) >= max_utts):
            yield _pack_group(keys, mats)
            keys, mats, frames = [], [], 0
        keys.append(key)
        mats.append(mat)
        frames += mat.shape[0]
    if mats:
        yield _pack_group(keys, mats)


class PackedArkReader(object):
    """Sequential reader of a float-matrix ark stream into caller-owned packed buffers - the feed of the batched
    extractor (pinned host memory the H2D copy starts from).  Keys and headers are parsed out of small block reads (8 KiB);
    the bulk of every uncompressed float32 payload lies past the block and is `readinto` straight into the destination
    rows (one copy, page cache -> pinned buffer): no per-utterance Python objects besides the key.  float64, compressed ('CM ') and text matrices
    take the generic decoders above and are converted on the way in.

        rd = PackedArkReader(fd)
        dim = rd.peek_dim()                                   # feature dimension of the first matrix (None: empty stream)
        keys, offsets, n = rd.read_group(feats, max_utts)     # fills feats[:n]; keys == [] at end of stream
    """

    def __init__(self, fd, block=1 << 13):
        self.fd = fd
        self.block = block
        self.buf = b""
        self.pos = 0
        self.eof = False
        self._pending = None                                  # (key, rows, cols, kind) parsed but not consumed yet
        self.row_pad = 0                                      # rows the consumer adds per utterance (the device layout's gap rows): a group is
                                                              # cut so that frames + row_pad * (utterances + 1) fits the buffer's row count too

    # -- byte supply ---------------------------------------------------------------------------------
    def _avail(self):
        return len(self.buf) - self.pos

    def _fill(self, need):
        """Makes at least `need` bytes available at self.pos (fewer only at end of stream)."""
        while self._avail() < need and not self.eof:
            chunk = self.fd.read(max(self.block, need - self._avail()))
            if not chunk:
                self.eof = True
                break
            self.buf = self.buf[self.pos:] + chunk if self._avail() else chunk
            self.pos = 0
        return self._avail() >= need

    def read(self, n):                                        # file-like face for the generic decoders
        self._fill(n)
        out = self.buf[self.pos:self.pos + n]
        self.pos += len(out)
        return out

    def readline(self):
        while True:
            k = self.buf.find(b"\n", self.pos)
            if k >= 0 or self.eof:
                break
            self._fill(self._avail() + 1)
        end = len(self.buf) if k < 0 else k + 1
        out = self.buf[self.pos:end]
        self.pos = end
        return out

    # -- records --------------------------------------------------------------------------------------
    def _next_header(self):
        if self._pending is not None:
            return self._pending
        while True:                                           # key: bytes up to the first space
            k = self.buf.find(b" ", self.pos)
            if k >= 0 or self.eof:
                break
            self._fill(self._avail() + 1)
        if k < 0:
            if self.buf[self.pos:].strip():
                raise BadInputFormat("trailing bytes without a key terminator at the end of the ark stream")
            return None
        key = self.buf[self.pos:k].decode("latin1").strip()
        self.pos = k + 1
        if key == "":
            return None
        if not self._fill(2):
            raise BadInputFormat("ark entry '%s' ends after its key" % key)
        flag = self.buf[self.pos:self.pos + 2]
        if flag != b"\0B":                                     # text matrix: leave the stream at the ' [' for read_mat
            self._pending = (key, -1, -1, "generic")
            return self._pending
        if not self._fill(5):
            raise BadInputFormat("ark entry '%s': truncated header" % key)
        tag = self.buf[self.pos + 2:self.pos + 5]
        if tag in (b"FM ", b"DM "):
            if not self._fill(15):
                raise BadInputFormat("ark entry '%s': truncated header" % key)
            s1, rows, s2, cols = struct.unpack_from("<bibi", self.buf, self.pos + 5)
            self.pos += 15
            self._pending = (key, rows, cols, "f4" if tag == b"FM " else "f8")
        else:
            self._pending = (key, -1, -1, "generic")          # 'CM ' and friends: decoded by read_mat
        return self._pending

    def peek_dim(self):
        h = self._next_header()
        if h is None:
            return None
        if isinstance(h[3], np.ndarray):                      # already decoded by an earlier peek
            return h[2]
        if h[3] == "generic":                                 # decode it now, keep the matrix for read_group
            m = np.ascontiguousarray(read_mat(self), dtype=np.float32)
            self._pending = (h[0], m.shape[0], m.shape[1], m)
            return m.shape[1]
        return h[2]

    def _payload_into(self, dst, nbytes):
        """nbytes of payload -> the writable byte view `dst`."""
        have = min(self._avail(), nbytes)
        if have:
            dst[:have] = np.frombuffer(self.buf, dtype=np.uint8, count=have, offset=self.pos)
            self.pos += have
        got = have
        while got < nbytes:                                   # past the block: straight from the stream
            if hasattr(self.fd, "readinto"):
                k = self.fd.readinto(dst[got:nbytes])
            else:
                chunk = self.fd.read(nbytes - got)
                k = len(chunk)
                dst[got:got + k] = np.frombuffer(chunk, dtype=np.uint8)
            if not k:
                raise BadInputFormat("unexpected end of stream inside a matrix (wanted %d more bytes)" % (nbytes - got))
            got += k

    def read_group(self, feats, max_utts=1024):
        """Fills feats [capacity, D] float32 (C-contiguous) with the next utterances that fit; returns
        (keys, offsets int32 [n+1], frames).  An utterance longer than the whole buffer is returned alone as
        (keys=[key], offsets, its own array) - the caller sees `frames > capacity` and uses the third value as the data."""
        cap, dim = feats.shape
        flat = feats.reshape(-1).view(np.uint8)
        keys, offs, used = [], [0], 0
        while len(keys) < max_utts:
            h = self._next_header()
            if h is None:
                break
            key, rows, cols, kind = h
            # `kind` is a tag string, or the decoded matrix itself once peek_dim() / an earlier call has decoded a
            # text or compressed entry that did not fit its batch (ndarray == str is an array on numpy >= 2)
            if isinstance(kind, str) and kind == "generic":
                m = np.ascontiguousarray(read_mat(self), dtype=np.float32)
                self._pending = h = (key, m.shape[0], m.shape[1], m)
                key, rows, cols, kind = h
            if cols != dim:
                raise BadInputFormat("ark entry '%s' has %d columns, the stream started with %d" % (key, cols, dim))
            if rows > cap:
                if keys:
                    break                                     # flush what we have first
                self._pending = None
                if isinstance(kind, np.ndarray):
                    big = kind
                else:
                    big = np.empty((rows, cols), dtype=np.float32 if kind == "f4" else np.float64)
                    self._payload_into(big.reshape(-1).view(np.uint8), big.nbytes)
                    big = big.astype(np.float32, copy=False)
                return [key], np.array([0, rows], dtype=np.int32), big
            if used + rows > cap or (keys and used + rows + self.row_pad * (len(keys) + 2) > cap):
                break
            self._pending = None
            if isinstance(kind, np.ndarray):
                feats[used:used + rows] = kind
            elif kind == "f4":
                self._payload_into(flat[used * dim * 4:(used + rows) * dim * 4], rows * dim * 4)
            else:
                tmp = np.empty((rows, cols), dtype=np.float64)
                self._payload_into(tmp.reshape(-1).view(np.uint8), tmp.nbytes)
                feats[used:used + rows] = tmp
            keys.append(key)
            used += rows
            offs.append(used)
        return keys, np.asarray(offs, dtype=np.int32), used


class IndexedArkReader(object):
    """PackedArkReader's interface (peek_dim / read_group) for an ark FILE of plain float32 matrices, on libasv_io.so: the entries are
    indexed by a native header scan (asv_io_scan_ark: small positioned reads, payloads skipped) and the payloads of a whole batch are
    read into the caller's packed buffer by ONE call (asv_io_pread_batch: native threads, no per-utterance Python) - 200 k
    utterances/s where the sequential reader parses and copies 56 k (build host, [200, 80] matrices from the page cache).
    `IndexedArkReader.open(f)` returns None when this does not apply - a pipe / stdin, the library not built, an archive that does not
    start with a float32 entry; entries of another kind later in the file (text, float64, compressed) are handed to a
    PackedArkReader that continues at that byte."""

    CHUNK = 8192

    @classmethod
    def open(cls, f, threads=4):
        import stat
        from . import native_io
        try:
            if (native_io.lib() is None or not isinstance(f, (io.BufferedReader, io.FileIO)) or not f.seekable()      # (a GzipFile has a fileno too)
                    or not stat.S_ISREG(os.fstat(f.fileno()).st_mode)):
                return None
            rd = cls(f, threads)
        except (AttributeError, OSError, ValueError):
            return None
        if rd._n == 0 and rd._stopped != 0:                    # the first entry is of another kind (or malformed): the generic reader's business
            return None
        return rd

    def __init__(self, f, threads=4):
        self.f, self.fd, self.threads = f, f.fileno(), int(threads)
        self._next = f.tell()
        self._keys, self._off, self._rows, self._cols = [], np.empty(0, np.int64), np.empty(0, np.int32), np.empty(0, np.int32)
        self._at = self._n = 0
        self._stopped = 1
        self._tail = None                                    # PackedArkReader for the rest of the file once another kind of entry shows up
        self.row_pad = 0                                     # as PackedArkReader.row_pad
        self._more()

    def _more(self):
        """Indexes the next CHUNK entries once the current ones are used up; False at the end of the indexed part."""
        if self._at < self._n:
            return True
        if self._stopped != 1:
            return False
        from . import native_io
        self._keys, self._off, self._rows, self._cols, self._next, self._stopped = native_io.scan_ark(self.fd, self._next, self.CHUNK)
        self._at, self._n = 0, len(self._keys)
        return self._n > 0

    def _rest(self):
        """The generic reader behind the indexed part (None at a clean end of file)."""
        if self._tail is None and self._stopped in (2, 3):
            self.f.seek(self._next)
            self._tail = PackedArkReader(self.f)
            self._tail.row_pad = self.row_pad
        return self._tail

    def peek_dim(self):
        if self._more():
            return int(self._cols[self._at])
        t = self._rest()
        return t.peek_dim() if t is not None else None

    def read_group(self, feats, max_utts=1024):
        cap, dim = feats.shape
        keys, offs, used = [], [0], 0
        src, nbytes, dst = [], [], []
        while len(keys) < max_utts and self._more():
            a = self._at
            m = min(self._n - a, max_utts - len(keys))
            bad = np.flatnonzero(self._cols[a:a + m] != dim)
            if bad.size:
                if bad[0] == 0:
                    if keys:
                        break                                 # flush what we have first
                    raise BadInputFormat("ark entry '%s' has %d columns, the stream started with %d" % (self._keys[a], int(self._cols[a]), dim))
                m = int(bad[0])                               # the entries in front of it now, the error on the next call
            rows = self._rows[a:a + m].astype(np.int64)
            ends = np.cumsum(rows)
            take = int(np.searchsorted(ends, cap - used, side="right"))         # how many of them still fit
            if self.row_pad and take:
                # ... with the consumer's gap rows counted in: frames + row_pad * (utterances + 1) <= cap (at least one utterance per group)
                padded = ends + self.row_pad * (np.arange(1, len(ends) + 1) + len(keys) + 1)
                take = min(take, max(int(np.searchsorted(padded, cap - used, side="right")), 0 if keys else 1))
            if take == 0:
                if rows[0] > cap and not keys:                # an utterance longer than the whole buffer: alone, in an array of its own
                    self._at += 1
                    big = np.empty((int(rows[0]), dim), dtype=np.float32)
                    self._read([int(self._off[a])], [big.nbytes], big.ctypes.data, [0], self._keys[a:a + 1])
                    return [self._keys[a]], np.array([0, int(rows[0])], dtype=np.int32), big
                break
            src.append(self._off[a:a + take])
            nbytes.append(rows[:take] * (dim * 4))
            dst.append((used + ends[:take] - rows[:take]) * (dim * 4))
            keys.extend(self._keys[a:a + take])
            offs.extend((used + ends[:take]).tolist())
            used += int(ends[take - 1])
            self._at += take
            if take < m:
                break                                         # the next one does not fit: flush
        if keys:
            self._read(np.concatenate(src), np.concatenate(nbytes), feats.ctypes.data, np.concatenate(dst), keys)
            return keys, np.asarray(offs, dtype=np.int32), used
        if self._at < self._n:                                # (max_utts == 0, or an entry of the wrong width is next: raised above on the next call)
            return keys, np.asarray(offs, dtype=np.int32), used
        t = self._rest()
        if t is not None:
            return t.read_group(feats, max_utts)
        return keys, np.asarray(offs, dtype=np.int32), used

    def _read(self, src, nbytes, base, dst, keys):
        from . import native_io
        try:
            native_io.pread_batch([self.fd] * len(src), src, nbytes, base, dst, threads=self.threads)
        except OSError as e:
            k = e.args[2] if len(e.args) > 2 else 0
            raise BadInputFormat("ark entry '%s': %s" % (keys[k], e.args[1]))


def vec_flt_ark_bytes(keys, vectors, as_buffer=False):
    """One bytes object holding the binary ark entries `key SP \\0B FV \\4 dim data` of float32 row vectors [n, dim]
    (write_vec_flt's format, assembled per batch instead of written per key).  as_buffer: any bytes-like object will do (a file's
    write() takes it): the native packer's own array is handed over without the copy into a bytes object."""
    vectors = np.ascontiguousarray(vectors, dtype=np.float32)
    if len(keys) >= 16:
        from . import native_io
        L = native_io.lib()
        if L is not None and L.asv_io_version() >= 2 and not any("\n" in k for k in keys):
            return native_io.pack_vec_ark(keys, vectors, as_array=as_buffer)
    head = b"\0BFV \4" + struct.pack("<I", vectors.shape[1])
    rows = vectors.view(np.uint8).reshape(vectors.shape[0], -1)
    return b"".join([(k + " ").encode("latin1") + head + rows[i].tobytes() for i, k in enumerate(keys)])


def _pack_group(keys, mats):
    offsets = np.zeros(len(mats) + 1, dtype=np.int32)
    np.cumsum([m.shape[0] for m in mats], out=offsets[1:])
    feats = np.ascontiguousarray(np.concatenate(mats, axis=0), dtype=np.float32)
    return keys, feats, offsets


def write_mat(file_or_fd, m, key=""):
    assert isinstance(m, np.ndarray)
    assert len(m.shape) == 2, "'m' has to be 2d matrix!"
    fd = open_or_fd(file_or_fd, mode="wb")
    try:
        if m.dtype == np.float32:
            tag = b"FM "
        elif m.dtype == np.float64:
            tag = b"DM "
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % m.dtype)
        head = (key + " ").encode("latin1") if key != "" else b""
        fd.write(head + b"\0B" + tag + b"\4" + struct.pack("<I", m.shape[0]) + b"\4" + struct.pack("<I", m.shape[1]))
        fd.write(np.ascontiguousarray(m).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_vec_flt_ark_scp(ark_path, scp_path, items):
    """Writes (key, vector) pairs to `ark_path` and the matching 'key path:offset' lines to
    `scp_path` - what `copy-vector ark:- ark,scp:...` does in
    pipeline/extract_xvectors_for_pytorch.sh:120."""
    with open(ark_path, "wb") as ark, open(scp_path, "w") as scp:
        for key, vec in items:
            ark.write((key + " ").encode("latin1"))
            scp.write("%s %s:%d\n" % (key, os.path.abspath(ark_path), ark.tell()))
            write_vec_flt(ark, np.ascontiguousarray(vec), key="")
