# -*- coding:utf-8 -*-
"""Offline embedding extractor - same command line as the reference script
(/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:17-45), so
`pipeline/extract_xvectors_for_pytorch.sh` can call it unchanged:

    extract_embeddings.py [--nnet-config CFG | --model-blueprint PY --model-creation CTOR]
                          [--use-gpu true] [--gpu-id N] model-path feats-rspecifier vectors-wspecifier

Differences are internal: features are read in bulk and embedded in packed ragged batches by
libasv_amd.so instead of one `model.extract_embedding()` call, one H2D copy, ~15 launches and one
blocking D2H per utterance (reference lines 73-83).  Output order and bytes per entry are the same
(`key SP \\0B FV \\4 dim data`).  Any error prints a traceback and exits 1 (reference 85-88) - the
calling shell script greps the log for "Error".

Extra options: --batch-frames / --batch-utts bound a batch; --verbose true restores the reference's
per-utterance "Process utterance for key ..." line (a measurable cost at >100k utterances/s).

Multi-GPU (--sharded true, launched as `python -m torch.distributed.run --nproc-per-node N ... extract_embeddings.py ...`, one
process per GPU): replaces the reference's `nj` jobs over a length-balanced split of feats.scp and the final
`cat xvector.JOB.scp` (pipeline/extract_xvectors_for_pytorch.sh:90-100,125-151; splitDataByLength.sh:44-80).  Every rank reads
the scp (feats-rspecifier must be `scp:...`; lengths from --utt2num-frames or from the matrix headers), takes its shard of
libs.amd.shard.balance_by_length, extracts it, ONE all-gather (RCCL over xGMI) collects the embeddings and rank 0 writes the
vectors in scp order.
"""

import argparse
import os
import re
import sys
import time
import traceback

sys.path.insert(0, "subtools/pytorch")
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

import numpy as np
import torch

import libs.support.kaldi_io as kaldi_io
import libs.support.utils as utils


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Extract embeddings form a piece of feats.scp or pipeline")
    parser.add_argument("--nnet-config", type=str, default="", help="This config contains model_blueprint and model_creation.")
    parser.add_argument("--model-blueprint", type=str, default=None, help="A *.py which includes the instance of nnet in this training.")
    parser.add_argument("--model-creation", type=str, default=None, help="A command to create the model class, e.g. Xvector(40,2).")
    parser.add_argument("--use-gpu", type=str, default="true", choices=["true", "false"], help="The MI355X path needs a GPU; 'false' is rejected.")
    parser.add_argument("--gpu-id", type=str, default="", help="Specify a fixed gpu, or select gpu automatically.")
    parser.add_argument("--batch-frames", type=int, default=65536, help="Upper bound of frames per packed batch.")
    parser.add_argument("--batch-utts", type=int, default=1024, help="Upper bound of utterances per packed batch.")
    parser.add_argument("--max-chunk", type=int, default=0, help="Override the model's maxChunk (0 = the decorator's value).")
    parser.add_argument("--verbose", type=str, default="false", choices=["true", "false"])
    parser.add_argument("--sharded", type=str, default="false", choices=["true", "false"],
                        help="One process per GPU under torch.distributed.run: shard the scp by length, all-gather, rank 0 writes.")
    parser.add_argument("--utt2num-frames", type=str, default="", help="Kaldi utt2num_frames of the scp (sharded mode; default: read the matrix headers).")
    parser.add_argument("model_path", metavar="model-path", type=str, help="The model used to extract embeddings.")
    parser.add_argument("feats_rspecifier", metavar="feats-rspecifier", type=str, help="")
    parser.add_argument("vectors_wspecifier", metavar="vectors-wspecifier", type=str, help="")
    return parser.parse_args(argv)


def extract_stream(model, r, w, batch_frames, batch_utts, max_chunk, verbose=False, reader=None):
    """feature ark stream -> embedding ark stream, pipelined in three stages over two buffer sets:
      reader thread   next packed batch -> pinned host buffer   (kaldi_io.IndexedArkReader for ark files - native index + batched reads -,
                      kaldi_io.PackedArkReader for pipes: block reads, no per-utterance arrays; ScpGroupReader for `scp:` input)
      device          async H2D, asv_net_extract, async D2H into a pinned result buffer + the range status word   (libs.amd.pipeline.DeviceSets:
                      each buffer set has its own engine - same weights, own activation arena - on its own HIP stream)
      writer          previous batch's vectors -> one write() of the assembled ark bytes
    so reading batch i+1 and writing batch i-1 overlap the device work of batch i.  Output order = input order.
    A batch whose activations left the range of the default mode's operand split is re-run with bf16 halves before it is written
    (DeviceSets.finish; the reference's f32 forward has no such limit)."""
    import queue
    import threading
    from libs.amd.pipeline import DeviceSets
    method = type(model).extract_embedding
    if max_chunk is None:
        max_chunk = getattr(method, "max_chunk", 10000)
    if reader is None:
        # an ark FILE of plain float32 matrices: native header index + one batched positioned read per group (libasv_io.so: 240 k
        # utterances/s from the page cache against 58 k for the sequential parser); pipes, stdin, other matrix kinds: the sequential reader
        reader = kaldi_io.IndexedArkReader.open(r, threads=_reader_threads()) if os.environ.get("ASV_AMD_INDEXED_READER", "1") != "0" else None
        if reader is None:
            reader = kaldi_io.PackedArkReader(r)
    dim = reader.peek_dim()
    if dim is None:
        return 0
    # the device lays a batch out with kHalo = 4 zero rows in front of, between and behind the utterances and works in tiles of 64 /
    # 128 / 256 rows: a group is cut so that its ROWS fit --batch-frames (65 536 = 256 tiles of 256 rows: 321 utterances of 200 frames
    # fill whole rounds of tiles on the 256 CUs; 327 - the frames alone - were 2.04 rounds of 128-row tiles, i.e. three)
    reader.row_pad = 4
    sets = DeviceSets(model, batch_frames, batch_utts, dim, max_chunk, n_sets=3, results="host")
    free_sets, batches = queue.Queue(), queue.Queue()
    for k in range(sets.n_sets):
        free_sets.put(k)

    def produce():
        try:
            while True:
                k = free_sets.get()
                keys, offsets, frames = reader.read_group(sets.host_buffer(k), batch_utts)
                batches.put((k, keys, offsets, frames))
                if not keys:
                    return
        except BaseException as e:                    # surfaces in the consumer
            batches.put(e)

    t = threading.Thread(target=produce, daemon=True)
    t.start()
    n_done, in_flight = 0, None

    clock = time.perf_counter
    spent = {"reader": 0.0, "submit": 0.0, "device": 0.0, "write": 0.0}       # where the consumer thread's time goes (ASV_AMD_REPORT_TIMING)

    def finish(item):
        k, keys = item
        a = clock()
        vectors = sets.finish(k)                      # waits for the batch; range guard (re-run on the bf16-halves twin if flagged)
        b = clock()
        if verbose:
            for key in keys:
                print("Process utterance for key {0}".format(key))
        w.write(kaldi_io.vec_flt_ark_bytes(keys, vectors))
        free_sets.put(k)
        spent["device"] += b - a
        spent["write"] += clock() - b
        return len(keys)

    t0 = clock()
    while True:
        a = clock()
        item = batches.get()
        b = clock()
        spent["reader"] += b - a
        if isinstance(item, BaseException):
            raise item
        k, keys, offsets, frames = item
        if not keys:
            break
        sets.submit(k, offsets, frames)
        spent["submit"] += clock() - b
        if in_flight is not None:
            n_done += finish(in_flight)
        in_flight = (k, keys)
    if in_flight is not None:
        n_done += finish(in_flight)
    t.join()
    _report_loop("stream", n_done, clock() - t0, sets, spent)
    return n_done


def _reader_threads():
    """Native reader threads per process: ASV_AMD_READER_THREADS, else 8 for one or two ranks on the host and 4 from three ranks on
    (WORLD_SIZE / LOCAL_WORLD_SIZE), at most the cores this process may run on.  Positioned reads from the page cache scale to ~8 threads
    for one process (186 k utterances/s of 200 x 80 f32 on 8 threads against 172 k on 4 and 92 k on one, on the GPU box's 256-core host);
    with 4 - 8 ranks reading at once 4 threads per rank give the best aggregate and 8 lose 6 - 16 % (tools/bench_loaders.py,
    profiles/r6*_loaders*.txt)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    try:
        ranks = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    default = 8 if ranks <= 2 else 4
    return max(1, min(int(os.environ.get("ASV_AMD_READER_THREADS", str(default))), cores))


def _report_loop(path, n, seconds, sets, spent=None):
    """ASV_AMD_REPORT_TIMING=1: one line with the rate of the read -> device -> write loop alone (model load, engine compilation and
    process start excluded; tools/bench_pipeline.py and bench.py's supplementary.ark_to_ark read it)."""
    if os.environ.get("ASV_AMD_REPORT_TIMING", "0") not in ("0", "", "false"):
        print("Loop[{0}]: {1} utterances in {2:.4f} s = {3:.1f} utterances/s (reader -> device -> writer; range re-runs: {4})".format(
            path, n, seconds, n / max(seconds, 1e-9), sets.range_reruns))
        if spent:
            print("Loop[{0}] consumer thread: ".format(path) + ", ".join("{0} {1:.4f} s".format(k, v) for k, v in spent.items()))
        print("Loop[{0}] submit pieces: ".format(path) + ", ".join("{0} {1:.4f} s".format(k, v) for k, v in sets.submit_seconds.items()))


_SCP_PREFIX = re.compile(r"^scp(,[a-z_,]*)?:")


class ScpTable(object):
    """The entries of a Kaldi scp table as a read-only sequence of (key, rxfile) tuples over ONE native parse of the file's bytes
    (libasv_io.so asv_io_parse_scp): the tuples are made when asked for.  Every rank of a --sharded run walks the WHOLE table - it needs
    every length to balance the shards - and as a Python list of tuples + a per-entry parse in ScpBatchLoader.index_all that was
    0.12 - 0.18 s per 50 000 entries and rank: the part of the host work that did not scale with the ranks (tools/bench_loaders.py,
    profiles/r6h_loaders_before.txt).  `plain_index()` hands the loader the parsed 'path:offset' arrays."""

    def __init__(self, data, parsed):
        self._data, self._p, self._n = data, parsed, int(parsed["n"])

    def __len__(self):
        return self._n

    def _entry(self, i):
        p, d = self._p, self._data
        a, b = int(p["key_off"][i]), int(p["rx_off"][i])
        return d[a:a + int(p["key_len"][i])].decode("latin1"), d[b:b + int(p["rx_len"][i])].decode("latin1")

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._entry(k) for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._entry(i)

    def __iter__(self):
        return (self._entry(i) for i in range(self._n))

    def __eq__(self, other):                               # (a sequence of tuples: equal to the list the Python parse returns)
        try:
            return len(other) == self._n and all(x == tuple(y) for x, y in zip(self, other))
        except TypeError:
            return NotImplemented

    __hash__ = None

    def keys(self, a=0, b=None):
        """The keys of entries [a, b) (what rank 0 writes a segment's vectors under)."""
        p, d = self._p, self._data
        b = self._n if b is None else b
        return [d[int(o):int(o) + int(l)].decode("latin1") for o, l in zip(p["key_off"][a:b], p["key_len"][a:b])]

    def plain_index(self):
        """(path id per entry or -1, byte offset per entry, the distinct paths) of the entries of the plain form 'path:offset'."""
        return self._p["path_id"], self._p["offset"], self._p["paths"]


def read_scp(path):
    """The (key, rxfile) entries of a Kaldi scp rspecifier: 'scp:feats.scp', with options ('scp,p:', 'scp,s,cs:' - accepted and ignored,
    like the reference's read_mat_scp), 'scp:-' (stdin), 'scp:cmd |' (a pipe), .gz; the prefix is optional.  A ScpTable (one native
    parse) when libasv_io.so is there, else a list of tuples - the same sequence either way."""
    m = _SCP_PREFIX.match(path)
    name = path[m.end():] if m else path
    if name.strip() == "-":
        data = sys.stdin.buffer.read() if hasattr(sys.stdin, "buffer") else sys.stdin.read().encode("latin1")
    else:
        fd = kaldi_io.open_or_fd(name.strip(), "rb")
        try:
            data = fd.read()
        finally:
            fd.close()
    from libs.support import native_io
    parsed = native_io.parse_scp(data) if native_io.lib() is not None else None
    if parsed is not None:
        return ScpTable(data, parsed)
    return [tuple(line.strip().split(None, 1)) for line in data.decode("latin1").splitlines() if line.strip()]


_RANGE = re.compile(r"\[([0-9]*):?([0-9]*)(?:,([0-9]*):?([0-9]*))?\]$")


def split_range(rxfile):
    """Kaldi scp range specifier 'file.ark:123[r0:r1]' / '[r0:r1,c0:c1]' (both ends inclusive, either part may be empty) ->
    (rxfile without it, row slice, column slice).  The nj-job splits of the reference go through Kaldi's copy-feats, which
    accepts such entries."""
    m = _RANGE.search(rxfile)
    if not m:
        return rxfile, slice(None), slice(None)
    def sl(a, b):
        return slice(int(a) if a else None, int(b) + 1 if b else None)
    body = rxfile[:m.start()]
    return body, sl(m.group(1), m.group(2)), sl(m.group(3), m.group(4))


def read_matrix(rxfile):
    """float32 matrix behind an scp entry, range specifier applied."""
    body, rows, cols = split_range(rxfile)
    return np.ascontiguousarray(kaldi_io.read_mat(body)[rows, cols], dtype=np.float32)


def matrix_rows(rxfile):
    """Number of rows of the matrix behind an scp entry, from its header only (FM / DM / CM / CM2 / CM3 - the compressed
    formats share the GlobalHeader), or by decoding it (text); a range specifier is applied to the count."""
    import struct
    body, rows, _ = split_range(rxfile)
    fd = kaldi_io.open_or_fd(body, "rb")
    try:
        head = fd.read(2)
        if head != b"\0B":
            fd.close()
            n = int(kaldi_io.read_mat(body).shape[0])
        else:
            tag = fd.read(3)
            if tag in (b"FM ", b"DM "):
                n = struct.unpack("<bibi", fd.read(10))[1]
            elif tag == b"CM ":
                n = struct.unpack("<ffii", fd.read(16))[2]
            elif tag in (b"CM2", b"CM3"):
                fd.read(1)                                # the space behind the 3-letter token
                n = struct.unpack("<ffii", fd.read(16))[2]
            else:
                raise kaldi_io.UnknownMatrixHeader("The header contained '%s'" % tag)
    finally:
        fd.close()
    return len(range(*rows.indices(n)))


class PackedBatch(object):
    """The matrices of one batch as row slices of ONE buffer: `packed` [frames, D] float32, `offsets` int32 [n + 1]; `turn`: the slot of
    the loader's buffer rotation this batch consumed (always - the caller's pipeline rotates with it); `own_array`: the batch did not
    fit that buffer (one utterance longer than a whole batch) and lies in an array of its own.  A sequence of the [T_k, D] views, made
    when asked for (the device path takes `packed` and `offsets` as they are: 321 views per batch were 50 000 slicing calls per run
    that nobody looked at)."""

    def __init__(self, packed, offsets, turn, own_array=False):
        self.packed, self.offsets, self.turn, self.own_array = packed, offsets, turn, own_array

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[j] for j in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return self.packed[int(self.offsets[k]):int(self.offsets[k + 1])]

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class ScpBatchLoader(object):
    """Loads the utterances of a batch from scp entries into one packed buffer - the host side of the sharded path and of `scp:` input.

    Until round 4 a rank read its shard one `read_mat` at a time (open, parse, array) and concatenated the batch: 10 k utterances/s
    (0.65 GB/s) per rank against 56 k for the sequential stream reader and ~950 k the device extracts.  Here an uncompressed float32
    entry 'file.ark:offset' (what copy-feats / the reference's make_features write) costs one 15-byte header pread and one read of its
    payload straight into its rows of the batch buffer; everything else - float64, compressed, text matrices, range specifiers,
    pipes - goes through read_matrix and is copied in.  Two buffers alternate: libs.amd.shard.extract_sharded fetches batch k + 1
    while batch k is on the device.  The buffers are the loader's own arrays, or the CALLER's (`buffers`: e.g. the page-locked
    input buffers of libs.amd.pipeline.DeviceSets, so that the H2D copy starts from where the file was read to; `before_fill(turn)`
    is then called before buffer `turn` is overwritten - the caller waits for its copy out of it there).  `loader(i)` (one
    utterance) stays available: it is what extract_sharded takes as `load_utt`.
    The payload reads of a batch are ONE call into libasv_io.so (csrc/host_io.c: positioned reads on `threads` native threads,
    the GIL released for the whole call); without that library (not built) the same reads are issued from Python, one per
    utterance.  Measured on the build host (20 000 x [200, 80] float32 from the page cache; LABLOG.md row "a16, e (host side)"):
    203 k utterances/s = 13 GB/s on 4 native threads (1: 92 k, 8: 213 k); the Python reads: 66 k on one thread and LESS when
    split over Python threads (4: 43 k - a 64 KiB read is ~12 us, the Python around it ~3 us under the GIL, the hand-over
    between threads costs more than it buys); the round-3 path (read_mat per utterance + concatenate): 10 k.
    Descriptors: one per ark file, kept in an LRU of `max_open` (a feats.scp of an augmented / nj-split feature directory names
    more files than the soft RLIMIT_NOFILE allows open at once - ADVICE r4); the files of the batch being read are never evicted."""

    def __init__(self, entries, threads=4, buffers=None, before_fill=None, max_open=256):
        import collections
        import threading
        self.entries = entries
        self.threads = max(1, int(threads))
        self.max_open = max(1, int(max_open))
        self._fds = collections.OrderedDict()            # path -> descriptor, least recently used first
        self._heads = {}
        self._table = None                               # index_all(): (plain, file id, payload offset, rows, cols) arrays + the file list
        self._bufs = list(buffers) if buffers is not None else [None, None]
        self._own = buffers is None
        self._before_fill = before_fill
        self._turn = 0
        self._lock = threading.Lock()

    def __call__(self, i):
        return read_matrix(self.entries[i][1])

    def close(self):
        with self._lock:
            for fd in self._fds.values():
                os.close(fd)
            self._fds.clear()

    def _fd(self, path, keep=()):
        """Descriptor of `path` (opened on demand, most recently used last); beyond max_open the least recently used one that the
        current batch (`keep`) does not need is closed."""
        with self._lock:
            fd = self._fds.get(path)
            if fd is not None:
                self._fds.move_to_end(path)
                return fd
            if len(self._fds) >= self.max_open:
                for old in list(self._fds):
                    if len(self._fds) < self.max_open:
                        break
                    if old not in keep:
                        os.close(self._fds.pop(old))
            fd = self._fds[path] = os.open(path, os.O_RDONLY)
            return fd

    def _direct(self, i):
        """(path, payload offset, rows, cols) of entry i if it is a plain float32 'file:offset' entry, else None (parsed once)."""
        h = self._heads.get(i, 0)
        if h != 0:
            return h
        if self._table:
            plain, fid, pos, rows, cols, files = self._table
            return (files[fid[i]], int(pos[i]), int(rows[i]), int(cols[i])) if plain[i] else None
        import struct
        rxfile = self.entries[i][1]
        h = None
        if not (rxfile.endswith("]") or rxfile.endswith("|") or ":" not in rxfile):
            path, _, off = rxfile.rpartition(":")
            if off.isdigit() and (path in self._fds or os.path.isfile(path)):
                off = int(off)
                head = os.pread(self._fd(path), 15, off)
                if len(head) == 15 and head[:5] == b"\0BFM " and head[5] == 4 and head[10] == 4:
                    h = (path, off + 15, struct.unpack_from("<i", head, 6)[0], struct.unpack_from("<i", head, 11)[0])
        self._heads[i] = h
        return h

    def index_all(self, ranks=None):
        """`ranks` = (rank, world, gather) under --sharded with more than one rank: the header pass is SHARED - every rank reads the
        headers of one contiguous 1 / world of the entries and `gather(local uint8 array) -> [world, ...] array` (an all-gather)
        completes the table on every rank (round 6: the per-rank pass over the WHOLE table was what did not scale on one host - 0.10 s
        per 50 000 entries on every rank, 64 % of a rank's host work at 8 ranks, tools/bench_loaders.py).  Every rank must hold the same
        table; the ranks first compare (entries, plain candidates, a checksum of their offsets) and fall back to the whole pass, each on
        its own, if they differ; a failed header read on any rank sends all of them to the per-entry path together.

        Headers of ALL plain float32 'file:offset' entries in one native call (15 bytes each, libasv_io.so asv_io_pread_batch) and a
        vectorised parse: arrays (plain, file id, payload offset, rows, cols) that load_batch() indexes with a batch's entry numbers -
        per batch a handful of numpy operations instead of ~10 Python operations per utterance (50 000 utterances: 0.15 s of
        interpreter time on the reader thread, under the GIL the submitting thread needs too; the header pass itself 0.2 s -> 0.03 s).
        Without the native library, with more files than `max_open`, or if any header read fails, the per-entry path stays in charge
        (returns False)."""
        if self._table is not None:
            return self._table is not False
        self._table = False
        from libs.support import native_io
        sharing = ranks is not None and ranks[1] > 1
        if native_io.lib() is None:
            if sharing:
                ranks[2](np.zeros((1, 16), dtype=np.uint8))          # take part in the agreement (as "nothing to share"): nobody waits for this rank
            return False
        n = len(self.entries)
        if hasattr(self.entries, "plain_index"):
            # the table was parsed natively (ScpTable): path ids and offsets are arrays already - no per-entry Python
            pid, poff, paths = self.entries.plain_index()
            remap = np.full(len(paths) + 1, -1, dtype=np.int32)          # (slot -1 = "not plain": stays -1)
            files = []
            for k, path in enumerate(paths):
                if os.path.isfile(path):
                    remap[k] = len(files)
                    files.append(path)
            fid_all = remap[pid]                                          # pid = -1 indexes the last slot
            cand = np.flatnonzero(fid_all >= 0).astype(np.int64)
            fid, off = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int64)
            fid[cand], off[cand] = fid_all[cand], poff[cand]
        else:
            files, ids, cand, c_fid, c_off = [], {}, [], [], []
            for i, (_, rx) in enumerate(self.entries):
                path, sep, o = rx.rpartition(":")
                if not sep or not o.isdigit():                # (range specifiers end in ']', pipes in '|': neither is all digits)
                    continue
                j = ids.get(path)
                if j is None:
                    j = ids[path] = len(files) if os.path.isfile(path) else -1
                    if j >= 0:
                        files.append(path)
                if j >= 0:
                    cand.append(i)
                    c_fid.append(j)
                    c_off.append(int(o))
            fid, off = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int64)
            fid[cand], off[cand] = c_fid, c_off
        if len(cand) == 0 or len(files) > self.max_open:
            if sharing:
                ranks[2](np.zeros((1, 16), dtype=np.uint8))          # (as above)
            return False
        cand = np.asarray(cand, dtype=np.int64)
        keep = set(files)
        lut = np.asarray([self._fd(path, keep=keep) for path in files], dtype=np.int32)
        nc = len(cand)
        shared = False
        if sharing:
            rank, world, gather = ranks
            mine = np.zeros((1, 16), dtype=np.uint8)
            mine.view("<i8")[0, 0], mine.view("<i8")[0, 1] = nc * 1000003 + n, int(np.bitwise_xor.reduce(off[cand] * 31 + fid[cand]))
            seen = gather(mine)
            shared = bool((seen == mine[None]).all())          # the same table on every rank (else: every rank for itself, no further collective)
        if shared:
            per = -(-nc // world)
            lo, hi = min(rank * per, nc), min((rank + 1) * per, nc)
            local = np.zeros((per + 1, 16), dtype=np.uint8)      # last row: this rank's reads succeeded
            try:
                if hi > lo:
                    native_io.pread_batch(lut[fid[cand[lo:hi]]], off[cand[lo:hi]], np.full(hi - lo, 15, dtype=np.int64), local.ctypes.data,
                                          np.arange(hi - lo, dtype=np.int64) * 16, threads=self.threads)
                local[per, 0] = 1
            except OSError:
                pass
            full = gather(local)
            if not (full[:, per, 0] == 1).all():
                return False                              # every rank sees the same flags: all of them take the per-entry path
            heads = np.ascontiguousarray(full[:, :per].reshape(-1, 16)[:nc])
        else:
            heads = np.zeros((nc, 16), dtype=np.uint8)
            try:
                native_io.pread_batch(lut[fid[cand]], off[cand], np.full(nc, 15, dtype=np.int64), heads.ctypes.data,
                                      np.arange(nc, dtype=np.int64) * 16, threads=self.threads)
            except OSError:
                return False                              # (an offset at the very end of a file: the per-entry path names the entry)
        ok = (heads[:, 0] == 0) & (heads[:, 1] == ord("B")) & (heads[:, 2] == ord("F")) & (heads[:, 3] == ord("M")) & (heads[:, 4] == ord(" ")) & \
             (heads[:, 5] == 4) & (heads[:, 10] == 4)
        rows_c = np.ascontiguousarray(heads[:, 6:10]).view("<i4").ravel()
        cols_c = np.ascontiguousarray(heads[:, 11:15]).view("<i4").ravel()
        plain = np.zeros(n, dtype=bool)
        rows, cols = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        good = cand[ok]
        plain[good], rows[good], cols[good] = True, rows_c[ok], cols_c[ok]
        self._table = (plain, fid, off + 15, rows, cols, files)
        return True

    def lengths(self):
        """Frames of every entry, from the headers (what the length-balanced sharding needs when no utt2num_frames is given)."""
        if self.index_all():
            plain, _, _, rows, _, _ = self._table
            out = rows.copy()
            for i in np.nonzero(~plain)[0]:
                out[i] = matrix_rows(self.entries[i][1])
            return out
        out = np.empty(len(self.entries), dtype=np.int64)
        for i, (_, rx) in enumerate(self.entries):
            h = self._direct(i)
            out[i] = h[2] if h is not None else matrix_rows(rx)
        return out

    def shape(self, i, slow=None):
        """(rows, cols) of entry i: from its header if it is a plain float32 entry, else by decoding it (the matrix is kept in
        `slow[i]` for the fill that follows)."""
        h = self._direct(i)
        if h is not None:
            return h[2], h[3]
        m = slow.get(i) if slow is not None else None
        if m is None:
            m = self(i)
            if slow is not None:
                slow[i] = m
        return m.shape

    def fill(self, indices, packed, offs, slow=None):
        """Reads the matrices of `indices` into packed[offs[k]:offs[k + 1]] (C-contiguous [frames, dim] float32); `slow`: already
        decoded matrices of non-plain entries by entry index."""
        heads = [self._direct(i) for i in indices]
        dim = packed.shape[1]
        n = len(indices)
        slow = slow if slow is not None else {}
        ks = [k for k in range(n) if heads[k] is not None]
        paths = {heads[k][0] for k in ks}
        fds = {path: self._fd(path, keep=paths) for path in paths}
        for k in range(n):
            if heads[k] is None:
                i = indices[k]
                m = slow[i] if i in slow else self(i)
                packed[int(offs[k]):int(offs[k + 1])] = m
        if not ks:
            return
        from libs.support import native_io
        if native_io.lib() is not None:
            try:
                native_io.pread_batch([fds[heads[k][0]] for k in ks], [heads[k][1] for k in ks], [(int(offs[k + 1]) - int(offs[k])) * dim * 4 for k in ks],
                                      packed.ctypes.data, [int(offs[k]) * dim * 4 for k in ks], threads=self.threads)
            except OSError as e:
                k = ks[e.args[2]] if len(e.args) > 2 else ks[0]
                raise kaldi_io.BadInputFormat("scp entry %r: %s" % (self.entries[indices[k]][1], e.args[1]))
            return
        raw = memoryview(packed.reshape(-1).view(np.uint8))
        for k in ks:
            h, a, b = heads[k], int(offs[k]), int(offs[k + 1])
            fd, pos, want = fds[h[0]], h[1], (b - a) * dim * 4
            got, view = 0, raw[a * dim * 4:b * dim * 4]
            while got < want:
                r = os.preadv(fd, [view[got:]], pos + got)
                if r <= 0:
                    raise kaldi_io.BadInputFormat("scp entry %r: the archive ends inside the matrix" % (self.entries[indices[k]][1],))
                got += r

    def load_batch(self, indices):
        slow = {}
        idx = np.asarray(indices, dtype=np.int64)
        fast = bool(self._table) and len(idx) > 0 and bool(self._table[0][idx].all())     # every entry of the batch is in the header table
        if fast:
            _, _, _, t_rows, t_cols, _ = self._table
            cols = t_cols[idx]
            dim = int(cols[0])
            if (cols != dim).any():
                raise ValueError("feature matrices of different widths in one batch: %s" % sorted(set(cols.tolist())))
            offs = np.zeros(len(idx) + 1, dtype=np.int32)
            np.cumsum(t_rows[idx], out=offs[1:])
        else:
            shapes = [self.shape(i, slow) for i in indices]
            dims = {c for _, c in shapes}
            if len(dims) != 1:
                raise ValueError("feature matrices of different widths in one batch: %s" % sorted(dims))
            dim = dims.pop()
            offs = np.zeros(len(indices) + 1, dtype=np.int32)
            np.cumsum([r for r, _ in shapes], out=offs[1:])
        total = int(offs[-1])
        own = False
        turn, self._turn = self._turn, (self._turn + 1) % len(self._bufs)
        if self._before_fill is not None:
            self._before_fill(turn)
        buf = self._bufs[turn]
        if self._own:
            if buf is None or buf.size < total * dim:
                buf = self._bufs[turn] = np.empty(max(total * dim, 1), dtype=np.float32)
            packed = buf[:total * dim].reshape(total, dim)
        elif buf.ndim == 2 and buf.shape[1] == dim and buf.shape[0] >= total:
            packed = buf[:total]
        else:                                          # does not fit the caller's buffer (one utterance longer than a whole batch): its own array
            packed, own = np.empty((total, dim), dtype=np.float32), True       # (the slot of the rotation is consumed all the same)
        if fast:
            self._fill_table(idx, packed, offs, dim)
        else:
            self.fill(indices, packed, offs, slow)
        return PackedBatch(packed, offs, turn, own)

    def _fill_table(self, idx, packed, offs, dim):
        """fill() for a batch whose entries are all in the header table: the arguments of the one native read call by array indexing."""
        from libs.support import native_io
        _, t_fid, t_pos, _, _, files = self._table
        fid = t_fid[idx]
        used = np.unique(fid)
        keep = {files[j] for j in used.tolist()}
        lut = np.zeros(len(files), dtype=np.int32)
        for j in used.tolist():
            lut[j] = self._fd(files[j], keep=keep)
        o64 = offs.astype(np.int64)
        try:
            native_io.pread_batch(lut[fid], t_pos[idx], (o64[1:] - o64[:-1]) * (dim * 4), packed.ctypes.data, o64[:-1] * (dim * 4), threads=self.threads)
        except OSError as e:
            k = e.args[2] if len(e.args) > 2 else 0
            raise kaldi_io.BadInputFormat("scp entry %r: %s" % (self.entries[int(idx[k])][1], e.args[1]))


class ScpGroupReader(object):
    """kaldi_io.PackedArkReader's interface (peek_dim / read_group) over scp entries IN SCP ORDER: what extract_stream reads `scp:`
    input through when the run is not sharded - batches are written as they finish, memory is one batch, the keys print as they go
    (the reference's loop over read_mat_scp does the same one utterance at a time, extract_embeddings.py:70-83).  Packed ragged
    batches have no padding, so nothing is gained by sorting a single process's utterances by length."""

    def __init__(self, entries, threads=4):
        self.entries = entries
        self.loader = ScpBatchLoader(entries, threads=threads)
        self._at = 0
        self._slow = {}
        self.row_pad = 0                                   # as kaldi_io.PackedArkReader.row_pad

    def close(self):
        self.loader.close()

    def peek_dim(self):
        if self._at >= len(self.entries):
            return None
        return int(self.loader.shape(self._at, self._slow)[1])

    def read_group(self, feats, max_utts=1024):
        cap, dim = feats.shape
        idx, offs, used = [], [0], 0
        while len(idx) < max_utts and self._at < len(self.entries):
            i = self._at
            rows, cols = self.loader.shape(i, self._slow)
            if cols != dim:
                raise kaldi_io.BadInputFormat("scp entry '%s' has %d columns, the table started with %d" % (self.entries[i][0], cols, dim))
            if rows > cap:
                if idx:
                    break                                  # flush what we have first
                self._at += 1
                big = np.empty((rows, dim), dtype=np.float32)
                self.loader.fill([i], big, np.array([0, rows], dtype=np.int32), self._slow)
                self._slow.pop(i, None)
                return [self.entries[i][0]], np.array([0, rows], dtype=np.int32), big
            if used + rows > cap or (idx and used + rows + self.row_pad * (len(idx) + 2) > cap):
                break
            idx.append(i)
            used += rows
            offs.append(used)
            self._at += 1
        offs = np.asarray(offs, dtype=np.int32)
        if idx:
            self.loader.fill(idx, feats[:used], offs, self._slow)
            for i in idx:
                self._slow.pop(i, None)
        return [self.entries[i][0] for i in idx], offs, used


def _shard_segment_utts(lengths=None, batch_frames=0, batch_utts=0, row_pad=0):
    """Utterances per rank between two gathers of the sharded path (ASV_AMD_SHARD_SEGMENT, default 4096; 0 = one gather at the very
    end), rounded to whole batches of the mean utterance: a segment's last batch is a partial one, and on a device-bound run (f32x)
    a 245-utterance batch costs what a 321-utterance batch costs - 13 segments of 4096 x 200-frame utterances were 169 batches
    instead of 156 (profiles/r5x_ark_*.json: --sharded 195 k utterances/s in f32x against 250 k through the stream path, while the
    host-bound bf16 run was already level)."""
    want = max(0, int(os.environ.get("ASV_AMD_SHARD_SEGMENT", "4096")))
    if want == 0 or lengths is None or len(lengths) == 0 or batch_frames <= 0:
        return want
    mean = float(np.mean(lengths))
    per_batch = int(max(1, min(batch_utts if batch_utts > 0 else 1 << 30, (batch_frames - row_pad) // max(1.0, mean + row_pad))))
    return max(per_batch, int(round(want / float(per_batch))) * per_batch)


def extract_sharded_scp(extract_batch, entries, lengths, w, batch_frames, batch_utts, verbose=False, device=None, loader=None, row_pad=0,
                        segment_utts=None, timing=None):
    """Sharded extraction of scp entries (one call per rank, torch.distributed initialised or not):
        extract_batch(list of [T, D] float32 matrices) -> [b, E] tensor
    The scp is walked in segments of `segment_utts` utterances per rank (default: ASV_AMD_SHARD_SEGMENT = 4096): every rank extracts
    its length-balanced share of a segment, one all-gather per segment, and rank 0 writes the segment's ark entries to `w` - in scp
    order - on a writer thread while the next segment is being extracted (libs.amd.shard.extract_sharded_segments).
    Returns the number of embeddings (on every rank)."""
    import queue
    import threading
    import torch.distributed as dist
    from libs.amd import shard
    rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
    load = loader if loader is not None else ScpBatchLoader(entries, threads=_reader_threads())
    # (only rank 0 writes: the keys of a segment are made when it is written - 50 000 tuples up front on every rank were part of the
    #  per-rank cost that does not shrink with the number of ranks)
    keys_of = entries.keys if hasattr(entries, "keys") and not isinstance(entries, dict) else (lambda a, b: [k for k, _ in entries[a:b]])
    todo, failed = queue.Queue(), []

    def write_segments():
        while True:
            item = todo.get()
            if item is None:
                return
            a, b, emb = item
            if failed:
                continue
            try:
                seg_keys = keys_of(a, b)
                if verbose:
                    for key in seg_keys:
                        print("Process utterance for key {0}".format(key))
                w.write(kaldi_io.vec_flt_ark_bytes(seg_keys, emb.cpu().numpy(), as_buffer=True))
            except Exception as e:                       # re-raised on the caller's thread below
                failed.append(e)

    writer = threading.Thread(target=write_segments, name="asv-shard-writer", daemon=True) if rank == 0 else None
    if writer is not None:
        writer.start()
    try:
        n = shard.extract_sharded_segments(extract_batch, lengths, load, (lambda a, b, emb: todo.put((a, b, emb))) if rank == 0 else (lambda a, b, emb: None),
                                           _shard_segment_utts(lengths, batch_frames, batch_utts, row_pad) if segment_utts is None else segment_utts,
                                           max_frames=batch_frames, max_utts=batch_utts, device=device, row_pad=row_pad, timing=timing)
    finally:
        t0 = time.perf_counter()
        if writer is not None:
            todo.put(None)
            writer.join()
        if timing is not None:
            timing["writer_join"] = timing.get("writer_join", 0.0) + time.perf_counter() - t0
        if loader is None:
            load.close()
    if failed:
        raise failed[0]
    return n


def index_ranks(device=None):
    """(rank, world, gather) for ScpBatchLoader.index_all under an initialised process group of more than one rank, else None.  `gather`
    all-gathers a uint8 array - through device memory under NCCL / RCCL (`device`), host memory under gloo."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() < 2 or os.environ.get("ASV_AMD_SHARD_INDEX", "1") == "0":
        return None
    world = dist.get_world_size()
    where = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")

    def gather(local):
        t = torch.from_numpy(np.ascontiguousarray(local)).to(where)
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=torch.uint8, device=where)      # (the concatenated form: gloo takes no other)
        dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy().reshape((world,) + tuple(t.shape))

    return dist.get_rank(), world, gather


def run_sharded(args, model, max_chunk, verbose):
    import torch.distributed as dist
    from libs.amd.pipeline import DeviceSets
    if not _SCP_PREFIX.match(args.feats_rspecifier):
        raise ValueError("--sharded needs random access to the features: pass 'scp:feats.scp' (the reference shards feats.scp too, "
                         "splitDataByLength.sh:44-80), not %r" % args.feats_rspecifier)
    entries = read_scp(args.feats_rspecifier)
    engine = model._amd_engine()
    dev = torch.device("cuda", engine.device_index)
    if max_chunk is None:
        max_chunk = getattr(type(model).extract_embedding, "max_chunk", 10000)
    # the device side is the stream path's: page-locked input buffers the loader reads the files INTO, asynchronous H2D, two engines
    # on two HIP streams, the range status word behind every batch (libs.amd.pipeline.DeviceSets) - until round 5 this path copied a
    # pageable batch synchronously into ONE engine and never looked at the status word
    sets = DeviceSets(model, args.batch_frames, args.batch_utts, engine.feat_dim, max_chunk, n_sets=3, results="device")
    loader = ScpBatchLoader(entries, threads=_reader_threads(),
                            buffers=[sets.host_buffer(k) for k in range(sets.n_sets)], before_fill=sets.input_consumed)
    # all headers in one native call (also what lengths() below reads) - under several ranks each reads 1 / world of them and one
    # all-gather completes the table (ASV_AMD_SHARD_INDEX=0: every rank reads all of them)
    loader.index_all(ranks=index_ranks(dev))
    if args.utt2num_frames:
        table = dict(line.split() for line in open(args.utt2num_frames) if line.strip())
        lengths = np.array([int(table[k]) for k, _ in entries], dtype=np.int64)
    else:
        lengths = loader.lengths()                     # one 15-byte pread per plain float32 entry (a descriptor per ark file, not per entry)

    spent = {"finish": 0.0}

    def extract_batch(mats):
        if getattr(mats, "packed", None) is not None:        # ScpBatchLoader: the batch already lies packed in one (page-locked) buffer
            k = mats.turn                                    # the slot the loader consumed: an over-long batch (own array) keeps the rotation too
            t0 = time.perf_counter()
            sets.finish(k)                                   # the batch before last (same set): done by now; range guard
            spent["finish"] += time.perf_counter() - t0
            return sets.submit(k, mats.offsets, mats.packed if mats.own_array else int(mats.offsets[-1]))
        offs = np.zeros(len(mats) + 1, dtype=np.int32)
        np.cumsum([m.shape[0] for m in mats], out=offs[1:])
        sets.finish(0)
        return sets.submit(0, offs, np.concatenate(mats, axis=0))

    extract_batch.flush = sets.flush                       # extract_sharded_segments calls it before it reads the last results ...
    base = sets.submitted
    extract_batch.final_through = lambda: sets.final_through() - base      # ... and reads a segment's results once its last submission is final
    rank = dist.get_rank() if dist.is_initialized() else 0
    w = kaldi_io.open_or_fd(args.vectors_wspecifier, "wb") if rank == 0 else None
    try:
        t0 = time.perf_counter()
        with torch.cuda.device(dev):
            n = extract_sharded_scp(extract_batch, entries, lengths, w, args.batch_frames, args.batch_utts, verbose, device=dev, loader=loader, row_pad=4,
                                    timing=spent)
        if rank == 0:
            _report_loop("sharded", n, time.perf_counter() - t0, sets, spent)
        if rank == 0 and dist.is_initialized():
            with open("/proc/self/maps") as f:
                rccl = "librccl" in f.read()
            print("Sharded over {0} rank(s); embeddings collected by all_gather_into_tensor, backend {1} (librccl mapped: {2}).".format(
                dist.get_world_size(), dist.get_backend(), rccl))
        return n
    finally:
        loader.close()
        if w is not None:
            w.close()


def main(argv=None):
    print(" ".join(sys.argv))
    args = get_args(argv)
    try:
        if args.nnet_config != "":
            model_blueprint, model_creation = utils.read_nnet_config(args.nnet_config)
        elif args.model_blueprint is not None and args.model_creation is not None:
            model_blueprint, model_creation = args.model_blueprint, args.model_creation
        else:
            raise ValueError("Expected nnet_config or (model_blueprint, model_creation) to exist.")
        if not utils.to_bool(args.use_gpu):
            raise RuntimeError("asv-subtools_amd extracts on a ROCm device only (--use-gpu=true); there is no CPU path")

        sharded = utils.to_bool(args.sharded)
        if sharded:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL on these hosts
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            local_rank = int(os.environ.get("LOCAL_RANK", "0"))
            if args.gpu_id == "":
                args.gpu_id = str(local_rank)                               # one process per GPU
            torch.cuda.set_device(int(args.gpu_id))
            # the process group exists at every world size, 1 included (under torch.distributed.run --nproc-per-node 1 or without
            # a launcher): the collection step is then always RCCL's all-gather - one code path, exercised on a single GPU too
            if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
                os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
            if "MASTER_PORT" not in os.environ:
                # a free local port only for the launcher-less single-rank run: ranks of a real job that each picked their own
                # port would wait for one another until the rendezvous times out
                if int(os.environ["WORLD_SIZE"]) != 1:
                    raise RuntimeError("--sharded with WORLD_SIZE=%s needs MASTER_PORT (and MASTER_ADDR) in the environment: launch with "
                                       "`python -m torch.distributed.run --nproc-per-node N ...` or export them" % os.environ["WORLD_SIZE"])
                import socket
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(args.gpu_id)))   # "nccl" is RCCL on ROCm
            # RCCL builds its communicator inside the first collective (0.1 - 0.2 s): here, with the model still to be loaded, not
            # inside the extraction loop in front of the all-gather
            wdev = torch.device("cuda", int(args.gpu_id))
            warm = torch.zeros(2, dtype=torch.int64, device=wdev)
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)                       # (the two collectives of libs/amd/shard.py, each once)
            wsrc = torch.zeros((8, 8), dtype=torch.float32, device=wdev)
            wdst = torch.empty((8 * dist.get_world_size(), 8), dtype=torch.float32, device=wdev)
            dist.all_gather_into_tensor(wdst, wsrc)
            torch.cuda.synchronize(wdev)
            del warm, wsrc, wdst

        model = utils.create_model_from_py(model_blueprint, model_creation)
        model.load_state_dict(torch.load(args.model_path, map_location="cpu"), strict=False)
        model = utils.select_model_device(model, args.use_gpu, gpu_id=args.gpu_id)
        model.eval()
        max_chunk = args.max_chunk if args.max_chunk > 0 else None
        verbose = utils.to_bool(args.verbose)

        n_done = 0
        if sharded:
            import torch.distributed as dist
            n_done = run_sharded(args, model, max_chunk, verbose)
            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()
        elif _SCP_PREFIX.match(args.feats_rspecifier):
            # random-access input without --sharded (the reference's script reads `scp:` through read_mat_scp): the stream loop over
            # a reader that fills its batches from the scp entries in order - vectors are written batch by batch, in scp order
            reader = ScpGroupReader(read_scp(args.feats_rspecifier), threads=_reader_threads())
            try:
                with kaldi_io.open_or_fd(args.vectors_wspecifier, "wb") as w:
                    n_done = extract_stream(model, None, w, args.batch_frames, args.batch_utts, max_chunk, verbose, reader=reader)
            finally:
                reader.close()
        else:
            with kaldi_io.open_or_fd(args.feats_rspecifier, "rb") as r, kaldi_io.open_or_fd(args.vectors_wspecifier, "wb") as w:
                n_done = extract_stream(model, r, w, args.batch_frames, args.batch_utts, max_chunk, verbose)
        print("Extracted {0} embeddings.".format(n_done))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
