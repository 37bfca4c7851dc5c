"""N > 1 path on CPU: world_size-2 gloo processes shard utterances by length, "extract" with a
deterministic stand-in and collect with the same all-gather code the GPU run uses."""

import os
import socket

import numpy as np
import pytest

from libs.amd import shard


def test_balance_by_length_is_a_balanced_partition():
    r = np.random.RandomState(0)
    lengths = r.randint(200, 1001, size=997)
    for n in (1, 2, 3, 8):
        shards = shard.balance_by_length(lengths, n)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(len(lengths)))
        loads = np.array([lengths[s].sum() for s in shards])
        assert loads.max() - loads.min() <= lengths.max()          # LPT bound
        for s in shards:                                           # inside a shard: longest first
            assert np.all(np.diff(lengths[s]) <= 0)


def test_plan_batches_respects_limits():
    lengths = np.array([500, 400, 300, 300, 200, 100, 100])
    batches = shard.plan_batches(lengths, np.arange(7), max_frames=900, max_utts=3)
    assert batches == [[0, 1], [2, 3, 4], [5, 6]]
    assert shard.plan_batches(lengths, [0], max_frames=10, max_utts=1) == [[0]]     # an over-long utterance still gets a batch


def test_plan_batches_vectorised_form_equals_the_per_utterance_rule():
    """Round 5: plan_batches is one searchsorted per batch; the per-utterance statement of its rule stays as _plan_batches_loop."""
    rs = np.random.RandomState(0)
    for trial in range(200):
        n = int(rs.randint(1, 200))
        lengths = rs.randint(1, 400, size=n)
        if trial % 5 == 0:
            lengths[:] = 200
        idx = rs.permutation(n) if trial % 7 == 0 else np.argsort(-lengths, kind="stable")
        mf, mu, rp = int(rs.choice([10, 300, 900, 2000, 65536])), int(rs.choice([1, 3, 5, 1024])), int(rs.choice([0, 4]))
        assert shard.plan_batches(lengths, idx, mf, mu, rp) == shard._plan_batches_loop(lengths, idx, mf, mu, rp), (trial, mf, mu, rp)
    assert shard.plan_batches([], [], 100, 4) == []


class _LatePipeline(object):
    """Stand-in for libs.amd.pipeline.DeviceSets behind extract_batch: the tensor a call returns is filled in only when `depth`
    later calls have been made (or on flush) - reading it earlier yields NaN."""

    def __init__(self, depth):
        self.depth, self.pending, self.calls, self.flushes = depth, [], 0, 0

    def __call__(self, mats):
        import torch
        out = torch.full((len(mats), 16), float("nan"))
        self.pending.append((self.calls, out, np.stack([_fake_embedding(m) for m in mats])))
        self.calls += 1
        while self.pending and self.pending[0][0] + self.depth <= self.calls:
            _, t, v = self.pending.pop(0)
            t.copy_(torch.from_numpy(v))
        return out

    def flush(self):
        import torch
        self.flushes += 1
        for _, t, v in self.pending:
            t.copy_(torch.from_numpy(v))
        self.pending = []


@pytest.mark.parametrize("depth,segment", [(3, 4), (3, 1), (1, 7), (2, None), (5, 3)])
def test_segments_are_gathered_only_when_their_results_are_final(depth, segment):
    """extract_sharded_segments runs ONE pipeline through all segments: a segment is gathered once `extract_batch.depth` later batches
    have been submitted (or after the final flush), never by flushing at a segment's end - and never before its tensors are final."""
    from libs.amd import synth
    lengths = np.random.RandomState(8).randint(20, 300, size=41)
    load = lambda i: synth.synth_feats(int(lengths[i]), 8, 100 + i)
    pipe = _LatePipeline(depth)
    got = {}

    def on_segment(a, b, emb):
        assert not np.isnan(emb.numpy()).any(), "segment [%d, %d) was read before its results were final" % (a, b)
        got[(a, b)] = emb.numpy().copy()
    n = shard.extract_sharded_segments(pipe, lengths, load, on_segment, segment, max_frames=700, max_utts=4)
    assert n == len(lengths) and pipe.flushes == 1
    want = np.stack([_fake_embedding(load(i)) for i in range(len(lengths))])
    spans = sorted(got)
    assert spans[0][0] == 0 and spans[-1][1] == len(lengths) and all(spans[k][1] == spans[k + 1][0] for k in range(len(spans) - 1))
    if segment:
        assert all(b - a <= segment for a, b in spans)
    assert np.array_equal(np.concatenate([got[k] for k in spans]), want)


class _SetsPipeline(object):
    """Stand-in with libs.amd.pipeline.DeviceSets' bookkeeping behind extract_batch: a batch's tensor is filled in when ITS buffer set is
    finished (before the set's next submission, or on flush); `irregular` batches break the rotation - they run on set 0 whatever
    set is due (what run_sharded did with an utterance longer than --batch-frames until round 6).  final_through() is exact."""

    def __init__(self, n_sets, irregular):
        self.n_sets, self.irregular, self.pending, self.submitted, self.flushes = n_sets, irregular, [None] * n_sets, 0, 0

    def _finish(self, k):
        import torch
        if self.pending[k] is not None:
            _, t, v = self.pending[k]
            t.copy_(torch.from_numpy(v))
            self.pending[k] = None

    def __call__(self, mats):
        import torch
        k = 0 if self.submitted in self.irregular else self.submitted % self.n_sets
        self._finish(k)
        out = torch.full((len(mats), 16), float("nan"))
        self.submitted += 1
        self.pending[k] = (self.submitted, out, np.stack([_fake_embedding(m) for m in mats]))
        return out

    def final_through(self):
        live = [p[0] for p in self.pending if p is not None]
        return (min(live) - 1) if live else self.submitted

    def flush(self):
        self.flushes += 1
        for k in range(self.n_sets):
            self._finish(k)


@pytest.mark.parametrize("segment", [1, 2, 3, 5])
def test_segments_wait_for_final_results_when_the_rotation_is_broken(segment):
    """with `depth = n_sets` a segment was released three submissions after its last batch - wrong as soon as one
    batch leaves the rotation of the buffer sets (unfinished, unchecked results were gathered and written).  The release rule is now
    the pipeline's own count of finished submissions."""
    from libs.amd import synth
    lengths = np.random.RandomState(9).randint(20, 300, size=67)
    load = lambda i: synth.synth_feats(int(lengths[i]), 8, 100 + i)
    pipe = _SetsPipeline(3, irregular={2, 3, 7, 11, 12, 13, 20})
    seen = []

    def on_segment(a, b, emb):
        assert not np.isnan(emb.numpy()).any(), "segment [%d, %d) was read before its results were final" % (a, b)
        seen.append((a, b, emb.numpy().copy()))
    n = shard.extract_sharded_segments(pipe, lengths, load, on_segment, segment, max_frames=500, max_utts=3)
    assert n == len(lengths) and pipe.flushes == 1
    want = np.stack([_fake_embedding(load(i)) for i in range(len(lengths))])
    assert np.array_equal(np.concatenate([e for _, _, e in sorted(seen, key=lambda t: t[0])]), want)


def _fake_embedding(mat, dim=16):
    # deterministic, length- and content-dependent stand-in for the extractor
    v = np.zeros(dim, dtype=np.float32)
    v[: min(dim, mat.shape[1])] = mat.mean(axis=0)[:dim]
    v[-1] = mat.shape[0]
    return v


def _worker(rank, world, port, lengths, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from libs.amd import synth
        load = lambda i: synth.synth_feats(int(lengths[i]), 8, 100 + i)
        extract = lambda mats: torch.from_numpy(np.stack([_fake_embedding(m) for m in mats]))
        out = shard.extract_sharded(extract, lengths, load, max_frames=2000, max_utts=4)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), out.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_extraction_gathers_in_original_order(tmp_path, world):
    import torch.multiprocessing as mp
    from libs.amd import synth
    lengths = np.random.RandomState(5).randint(20, 400, size=37)
    mp.spawn(_worker, args=(world, _free_port(), lengths, str(tmp_path)), nprocs=world, join=True)
    want = np.stack([_fake_embedding(synth.synth_feats(int(n), 8, 100 + i)) for i, n in enumerate(lengths)])
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npy" % r))
        assert got.shape == want.shape and np.array_equal(got, want), "rank %d" % r


def test_single_process_gather_is_a_permutation():
    import torch
    lengths = np.array([5, 9, 2, 7])
    shards = shard.balance_by_length(lengths, 1)
    local = torch.arange(4, dtype=torch.float32)[:, None] * torch.ones(1, 3)
    out = shard.gather_embeddings(local, shards[0], shards)
    # local row j holds utterance shards[0][j]
    assert [int(out[i, 0]) for i in shards[0]] == [0, 1, 2, 3]


def _worker_edge(rank, world, port, lengths, out_dir, fail_rank):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from libs.amd import synth

        def load(i):
            if rank == fail_rank:
                raise IOError("rank %d cannot read utterance %d" % (rank, i))
            return synth.synth_feats(int(lengths[i]), 8, 100 + i)
        extract = lambda mats: torch.from_numpy(np.stack([_fake_embedding(m) for m in mats]))
        try:
            out = shard.extract_sharded(extract, lengths, load, max_frames=2000, max_utts=4)
            np.save(os.path.join(out_dir, "rank%d.npy" % rank), out.numpy())
        except Exception as e:
            with open(os.path.join(out_dir, "rank%d.err" % rank), "w") as f:
                f.write("%s: %s" % (type(e).__name__, e))
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_utterances_contributes_empty_shards(tmp_path):
    """a rank without utterances adds zero rows to the gather instead of raising in front of the collective."""
    import torch.multiprocessing as mp
    from libs.amd import synth
    lengths = np.array([50, 30])
    mp.spawn(_worker_edge, args=(3, _free_port(), lengths, str(tmp_path), -1), nprocs=3, join=True)
    want = np.stack([_fake_embedding(synth.synth_feats(int(n), 8, 100 + i)) for i, n in enumerate(lengths)])
    for r in range(3):
        assert np.array_equal(np.load(tmp_path / ("rank%d.npy" % r)), want), "rank %d" % r


def test_a_failing_rank_makes_every_rank_raise_before_the_gather(tmp_path):
    """a per-rank read error is agreed on with one tiny all-reduce; nobody waits in the all-gather for a timeout."""
    import torch.multiprocessing as mp
    lengths = np.random.RandomState(6).randint(20, 100, size=9)
    mp.spawn(_worker_edge, args=(2, _free_port(), lengths, str(tmp_path), 1), nprocs=2, join=True)
    e0, e1 = (tmp_path / "rank0.err").read_text(), (tmp_path / "rank1.err").read_text()
    assert e1.startswith("OSError: rank 1 cannot read") and "another rank failed" in e0, (e0, e1)


def _worker_late_failure(rank, world, port, lengths, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from libs.amd import synth

        def load(i):
            if rank == 1 and i >= 12:
                raise IOError("rank 1 cannot read utterance %d" % i)
            return synth.synth_feats(int(lengths[i]), 8, 100 + i)
        pipe = _LatePipeline(3)
        seen = []
        try:
            shard.extract_sharded_segments(pipe, lengths, load, lambda a, b, emb: seen.append((a, b)), 3, max_frames=600, max_utts=2)
            seen.append("finished")
        except Exception as e:
            seen.append("%s: %s" % (type(e).__name__, e))
        with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
            f.write(repr(seen))
    finally:
        dist.destroy_process_group()


def test_a_rank_failing_in_a_later_segment_stops_every_rank_at_the_same_agreement(tmp_path):
    """The failure reaches both ranks at ONE agreement (the same segments delivered on both before it - none behind the failing
    utterances; segments still waiting for their results when the read failed are given up, not flushed), nobody hangs."""
    import torch.multiprocessing as mp
    lengths = np.random.RandomState(6).randint(20, 100, size=30)
    mp.spawn(_worker_late_failure, args=(2, _free_port(), lengths, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = eval((tmp_path / "rank0.txt").read_text()), eval((tmp_path / "rank1.txt").read_text())
    assert r1[-1].startswith("OSError: rank 1 cannot read") and "another rank failed" in r0[-1], (r0, r1)
    assert r0[:-1] == r1[:-1] and all(b <= 12 for a, b in r0[:-1]), (r0, r1)


def _worker_one_sided(rank, world, port, lengths, out_dir, what):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=40))
    from libs.amd import synth
    import torch

    def load(i):
        if what == "interrupt" and rank == 1 and i >= 12:
            raise KeyboardInterrupt()
        return synth.synth_feats(int(lengths[i]), 8, 100 + i)

    def on_segment(a, b, emb):
        if what == "on_segment" and rank == 1 and a >= 6:
            raise RuntimeError("rank 1 cannot write segment %d" % a)
    extract = lambda mats: torch.from_numpy(np.stack([_fake_embedding(m) for m in mats]))
    import time
    t0 = time.time()
    try:
        shard.extract_sharded_segments(extract, lengths, load, on_segment, 3, max_frames=600, max_utts=2)
        res = "finished"
    except BaseException as e:
        res = "%s: %s" % (type(e).__name__, e)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%.1f %s" % (time.time() - t0, res))
    try:
        dist.destroy_process_group()
    except Exception:
        pass


@pytest.mark.parametrize("what", ["interrupt", "on_segment"])
def test_a_one_sided_exit_does_not_leave_the_other_rank_in_a_collective(tmp_path, what):
    """a KeyboardInterrupt / SystemExit on one rank's submitting loop is forwarded through the agreement like any other
    failure; a failure of one rank's on_segment (outside any agreement) tears its process group down, so the other rank's next
    collective fails at once instead of waiting for the collective's timeout (40 s here)."""
    import torch.multiprocessing as mp
    lengths = np.random.RandomState(6).randint(20, 100, size=30)
    mp.spawn(_worker_one_sided, args=(2, _free_port(), lengths, str(tmp_path), what), nprocs=2, join=True)
    r0, r1 = (tmp_path / "rank0.txt").read_text(), (tmp_path / "rank1.txt").read_text()
    assert float(r0.split()[0]) < 30 and float(r1.split()[0]) < 30, (r0, r1)
    assert "finished" not in r0 and "finished" not in r1, (r0, r1)
    if what == "interrupt":
        assert "KeyboardInterrupt" in r1 and "another rank failed" in r0, (r0, r1)
    else:
        assert "rank 1 cannot write" in r1, (r0, r1)


def _load_script():
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("extract_embeddings_mod", os.path.join(repo, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py"))
    ee = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ee)
    return ee


def _write_feature_table(directory, utts, broken_for=None):
    """feats.ark / feats.scp of `utts` float32 matrices of 1 .. 40 frames x 8; `broken_for`: an entry whose offset points behind the file."""
    rs = np.random.RandomState(3)
    ark, scp = os.path.join(directory, "feats.ark"), os.path.join(directory, "feats.scp")
    lens = []
    with open(ark, "wb") as f, open(scp, "w") as s:
        pos = 0
        for i in range(utts):
            t = int(rs.randint(1, 41))
            lens.append(t)
            key = ("u%05d " % i).encode()
            f.write(key)
            pos += len(key)
            s.write("u%05d %s:%d\n" % (i, ark, pos if i != broken_for else 10 ** 9))
            blob = b"\0BFM \4" + np.int32(t).tobytes() + b"\4" + np.int32(8).tobytes() + rs.randn(t, 8).astype(np.float32).tobytes()
            f.write(blob)
            pos += len(blob)
    return scp, lens


def _worker_index(rank, world, port, scp, out_dir, differ):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ee = _load_script()
        entries = ee.read_scp("scp:" + scp)
        if differ and rank == 1:
            entries = ee.read_scp("scp:" + scp + (".missing" if differ == "missing" else ".short"))
        calls = []
        ranks = ee.index_ranks()
        assert ranks is not None and ranks[:2] == (rank, world)
        counted = (ranks[0], ranks[1], lambda a: (calls.append(a.shape), ranks[2](a))[1])
        shared = ee.ScpBatchLoader(entries, threads=2)
        ok = shared.index_all(ranks=counted)
        alone = ee.ScpBatchLoader(entries, threads=2)
        ok_alone = alone.index_all()
        same = bool(ok == ok_alone) and (not ok or all(np.array_equal(a, b) for a, b in zip(shared._table[:5], alone._table[:5])))
        np.save(os.path.join(out_dir, "index%d.npy" % rank), np.array([int(ok), int(ok_alone), int(same), len(calls)] + ([int(calls[-1][0])] if calls else [0])))
        if ok:
            np.save(os.path.join(out_dir, "lengths%d.npy" % rank), shared.lengths())
        shared.close(); alone.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_header_pass_shared_between_the_ranks_equals_the_whole_pass(tmp_path, world):
    """--sharded with several ranks (round 6): every rank reads the headers of 1 / world of the table and one all-gather completes it - the
    table (plain flags, file ids, payload offsets, rows, columns) and the lengths are what the whole pass on one rank gives; each rank's
    second gather carried ceil(n / world) + 1 rows, not n."""
    import torch.multiprocessing as mp
    from libs.support import native_io
    if native_io.lib() is None:
        pytest.skip("libasv_io.so not built")
    scp, lens = _write_feature_table(str(tmp_path), 101)
    mp.spawn(_worker_index, args=(world, _free_port(), scp, str(tmp_path), False), nprocs=world, join=True)
    for r in range(world):
        ok, ok_alone, same, n_calls, rows = np.load(tmp_path / ("index%d.npy" % r))
        assert (ok, ok_alone, same, n_calls) == (1, 1, 1, 2) and rows == -(-101 // world) + 1, (r, ok, ok_alone, same, n_calls, rows)
        assert np.array_equal(np.load(tmp_path / ("lengths%d.npy" % r)), np.asarray(lens))


def test_header_pass_ranks_with_different_tables_or_a_failed_read_agree_on_the_fallback(tmp_path):
    """Two ranks that do not hold the same table notice it in the first (16-byte) gather and index for themselves - no second collective
    with mismatched shapes; a header read that fails on one rank sends BOTH to the per-entry path (index_all -> False on both)."""
    import torch.multiprocessing as mp
    from libs.support import native_io
    if native_io.lib() is None:
        pytest.skip("libasv_io.so not built")
    d1 = tmp_path / "a"
    d1.mkdir()
    scp, lens = _write_feature_table(str(d1), 50)
    with open(scp) as f, open(scp + ".short", "w") as g:
        g.writelines(f.readlines()[:-1])
    mp.spawn(_worker_index, args=(2, _free_port(), scp, str(d1), True), nprocs=2, join=True)
    for r in range(2):
        ok, ok_alone, same, n_calls, rows = np.load(d1 / ("index%d.npy" % r))
        assert (ok, ok_alone, same, n_calls) == (1, 1, 1, 1), (r, ok, ok_alone, same, n_calls)
    # a rank that cannot index at all (its table names a file it does not see) still takes part in the agreement: nobody waits for it
    with open(scp) as f, open(scp + ".missing", "w") as g:
        g.writelines(line.replace("feats.ark", "not_there.ark") for line in f)
    mp.spawn(_worker_index, args=(2, _free_port(), scp, str(d1), "missing"), nprocs=2, join=True)
    got = [tuple(int(v) for v in np.load(d1 / ("index%d.npy" % r))[:4]) for r in range(2)]
    assert got == [(1, 1, 1, 1), (0, 0, 1, 1)], got
    d2 = tmp_path / "b"
    d2.mkdir()
    scp2, _ = _write_feature_table(str(d2), 50, broken_for=40)           # the second rank's half holds the entry behind the end of the file
    mp.spawn(_worker_index, args=(2, _free_port(), scp2, str(d2), False), nprocs=2, join=True)
    for r in range(2):
        ok, ok_alone, same, n_calls, rows = np.load(d2 / ("index%d.npy" % r))
        assert (ok, ok_alone, n_calls) == (0, 0, 2), (r, ok, ok_alone, n_calls)
