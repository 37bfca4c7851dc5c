# -*- coding:utf-8 -*-
"""Multi-GPU extraction: one process per GPU, utterances sharded by length, embeddings collected
with ONE all-gather (RCCL over xGMI on the GPU box, gloo in the CPU tests).

The reference shards at the process level: `nj` jobs each run the Python extractor on a
length-balanced split of feats.scp and the per-job xvector.JOB.scp files are concatenated
(pipeline/extract_xvectors_for_pytorch.sh:90-100,125-151; splitDataByLength.sh:44-80).  Here
the same partition is computed identically on every rank (it only depends on the utterance
lengths), each rank extracts its shard, and `all_gather_into_tensor` replaces `cat`.  Utterances
are independent in eval mode, so there is no other exchange on the path.

The payload is tiny (VoxCeleb1-O: 4708 x 192 x 4 B = 3.6 MB in total), i.e. latency-bound: one
collective for the whole run, never one per batch.
"""

import numpy as np


def balance_by_length(lengths, n_shards):
    """Length-balanced partition (the goal of splitDataByLength.sh:44-80): longest first, each
    utterance goes to the shard with the fewest frames so far.  Deterministic: ties break on the
    lower index / lower shard id.  Returns a list of index arrays (ascending inside a shard by
    descending length, which is also the order the shard is batched in)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    if n_shards < 1:
        raise ValueError("n_shards must be >= 1")
    order = np.argsort(-lengths, kind="stable")
    if n_shards == 1:
        return [order.astype(np.int64)]
    import heapq
    heap = [(0, s) for s in range(n_shards)]                # (frames so far, shard id): the smallest pair = fewest frames, lower id on ties
    shards = [[] for _ in range(n_shards)]
    for idx, n in zip(order.tolist(), lengths[order].tolist()):
        load, s = heap[0]
        shards[s].append(idx)
        heapq.heapreplace(heap, (load + n, s))
    return [np.asarray(s, dtype=np.int64) for s in shards]


def plan_batches(lengths, indices, max_frames=65536, max_utts=1024, row_pad=0):
    """Groups a shard's utterances (already length-sorted) into batches bounded by total frames
    and count; neighbours have similar length, so the packed ragged batch wastes no padding.
    row_pad: rows the device layout adds per utterance (its 4 gap rows): frames + row_pad * (utterances + 1) <= max_frames as well,
    so that a batch fills whole tiles of the device's row count (see pipeline/onestep/extract_embeddings.py extract_stream).
    One searchsorted per BATCH on the running sum of (frames + row_pad) - the per-utterance Python loop this replaces was 20 ms per
    50 000 utterances in front of the first read (same batches: tests/test_shard_gloo.py compares the two)."""
    idx = np.asarray(indices, dtype=np.int64)
    if idx.size == 0:
        return []
    run = np.concatenate([[0], np.cumsum(np.asarray(lengths, dtype=np.int64)[idx] + row_pad)])      # run[j] = frames + pads of the first j utterances
    batches, i, n = [], 0, int(idx.size)
    while i < n:
        # the largest m >= 1 with frames(i .. i + m) + row_pad * (m + 1) <= max_frames and m <= max_utts (an over-long utterance still gets a batch)
        m = int(np.searchsorted(run, run[i] + max_frames - row_pad, side="right")) - 1 - i
        m = max(1, min(m, max_utts, n - i))
        batches.append(idx[i:i + m].tolist())
        i += m
    return batches


def _plan_batches_loop(lengths, indices, max_frames=65536, max_utts=1024, row_pad=0):
    """The per-utterance statement of plan_batches' rule (what the tests hold the vectorised form against)."""
    batches, cur, frames = [], [], 0
    for i in indices:
        n = int(lengths[i])
        if cur and (frames + n + row_pad * (len(cur) + 2) > max_frames or len(cur) >= max_utts):
            batches.append(cur)
            cur, frames = [], 0
        cur.append(int(i))
        frames += n
    if cur:
        batches.append(cur)
    return batches


def gather_embeddings(local, local_indices, shards, group=None):
    """local: [n_local, E] tensor of this rank's embeddings in `local_indices` order (== shards[rank]).
    Returns [n_total, E] in original utterance order on every rank."""
    import torch
    import torch.distributed as dist
    inited = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if inited else 1
    n_total = int(sum(len(s) for s in shards))
    E = local.shape[1]
    if not inited:                       # no process group: a plain single-process run.  With one, the collective runs at any world
                                         # size, 1 included (the one-GPU check that RCCL loads and delivers: tests/test_gpu_rccl.py)
        out = torch.empty((n_total, E), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(np.asarray(local_indices), device=local.device)] = local
        return out
    n_pad = max(len(s) for s in shards)
    padded = torch.zeros((n_pad, E), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    gathered = torch.empty((world * n_pad, E), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    # every rank knows every shard's index list (same deterministic plan): no index exchange
    src = np.concatenate([r * n_pad + np.arange(len(s)) for r, s in enumerate(shards)])
    dst = np.concatenate([np.asarray(s) for s in shards])
    out = torch.empty((n_total, E), dtype=local.dtype, device=local.device)
    out[torch.as_tensor(dst, device=local.device)] = gathered[torch.as_tensor(src, device=local.device)]
    return out


def _agree_or_raise(err, embed_dim, device, group, rank, world):
    """One tiny all-reduce (MAX over [failed, E]) in front of the gather: a rank whose read or extraction failed must not leave
    the others blocked in the all-gather until the collective's timeout, and a rank without utterances learns the
    embedding width from the others.  Every rank raises when any rank failed."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if err is not None else 0, int(embed_dim)], dtype=torch.int64, device=device if device is not None else "cpu")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    failed, width = (int(v) for v in flag.cpu().tolist())
    if err is not None:
        raise _Agreed(err)
    if failed:
        raise _Agreed(RuntimeError("sharded extraction: another rank failed (this is rank %d of %d); see its log" % (rank, world)))
    return width


def extract_sharded_segments(extract_batch, lengths, load_utt, on_segment, segment_utts=None, max_frames=65536, max_utts=1024, group=None,
                             device=None, row_pad=0, timing=None):
    """Sharded extraction in SEGMENTS of the utterance list, one collective pair per segment (round 5).
        extract_batch(list_of_mats) -> [b, E] tensor on `device`     (e.g. a libs.amd.pipeline.DeviceSets wrapper)
        load_utt(i) -> [T_i, D] float32 matrix of utterance i          (only called for this rank's share; `.load_batch(indices)` if it has one)
        on_segment(a, b, emb)                                          emb = [b - a, E] embeddings of utterances a .. b - 1 in their order,
                                                                       called on EVERY rank as soon as the segment has been gathered
    Utterances [a, b) of a segment (`segment_utts` per rank; None = everything in one segment) are balanced by length over the ranks
    and batched like the whole list used to be.  ONE pipeline runs through all segments: the reader stays one batch ahead across
    segment ends, and a segment is gathered when its results are final, not when its last batch has been submitted - a pipelined
    extract_batch says which submissions are final (`extract_batch.final_through()`: libs.amd.pipeline.DeviceSets counts the
    batches it has finished and range-checked; or `extract_batch.depth` = that many later submissions; neither = final on return), so nothing is flushed before the very end and the
    device never drains at a segment end.  Per segment the ranks agree that nobody failed (one tiny all-reduce) and all-gather it.
    What this buys: rank 0 can copy out and write segment s while later segments are extracted - with one gather at the very end
    the 102 MB of 50 000 x-vectors (device -> host, ark packing, write) were a serial tail of 0.14 s behind a 0.2 s extraction loop
    (profiles/r5z_bench.json: --sharded 150 k utterances/s against 251 k through the stream path).  The payload per collective stays
    latency-sized (4096 x 512 x 4 B = 8 MB), their number small.  Segments are planned one ahead of their first batch (the plan of
    50 000 utterances in one go stood in front of the first read).
    A rank with nothing in a segment contributes zero rows; a rank whose read / extraction raises makes EVERY rank raise at the next
    agreement (no rank is left waiting in a collective; all ranks run the same sequence of collectives).  Returns the number of
    utterances.  timing: a dict that receives where the submitting thread's time went (seconds: plan, reader = waiting for the batch
    being read, extract = inside extract_batch, join = waiting for the collector at the end) and the collector thread's own total."""
    import collections
    import time
    import torch
    import torch.distributed as dist
    inited = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if inited else 0
    world = dist.get_world_size(group) if inited else 1
    lengths = np.asarray(lengths, dtype=np.int64)
    n = int(lengths.shape[0])
    step = max(n, 1) if not segment_utts else max(1, int(segment_utts)) * world
    bounds = [(a, min(n, a + step)) for a in range(0, max(n, 1), step)]

    spent = timing if timing is not None else {}
    for key in ("plan", "reader", "extract", "join", "collector"):
        spent.setdefault(key, 0.0)

    def plan(s):
        t0 = time.perf_counter()
        a, b = bounds[s]
        shards = balance_by_length(lengths[a:b], world)
        out = a, b, shards, plan_batches(lengths, shards[rank] + a, max_frames, max_utts, row_pad)
        spent["plan"] += time.perf_counter() - t0
        return out

    from concurrent.futures import ThreadPoolExecutor
    import queue
    import threading
    whole = getattr(load_utt, "load_batch", None)
    fetch = (lambda batch: whole(batch)) if whole is not None else (lambda batch: [load_utt(i) for i in batch])
    flush = getattr(extract_batch, "flush", None)          # a pipelined extract_batch returns tensors whose work is still in flight
    depth = int(getattr(extract_batch, "depth", 0)) if flush is not None else 0
    # the exact form of the same knowledge: how many of the submissions so far are final (libs.amd.pipeline.DeviceSets.final_through);
    # preferred over `depth`, which presumes that batches rotate over the pipeline's buffer sets without exception
    final_through = getattr(extract_batch, "final_through", None) if flush is not None else None
    pool = ThreadPoolExecutor(1)
    done = collections.deque()                              # submitted segments whose results are not final yet: (a, b, shards, outs, index of the last submission)
    submitted = 0
    # The collectives, the reordering and on_segment run on a COLLECTOR thread, segment after segment in order (every rank: the same
    # sequence of collectives): the agreement reads its flag back on the host, i.e. waits until the device has run the tiny all-reduce
    # behind whatever the engines have queued - on the submitting thread that wait drained the device's queue at every segment
    # (f32x, 1 ms kernels: --sharded 170 k utterances/s against 240 k through the stream path, profiles/r5v_ark_*.json).
    todo, state = queue.Queue(), {"err": None}

    def gather(entry, err):
        a, b, shards, outs, _ = entry
        local = torch.cat(outs, dim=0) if (outs and err is None) else None
        width = _agree_or_raise(err, local.shape[1] if local is not None else 0, device, group, rank, world)
        if local is None:
            if width == 0:
                raise ValueError("sharded extraction: no utterances at all")
            local = torch.zeros((0, width), dtype=torch.float32)
        if device is not None:
            local = local.to(device)
        on_segment(a, b, gather_embeddings(local, shards[rank], shards, group=group))

    def collect():
        if device is not None and getattr(device, "type", "cpu") == "cuda":
            torch.cuda.set_device(device)
        while True:
            item = todo.get()
            if item is None:
                return
            if state["err"] is None:
                t0 = time.perf_counter()
                try:
                    gather(*item)
                except BaseException as e:                  # (an agreed failure, or anything else: the submitting thread stops at its next batch)
                    state["err"] = e
                spent["collector"] += time.perf_counter() - t0

    collector = threading.Thread(target=collect, name="asv-shard-collector", daemon=True)
    collector.start()
    try:
        try:
            nxt = plan(0)
            pending = collections.deque(nxt[3])              # batches planned but not handed to the reader yet
            ahead = pool.submit(fetch, pending.popleft()) if pending else None
            for s in range(len(bounds)):
                a, b, shards, batches = nxt
                nxt = plan(s + 1) if s + 1 < len(bounds) else None
                if nxt is not None:
                    pending.extend(nxt[3])
                    if ahead is None and pending:
                        ahead = pool.submit(fetch, pending.popleft())
                outs = []
                for _ in batches:
                    if state["err"] is not None:
                        raise _Stop()
                    t0 = time.perf_counter()
                    mats = ahead.result()
                    ahead = pool.submit(fetch, pending.popleft()) if pending else None
                    t1 = time.perf_counter()
                    outs.append(extract_batch(mats))
                    spent["reader"] += t1 - t0
                    spent["extract"] += time.perf_counter() - t1
                    submitted += 1
                    limit = final_through() if final_through is not None else submitted - depth
                    while done and done[0][4] <= limit:                 # (its last batch has been finished by the pipeline itself)
                        todo.put((done.popleft(), None))
                done.append((a, b, shards, outs, submitted))
                if depth == 0 and final_through is None:
                    if flush is not None:
                        flush()
                    todo.put((done.popleft(), None))
            if flush is not None:                           # finished (and range-checked) before the last results are read
                flush()
            while done:
                todo.put((done.popleft(), None))
        except _Stop:
            pass
        except BaseException as e:                          # a failure of THIS rank (KeyboardInterrupt / SystemExit included): told to
            todo.put(((0, 0, [np.zeros(0, dtype=np.int64)] * world, [], 0), e))      # every rank at the collector's next agreement
    finally:
        t0 = time.perf_counter()
        todo.put(None)
        collector.join()
        pool.shutdown(wait=True)
        spent["join"] += time.perf_counter() - t0
        # the collector failed on THIS rank outside an agreement (on_segment raised, a collective failed locally; an agreed failure has
        # reached every rank already): it has skipped its part of every later collective, and the other ranks would sit in the next
        # segment's all-reduce until the collective times out - the process group is torn down instead
        if world > 1 and state["err"] is not None and not isinstance(state["err"], _Agreed):
            _abort_group(group, inited)
    if state["err"] is not None:
        raise state["err"].error if isinstance(state["err"], _Agreed) else state["err"]
    return n


def _abort_group(group, inited):
    """Ends this rank's part in the process group so that the other ranks' pending collectives fail instead of waiting for it."""
    if not inited:
        return
    import torch.distributed as dist
    try:
        abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
        if abort is not None:
            abort(group)
        else:
            dist.destroy_process_group(group)
    except Exception:                                       # nothing more can be done from here: the original error is what the caller sees
        pass


class _Stop(Exception):
    """The collector thread has failed: the submitting loop ends."""


class _Agreed(Exception):
    """An error every rank has been told about (carried out of the pipeline loop as it is)."""
    def __init__(self, error):
        Exception.__init__(self, str(error))
        self.error = error


def extract_sharded(extract_batch, lengths, load_utt, max_frames=65536, max_utts=1024, group=None, device=None, row_pad=0):
    """Full sharded extraction with ONE gather at the end (extract_sharded_segments with a single segment).
        extract_batch(list_of_mats) -> [b, E] tensor on `device`     (e.g. Engine.extract_device wrapper)
        load_utt(i) -> [T_i, D] float32 matrix of utterance i          (only called for this rank's shard)
    Returns [n_total, E] (original order) on every rank.  A rank with an empty shard (more ranks than utterances) contributes
    zero rows; a rank whose read / extraction raises makes EVERY rank raise before the all-gather (no rank is left waiting in a
    collective)."""
    got = []
    extract_sharded_segments(extract_batch, lengths, load_utt, lambda a, b, emb: got.append(emb), None, max_frames, max_utts, group, device, row_pad)
    return got[0]
