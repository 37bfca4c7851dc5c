#!/usr/bin/env python3
"""What does the HOST do under N ranks?  ("scaling risk is host-side feeding"; no device needed.)

N in {1, 2, 4, 8} concurrent processes - the ranks of `extract_embeddings.py --sharded true` on one node - each run the rank-side host work
of pipeline/onestep/extract_embeddings.py extract_sharded_scp on ITS share of one feature table read from the page cache:

    read_scp -> ScpBatchLoader.index_all (the headers: 1 / N of them per rank + one all-gather - a shared-memory gather between the harness'
    processes stands in for the script's RCCL one; --whole-index: all of them on every rank, the path before round 6's last step) -> lengths -> per segment: shard.balance_by_length over the N ranks,
    shard.plan_batches (row-aware, --batch-frames 65536 / --batch-utts 1024) -> ScpBatchLoader.load_batch of every batch of the rank
    (positioned native reads into three rotating [65536, 80] buffers, as into libs.amd.pipeline.DeviceSets' page-locked ones)

and, on rank 0 only (as in the script: rank 0 writes), the packing of ALL N shards' embeddings into ark bytes (kaldi_io.vec_flt_ark_bytes, 512-dim,
written to /dev/null).  No device, no collective but that one: the device side and the embeddings' gather are measured elsewhere (bench.py, tools/bench_pipeline.py) -
this is the part that shares ONE host between the ranks.  Reported per (table, N, reader threads per rank): the wall time from the common start
to the last rank's end, aggregate utterances/s and GB/s, the slowest rank's split, and the reader-thread count that maximises the aggregate.

    python tools/bench_loaders.py [--utts 50000] [--ragged-utts 20000] [--ranks 1,2,4,8] [--threads 1,2,4,8] [--dir /tmp/asv_loaders] [--repeats 3]

The reference shards the same way with `nj` independent processes per machine (/root/reference/pytorch/pipeline/extract_xvectors_for_pytorch.sh:
90-100, 125-151; splitDataByLength.sh:44-80), each reading its own split through Kaldi pipes."""
import argparse
import importlib.util
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "asv-subtools_amd", "pytorch"))


def load_script():
    spec = importlib.util.spec_from_file_location("extract_embeddings_mod", os.path.join(REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py"))
    ee = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ee)
    return ee


def write_table(directory, name, utts, lengths=None, frames=200, dim=80):
    """feats ark + scp of `utts` float32 matrices (64 distinct payloads); lengths = (lo, hi): one of 64 lengths ~ U[lo, hi] per utterance."""
    os.makedirs(directory, exist_ok=True)
    ark, scp = os.path.join(directory, name + ".ark"), os.path.join(directory, name + ".scp")
    rs = np.random.RandomState(7)
    lens = [frames] * 64 if lengths is None else [int(v) for v in rs.randint(lengths[0], lengths[1] + 1, size=64)]
    payload = [b"\0BFM \4" + np.int32(t).tobytes() + b"\4" + np.int32(dim).tobytes() + rs.randn(t, dim).astype(np.float32).tobytes() for t in lens]
    with open(ark, "wb") as f, open(scp, "w") as s:
        pos = 0
        for i in range(utts):
            key = ("utt%07d " % i).encode()
            f.write(key)
            pos += len(key)
            s.write("utt%07d %s:%d\n" % (i, ark, pos))
            j = (i * 37) % 64 if lengths is not None else i % 64
            f.write(payload[j])
            pos += len(payload[j])
    return scp, os.path.getsize(ark)


def rank_main(rank, world, scp, threads, ready, go, out, share=None, batch_frames=65536, batch_utts=1024, segment=4096, embed_dim=512):
    ee = load_script()                                   # (imports torch: seconds, outside the measurement)
    ranks = None
    if share is not None and world > 1:
        # The header pass is shared between the ranks (ScpBatchLoader.index_all(ranks=...)): each reads 1 / world of the headers and an all-gather
        # of 16 bytes per entry completes the table.  The script does it over its RCCL group (~0.1 ms for this size); here a shared-memory
        # gather between the harness' processes stands in for it (a gloo group over loopback was tried first: seconds per collective on the
        # GPU hosts' network set-up - it measured gloo, not the loaders: profiles/r6z_loaders_gloo_standin.txt).
        from multiprocessing import shared_memory
        shm_name, rows_cap, barrier = share
        shm = shared_memory.SharedMemory(name=shm_name)
        board = np.ndarray((world, rows_cap, 16), dtype=np.uint8, buffer=shm.buf)

        def gather(local):
            k = local.shape[0]
            board[rank, :k] = local
            barrier.wait()
            full = board[:, :k].copy()
            barrier.wait()
            return full

        ranks = (rank, world, gather)
    from libs.amd import shard
    from libs.support import kaldi_io, native_io
    native_io.lib()
    spent = {"index": 0.0, "plan": 0.0, "read": 0.0, "pack": 0.0}
    ready.put(rank)
    go.wait()                                            # common start: the ranks contend for the host from the same moment
    begin = time.time()
    t_begin = time.perf_counter()
    entries = ee.read_scp("scp:" + scp)
    bufs = [np.empty((batch_frames, 80), dtype=np.float32) for _ in range(3)]
    loader = ee.ScpBatchLoader(entries, threads=threads, buffers=bufs)
    t0 = time.perf_counter()
    loader.index_all(ranks=ranks)
    lengths = loader.lengths()
    spent["index"] = time.perf_counter() - t0
    n = len(lengths)
    seg = ee._shard_segment_utts(lengths, batch_frames, batch_utts, 4) if segment else n
    step = max(1, seg) * world
    keys_of = entries.keys if hasattr(entries, "keys") else (lambda a, b: [k for k, _ in entries[a:b]])      # (as the script: made per segment, rank 0 only)
    mine, frames, nbytes = 0, 0, 0
    sink = open(os.devnull, "wb")
    for a in range(0, n, step):
        b = min(n, a + step)
        t0 = time.perf_counter()
        shards = shard.balance_by_length(lengths[a:b], world)
        batches = shard.plan_batches(lengths, shards[rank] + a, batch_frames, batch_utts, 4)
        spent["plan"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        for batch in batches:
            pb = loader.load_batch(batch)
            mine += len(pb)
            frames += int(pb.offsets[-1])
            nbytes += int(pb.offsets[-1]) * pb.packed.shape[1] * 4
        spent["read"] += time.perf_counter() - t0
        if rank == 0:                                    # rank 0 writes every rank's vectors of the segment
            t0 = time.perf_counter()
            sink.write(kaldi_io.vec_flt_ark_bytes(keys_of(a, b), np.zeros((b - a, embed_dim), dtype=np.float32), as_buffer=True))
            spent["pack"] += time.perf_counter() - t0
    loader.close()
    total = time.perf_counter() - t_begin
    if ranks is not None:
        shm.close()
    out.put({"rank": rank, "utts": mine, "frames": frames, "bytes": nbytes, "seconds": total, "begin": begin, "end": time.time(), **{k: round(v, 4) for k, v in spent.items()}})


def run(scp, world, threads, repeats, shared_index=True):
    best = None
    for _ in range(repeats):
        ctx = mp.get_context("spawn")
        out, ready, go = ctx.Queue(), ctx.Queue(), ctx.Event()
        share, shm = None, None
        if shared_index and world > 1:
            from multiprocessing import shared_memory
            with open(scp) as f:
                rows_cap = sum(1 for _ in f) + 2
            shm = shared_memory.SharedMemory(create=True, size=world * rows_cap * 16)
            share = (shm.name, rows_cap, ctx.Barrier(world))
        procs = [ctx.Process(target=rank_main, args=(r, world, scp, threads, ready, go, out, share)) for r in range(world)]
        for p in procs:
            p.start()
        for _ in procs:
            ready.get(timeout=600)                       # every rank has imported its modules
        go.set()
        recs = [out.get(timeout=600) for _ in procs]
        for p in procs:
            p.join()
        if shm is not None:
            shm.close()
            shm.unlink()
        wall = max(r["end"] for r in recs) - min(r["begin"] for r in recs)
        utts, nbytes = sum(r["utts"] for r in recs), sum(r["bytes"] for r in recs)
        slow = max(recs, key=lambda r: r["seconds"])
        rec = {"ranks": world, "reader_threads_per_rank": threads, "wall_seconds": round(wall, 4), "utts": utts, "utts_per_s": round(utts / wall, 1),
               "GB_per_s": round(nbytes / wall / 1e9, 2), "slowest_rank": {k: slow[k] for k in ("rank", "seconds", "index", "plan", "read", "pack")}}
        rec["slowest_rank"]["seconds"] = round(rec["slowest_rank"]["seconds"], 4)
        if best is None or rec["utts_per_s"] > best["utts_per_s"]:
            best = rec
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=50000)
    ap.add_argument("--ragged-utts", type=int, default=20000)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--threads", default="1,2,4,8")
    ap.add_argument("--dir", default="/tmp/asv_loaders")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--whole-index", action="store_true", help="every rank reads ALL headers (the path before the shared header pass; ASV_AMD_SHARD_INDEX=0 in the script)")
    args = ap.parse_args()
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    res = {"host_cores": cores, "tables": {}}
    tables = []
    if args.utts > 0:
        tables.append(("fixed200", args.utts, None))
    if args.ragged_utts > 0:
        tables.append(("ragged200_1000", args.ragged_utts, (200, 1000)))
    try:
        for name, utts, lens in tables:
            scp, size = write_table(args.dir, name, utts, lens)
            with open(os.path.join(args.dir, name + ".ark"), "rb") as f:      # into the page cache
                while f.read(1 << 26):
                    pass
            rows = []
            for world in (int(v) for v in args.ranks.split(",")):
                per = [run(scp, world, th, args.repeats, shared_index=not args.whole_index) for th in (int(v) for v in args.threads.split(","))]
                top = max(per, key=lambda r: r["utts_per_s"])
                rows.append({"ranks": world, "best_reader_threads": top["reader_threads_per_rank"], "best_utts_per_s": top["utts_per_s"], "best_GB_per_s": top["GB_per_s"],
                             "by_threads": {str(r["reader_threads_per_rank"]): r["utts_per_s"] for r in per}, "slowest_rank_at_best": top["slowest_rank"]})
                print("%-15s ranks %d: best %9.1f utt/s (%.2f GB/s) with %d reader threads per rank; by threads %s" % (
                    name, world, top["utts_per_s"], top["GB_per_s"], top["reader_threads_per_rank"], rows[-1]["by_threads"]), file=sys.stderr, flush=True)
            res["tables"][name] = {"utterances": utts, "ark_GB": round(size / 1e9, 2), "rows": rows}
    finally:
        for name, _, _ in tables:
            for ext in (".ark", ".scp"):
                try:
                    os.remove(os.path.join(args.dir, name + ext))
                except OSError:
                    pass
        try:
            os.rmdir(args.dir)
        except OSError:
            pass
    print(json.dumps(res))


if __name__ == "__main__":
    main()
