"""Script-level multi-GPU path on CPU (world_size 2, gloo): the --sharded mode of pipeline/onestep/extract_embeddings.py
(scp -> length-balanced shards -> extract -> one all-gather -> rank 0 writes the ark in scp order) and the configs[3]
stand-in (tests/c4_standin.py) with a numpy stand-in for the engine - everything but the kernels."""

import importlib.util
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers

SCRIPT = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")


def _load_script():
    spec = importlib.util.spec_from_file_location("extract_embeddings_script", SCRIPT)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_embedding(mat):
    v = np.zeros(12, dtype=np.float32)
    v[:8] = mat.mean(axis=0)
    v[-1] = mat.shape[0]
    return v


def _worker(rank, world, port, scp, out_ark, use_table, segment=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _load_script()
        entries = m.read_scp("scp:" + scp)
        if use_table:
            table = dict(line.split() for line in open(scp.replace(".scp", ".utt2num_frames")))
            lengths = np.array([int(table[k]) for k, _ in entries])
        else:
            lengths = np.array([m.matrix_rows(rx) for _, rx in entries])
        extract = lambda mats: torch.from_numpy(np.stack([_fake_embedding(x) for x in mats]))
        w = open(out_ark, "wb") if rank == 0 else None
        n = m.extract_sharded_scp(extract, entries, lengths, w, batch_frames=900, batch_utts=5, segment_utts=segment)
        if w is not None:
            w.close()
        assert n == len(entries)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_table", [False, True])
def test_sharded_script_writes_all_vectors_in_scp_order(tmp_path, use_table):
    import torch.multiprocessing as mp
    from libs.support import kaldi_io
    from libs.amd import synth
    lens = np.random.RandomState(3).randint(10, 300, size=23)
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    mats = {}
    with open(ark, "wb") as f, open(scp, "w") as s, open(scp.replace(".scp", ".utt2num_frames"), "w") as u:
        for i, n in enumerate(lens):
            key = "utt%02d" % i
            mats[key] = synth.synth_feats(int(n), 8, 300 + i)
            f.write((key + " ").encode())
            s.write("%s %s:%d\n" % (key, ark, f.tell()))
            u.write("%s %d\n" % (key, n))
            if i % 3 == 0:
                kaldi_io.write_mat(f, mats[key].astype(np.float64))          # a few float64 entries
            else:
                kaldi_io.write_mat(f, mats[key])
    out = str(tmp_path / "xvector.ark")
    mp.spawn(_worker, args=(2, _free_port(), scp, out, use_table), nprocs=2, join=True)
    got = list(kaldi_io.read_vec_flt_ark(out))
    assert [k for k, _ in got] == ["utt%02d" % i for i in range(len(lens))]
    for k, v in got:
        assert np.array_equal(v, _fake_embedding(mats[k])), k


@pytest.mark.parametrize("world,segment", [(2, 3), (1, 4), (3, 1), (2, 0)])
def test_sharded_script_in_segments_writes_the_same_ark(tmp_path, world, segment):
    """Round 5: the sharded path gathers and writes in segments of `segment` utterances per rank (the writer thread of rank 0 works
    on segment s while segment s + 1 is extracted); whatever the segment size - 1 per rank, a size that leaves a ragged last segment,
    0 = one gather at the end - the ark holds every vector once, in scp order."""
    import torch.multiprocessing as mp
    from libs.support import kaldi_io
    from libs.amd import synth
    lens = np.random.RandomState(11).randint(10, 300, size=29)
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    mats = {}
    with open(ark, "wb") as f, open(scp, "w") as s:
        for i, n in enumerate(lens):
            key = "utt%02d" % i
            mats[key] = synth.synth_feats(int(n), 8, 900 + i)
            f.write((key + " ").encode())
            s.write("%s %s:%d\n" % (key, ark, f.tell()))
            kaldi_io.write_mat(f, mats[key])
    out = str(tmp_path / "xvector.ark")
    mp.spawn(_worker, args=(world, _free_port(), scp, out, False, segment), nprocs=world, join=True)
    got = list(kaldi_io.read_vec_flt_ark(out))
    assert [k for k, _ in got] == ["utt%02d" % i for i in range(len(lens))]
    for k, v in got:
        assert np.array_equal(v, _fake_embedding(mats[k])), k


def test_script_rejects_sharded_mode_without_an_scp():
    m = _load_script()
    args = m.get_args(["--sharded", "true", "model.params", "ark:feats.ark", "ark:out.ark"])
    with pytest.raises(ValueError):
        m.run_sharded(args, None, None, False)


def test_c4_standin_control_flow_under_gloo_world2():
    """tests/c4_standin.py end to end with the numpy stand-in extractor on two gloo ranks: the sharded run equals the
    single-process run (same embeddings in original order => same EER), i.e. shard + all-gather + scoring are transparent."""
    script = os.path.join(helpers.REPO, "tests", "c4_standin.py")
    base = [sys.executable, script, "--fake-extractor", "--utts", "96", "--per-spk", "4", "--t-lo", "20", "--t-hi", "60", "--trials", "2000", "--noise", "1.5"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    one = subprocess.run(base, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    two = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode == 0, two.stdout + two.stderr
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2
    assert r1["eer_percent"] == r2["eer_percent"] and 0.0 < r1["eer_percent"] < 50.0
    assert r2["eer_delta_percent"] == 0.0


def test_c5_standin_control_flow_under_gloo_world2():
    """tests/c5_standin.py (BASELINE configs[4]: variable-length extraction, one all-gather, PLDA EM + LLR + EER on rank 0) with the
    numpy stand-ins on two gloo ranks: the sharded run equals the single-process run."""
    script = os.path.join(helpers.REPO, "tests", "c5_standin.py")
    base = [sys.executable, script, "--fake-extractor", "--utts", "120", "--per-spk", "4", "--t-lo", "20", "--t-hi", "100", "--trials", "1500", "--noise", "1.5",
            "--plda-iters", "3"]
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    one = subprocess.run(base, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    two = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode == 0, two.stdout + two.stderr
    r1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    r2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert r1["n_gpus"] == 1 and r2["n_gpus"] == 2 and r1["frames"] == r2["frames"]
    assert r1["eer_percent"] == r2["eer_percent"] and 0.0 <= r1["eer_percent"] < 50.0
    assert r2["eer_delta_percent"] == 0.0 and r2["max_abs_llr_delta"] == 0.0
