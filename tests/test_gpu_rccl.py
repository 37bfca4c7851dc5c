"""RCCL on the device, at world size 1 (the GPU box has one GPU; the N > 1 curve is the driver's): the collection step that
replaces `cat xvector.JOB.scp` of /root/reference/pytorch/pipeline/extract_xvectors_for_pytorch.sh:125-151 -
`all_gather_into_tensor` of the ranks' embeddings - runs through librccl whenever a process group exists, whatever its size:

  * in this process: init_process_group("nccl") on cuda:0, libs.amd.shard.extract_sharded on real extractor output; what the
    gather delivers equals the local embeddings bit for bit, and librccl is mapped into the process;
  * `extract_embeddings.py --sharded true` and `bench.py --gpus 1` under `python -m torch.distributed.run --nproc-per-node 1`.

The world-2 / world-3 control flow (balanced shards, empty shards, a failing rank) runs under gloo in tests/test_shard_gloo.py,
tests/test_sharded_script_gloo.py and tests/test_bench_selflaunch_gloo.py."""

import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launcher(port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]


def test_all_gather_through_rccl_in_this_process():
    import torch
    import torch.distributed as dist
    from libs.amd import shard, synth
    g, sd, model = helpers.golden_model("xvector_near_ragged")
    model.cuda()
    model.amd_precision = "f32"
    mats = helpers.golden_feats(g)
    eng = model._amd_engine()
    dev = torch.device("cuda", eng.device_index)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    assert not dist.is_initialized()
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
    try:
        def extract_batch(ms):
            offs = np.zeros(len(ms) + 1, dtype=np.int32)
            np.cumsum([m.shape[0] for m in ms], out=offs[1:])
            return eng.extract_device(torch.from_numpy(np.concatenate(ms, axis=0)).to(dev), offs)
        lengths = [m.shape[0] for m in mats]
        out = shard.extract_sharded(extract_batch, lengths, lambda i: mats[i], max_frames=700, max_utts=5, device=dev)
        assert out.is_cuda and tuple(out.shape) == (len(mats), 512)
        got = out.cpu().numpy()
        for i in range(len(mats)):
            assert rel_err(got[i], g["embeddings"][i]) < 1e-4
        # the collective itself: what comes back is the local block, bit for bit (padded rows stay out)
        local = torch.arange(7 * 16, dtype=torch.float32, device=dev).reshape(7, 16) * 0.5
        back = shard.gather_embeddings(local, np.arange(7)[::-1].copy(), [np.arange(7)[::-1].copy()])
        assert torch.equal(back[torch.arange(6, -1, -1, device=dev)], local)
        t = torch.tensor([3.0], device=dev)
        dist.all_reduce(t)
        assert float(t.item()) == 3.0
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()
    with open("/proc/self/maps") as f:
        assert "librccl" in f.read(), "torch's nccl backend did not load librccl"


def test_sharded_script_under_the_launcher_collects_through_rccl(tmp_path):
    import torch
    from libs.support import kaldi_io
    import libs.support.utils as utils
    g, sd = helpers.golden_state_dict("xvector_near_ragged")
    mats = helpers.golden_feats(g)
    keys = ["utt%03d" % i for i in range(len(mats))]
    feats_ark, feats_scp = tmp_path / "feats.ark", tmp_path / "feats.scp"
    with open(feats_ark, "wb") as f, open(feats_scp, "w") as s:
        for k, m in zip(keys, mats):
            f.write((k + " ").encode())
            s.write("%s %s:%d\n" % (k, feats_ark, f.tell()))
            kaldi_io.write_mat(f, m)
    params = tmp_path / "final.params"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(params))
    cfg = tmp_path / "nnet.config"
    utils.write_nnet_config(os.path.join(helpers.MODEL_DIR, "xvector.py"), str(g["creation"]), str(cfg))
    out_ark = tmp_path / "xvector.ark"
    script = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")
    res = subprocess.run(_launcher(_free_port()) + [script, "--nnet-config", str(cfg), "--use-gpu", "true", "--sharded", "true", "--batch-frames", "700",
                                                    str(params), "scp:%s" % feats_scp, "ark:%s" % out_ark], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert "Extracted %d embeddings." % len(keys) in res.stdout
    assert "backend nccl (librccl mapped: True)" in res.stdout, res.stdout[-2000:]
    got = list(kaldi_io.read_vec_flt_ark(str(out_ark)))
    assert [k for k, _ in got] == keys
    for (k, v), ref in zip(got, g["embeddings"]):
        assert rel_err(v, ref) < 1e-4, k


def test_bench_under_the_launcher_gathers_through_rccl():
    bench = os.path.join(helpers.REPO, "bench.py")
    res = subprocess.run(_launcher(_free_port()) + [bench, "--gpus", "1", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05", "--settle-seconds", "0.1",
                                                    "--no-supplementary", "--cpu-seconds", "0", "--eer-trials", "0", "--no-profile", "--batch", "64"],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["collective"] == {"op": "all_gather_into_tensor", "backend": "nccl", "world": 1, "verified": True}, rec.get("collective")
