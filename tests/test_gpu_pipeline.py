"""ark -> ark: the drop-in extraction script with the reference's command line, on the GPU."""

import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_extract_embeddings_script_ark_to_ark(tmp_path):
    import torch
    from libs.support import kaldi_io
    import libs.support.utils as utils
    g, sd = helpers.golden_state_dict("xvector_near_ragged")
    mats = helpers.golden_feats(g)
    keys = ["utt%03d" % i for i in range(len(mats))]
    feats_ark = tmp_path / "feats.ark"
    with open(feats_ark, "wb") as f:
        for k, m in zip(keys, mats):
            kaldi_io.write_mat(f, m, key=k)
    params = tmp_path / "final.params"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(params))
    cfg = tmp_path / "nnet.config"
    utils.write_nnet_config(os.path.join(helpers.MODEL_DIR, "xvector.py"), str(g["creation"]), str(cfg))
    out_ark = tmp_path / "xvector.ark"
    script = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")
    env = dict(os.environ, ASV_AMD_PRECISION="f32")
    res = subprocess.run([sys.executable, script, "--nnet-config", str(cfg), "--use-gpu", "true", "--gpu-id", "0", "--batch-frames", "700",
                          str(params), "ark:cat %s |" % feats_ark, "ark:| cat > %s" % out_ark], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Error" not in res.stdout + res.stderr
    import time
    for _ in range(100):
        if out_ark.exists() and out_ark.stat().st_size >= len(keys) * (512 * 4 + 10):
            break
        time.sleep(0.05)
    got = list(kaldi_io.read_vec_flt_ark(str(out_ark)))
    assert [k for k, _ in got] == keys
    for (k, v), ref in zip(got, g["embeddings"]):
        assert v.dtype == np.float32 and v.shape == (512,)
        assert rel_err(v, ref) < 1e-4, k


def test_script_fails_loudly(tmp_path):
    script = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")
    res = subprocess.run([sys.executable, script, "--model-blueprint", "/nonexistent.py", "--model-creation", "X()", "nomodel", "ark:/dev/null", "ark:/dev/null"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 1 and "Error" in res.stderr
