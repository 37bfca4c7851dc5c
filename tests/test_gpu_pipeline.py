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


def test_online_script_wav_scp_to_ark(tmp_path):
    """wav.scp + feat yaml -> embeddings: the script output equals front-end + extractor called directly, and the oracle
    (numpy fbank restatement -> CMN -> numpy x-vector) within the f32 bar."""
    import wave
    import torch
    import yaml
    from libs.amd import frontend, synth
    from libs.support import kaldi_io
    import libs.support.utils as utils
    from oracle import fbank_oracle, np_oracle as O
    g, sd = helpers.golden_state_dict("xvector_c1")                # Xvector(30, ...)
    lens = [16000, 48000, 300, 24000, 20011]                       # the third is shorter than one 25 ms window: skipped with a warning
    waves = [synth.synth_wave(n, 800 + i).astype(np.int16) for i, n in enumerate(lens)]
    scp = tmp_path / "wav.scp"
    with open(scp, "w") as f:
        for i, wv in enumerate(waves):
            path = tmp_path / ("u%d.wav" % i)
            with wave.open(str(path), "wb") as wf:
                wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
                wf.writeframes(wv.tobytes())
            f.write("utt%d %s\n" % (i, path))
    featset = {"num_mel_bins": 30, "dither": 0.0, "energy_floor": 0.0}
    conf = tmp_path / "feat.yaml"
    conf.write_text(yaml.safe_dump({"feature_type": "fbank", "kaldi_featset": featset, "mean_var_conf": {"mean_norm": True, "std_norm": False}}))
    params = tmp_path / "final.params"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(params))
    cfg = tmp_path / "nnet.config"
    utils.write_nnet_config(os.path.join(helpers.MODEL_DIR, "xvector.py"), str(g["creation"]), str(cfg))
    out_ark = tmp_path / "xvector.ark"
    script = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings_online.py")
    env = dict(os.environ, ASV_AMD_PRECISION="f32")
    res = subprocess.run([sys.executable, script, "--nnet-config", str(cfg), "--data-type", "raw", "--feat-config", str(conf), "--gpu-id", "0",
                          "--batch-utts", "2", str(params), str(scp), str(out_ark)], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "RTF:" in res.stdout and "utt2 is shorter than one frame" in res.stdout
    got = list(kaldi_io.read_vec_flt_ark(str(out_ark)))
    assert [k for k, _ in got] == ["utt0", "utt1", "utt3", "utt4"]
    model = helpers.build_model("xvector.py", str(g["creation"]), sd)
    model.cuda()
    model.amd_precision = "f32"
    kept = [waves[i] for i in (0, 1, 3, 4)]
    direct = model.extract_embedding_batch(frontend.fbank(kept, mean_norm=True, **featset)).numpy()
    for (k, v), d, wv in zip(got, direct, kept):
        assert np.array_equal(v, d), k
        feat = fbank_oracle.fbank(wv.astype(np.float32), num_bins=30)
        feat = feat - feat.mean(0)
        want = O.extract_embedding(lambda c: O.xvector_embed(c, sd, "far"), feat)
        assert rel_err(v, want) < 1e-4, k


def test_extract_embeddings_script_sharded_mode_matches_stream_mode(tmp_path):
    """--sharded true (here: one rank, no launcher; the world-2 control flow runs under gloo in tests/test_sharded_script_gloo.py):
    scp in, length-balanced batches, vectors in scp order - the same embeddings as the streaming mode and the reference."""
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
    res = subprocess.run([sys.executable, script, "--nnet-config", str(cfg), "--use-gpu", "true", "--sharded", "true", "--batch-frames", "700",
                          str(params), "scp:%s" % feats_scp, "ark:%s" % out_ark], capture_output=True, text=True, timeout=600)     # default precision (f32x)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Extracted %d embeddings." % len(keys) in res.stdout
    got = list(kaldi_io.read_vec_flt_ark(str(out_ark)))
    assert [k for k, _ in got] == keys
    for (k, v), ref in zip(got, g["embeddings"]):
        assert rel_err(v, ref) < 1e-4, k


@pytest.mark.parametrize("mode", ["ark", "scp", "sharded"])
def test_script_range_guard_huge_features(tmp_path, mode):
    """VERDICT r4 / ADVICE r4: the range guard of the default (f32x) mode through the SCRIPT.  Features scaled by 1e5 in some
    utterances drive activations past the IEEE-half range of the operand split; the reference's f32 forward
    (/root/reference/pytorch/pipeline/onestep/extract_embeddings.py:70-83) has no such limit.  Every entry point - the ark stream, the
    `scp:` stream and `--sharded true` - must notice (status word behind every batch, libs.amd.pipeline.DeviceSets), re-run the flagged
    batch with bf16 halves and write embeddings within 1e-4 of the numpy oracle, with the RuntimeWarning in the log."""
    import torch
    from libs.support import kaldi_io
    import libs.support.utils as utils
    from oracle import np_oracle as O
    g, sd = helpers.golden_state_dict("xvector_near_ragged")
    mats = [m.copy() for m in helpers.golden_feats(g)]
    for i in (1, 4, len(mats) - 1):                                  # in different batches (--batch-frames 700)
        mats[i] = (mats[i] * 1.0e5).astype(np.float32)
    want = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in mats])
    assert np.isfinite(want).all()
    keys = ["utt%03d" % i for i in range(len(mats))]
    feats_ark, feats_scp = tmp_path / "feats.ark", tmp_path / "feats.scp"
    with open(feats_ark, "wb") as f, open(feats_scp, "w") as sc:
        for k, m in zip(keys, mats):
            f.write((k + " ").encode())
            sc.write("%s %s:%d\n" % (k, feats_ark, f.tell()))
            kaldi_io.write_mat(f, m)
    params = tmp_path / "final.params"
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(params))
    cfg = tmp_path / "nnet.config"
    utils.write_nnet_config(os.path.join(helpers.MODEL_DIR, "xvector.py"), str(g["creation"]), str(cfg))
    out_ark = tmp_path / "xvector.ark"
    script = os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")
    extra = ["--sharded", "true"] if mode == "sharded" else []
    rspec = "ark:%s" % feats_ark if mode == "ark" else "scp:%s" % feats_scp
    env = {k: v for k, v in os.environ.items() if k != "ASV_AMD_PRECISION"}          # the default mode (f32m since round 6; the same range guard and twin as f32x)
    res = subprocess.run([sys.executable, "-W", "always", script, "--nnet-config", str(cfg), "--use-gpu", "true", "--gpu-id", "0", "--batch-frames", "700"] + extra +
                         [str(params), rspec, "ark:%s" % out_ark], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "RuntimeWarning" in res.stderr and "f32x-bf16" in res.stderr, res.stderr[-2000:]
    got = list(kaldi_io.read_vec_flt_ark(str(out_ark)))
    assert [k for k, _ in got] == keys
    for (k, v), ref in zip(got, want):
        assert np.isfinite(v).all() and rel_err(v, ref) < 1e-4, (mode, k, rel_err(v, ref))


def test_device_sets_guard_and_plain_batches():
    """libs.amd.pipeline.DeviceSets directly: results on the host and on the device, a clean batch (no warning, no re-run) next to a
    flagged one (re-run on the twin, in place), the oversize-utterance path, and Engine.extract_device_guarded (the online script's
    call)."""
    import warnings
    import torch
    from libs.amd.pipeline import DeviceSets
    from oracle import np_oracle as O
    g, sd, model = helpers.golden_model("xvector_near_ragged")
    model.cuda()
    model.amd_precision = "f32x"
    mats = [m.copy() for m in helpers.golden_feats(g)][:6]
    huge = [m.copy() for m in mats]
    huge[3] = (huge[3] * 1.0e5).astype(np.float32)
    want_clean = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in mats])
    want_huge = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in huge])
    dim = mats[0].shape[1]
    rows = sum(m.shape[0] for m in mats)
    offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int32)
    for results in ("host", "device"):
        sets = DeviceSets(model, rows + 8, 16, dim, 10000, n_sets=2, results=results)
        assert sets.watch
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sets.host_buffer(0)[:rows] = np.concatenate(mats)
            sets.submit(0, offs, rows)
            sets.host_buffer(1)[:rows] = np.concatenate(huge)
            sets.submit(1, offs, rows)
            sets.input_consumed(0)
            a = sets.finish(0)
            a = a.copy() if results == "host" else a.cpu().numpy()
            assert not w and sets.range_reruns == 0
            b = sets.finish(1)
            b = b.copy() if results == "host" else b.cpu().numpy()
            assert sets.range_reruns == 1 and any("f32x-bf16" in str(x.message) for x in w)
            big = np.concatenate(mats)                               # an "utterance longer than the buffer": the ndarray path
            sets.submit(0, offs, big)
            c = sets.finish(0)
            c = c.copy() if results == "host" else c.cpu().numpy()
            sets.flush()
        assert rel_err(a, want_clean) < 1e-4 and rel_err(c, want_clean) < 1e-4
        for i in range(len(huge)):
            assert rel_err(b[i], want_huge[i]) < 1e-4, (results, i)
    eng = model._amd_engine()
    dev = torch.device("cuda", eng.device_index)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = eng.extract_device_guarded(torch.from_numpy(np.concatenate(huge)).to(dev), offs).cpu().numpy()
    assert any("f32x-bf16" in str(x.message) for x in w)
    for i in range(len(huge)):
        assert rel_err(out[i], want_huge[i]) < 1e-4, i


def test_c4_standin_full_size_single_gpu():
    """BASELINE configs[3] at SURVEY 8(d) size on one GPU (the 8-GPU run is the same script under torch.distributed.run):
    4 708 ECAPA-TDNN utterances of 400..1500 frames, 37 720 trials; the parity-grade f32x mode meets the north-star EER gate
    against reference-equivalent (f32, oracle-checked) embeddings; the bf16 throughput mode is reported and bounded."""
    import json
    script = os.path.join(helpers.REPO, "tests", "c4_standin.py")

    def run(prec, noise, checks):
        res = subprocess.run([sys.executable, script, "--precision", prec, "--noise", str(noise), "--oracle-checks", str(checks)], capture_output=True, text=True, timeout=1500)
        assert res.returncode == 0, res.stdout + res.stderr
        rec = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        print(rec)
        return rec

    # With 18 860 target trials one trial that changes side moves the EER by 0.0053 %: a single stand-in set resolves the
    # 0.01 % gate to one trial.  Round 2's f32x mode (bf16 hi + lo halves, ~6e-6 relative on the embeddings) sat AT that gate
    # here: at the 11-17 % EER of planted speakers behind random weights ~75 trials lie within its score error (<= 2e-3 after
    # the mean subtraction, which removes the large common component of these embeddings) of the threshold - measured 0.0000 /
    # 0.0053 / 0.0000 / 0.0212 % over four sets, so the gate was held on the mean of three.  With IEEE-half halves (22
    # significant bits per operand: f32-grade products) the mode is inside the gate on EVERY set, with score deltas an order
    # smaller; the bf16 split is still measured on one set for the record.
    f32x = [run("f32x", nz, 2 if i == 0 else 0) for i, nz in enumerate((0.05, 0.1, 0.3))]
    assert f32x[0]["oracle_max_rel_err_f32"] < 1e-4
    for rec in f32x:
        assert 0.5 < rec["eer_reference_equivalent_percent"] < 40.0
        assert abs(rec["eer_delta_percent"]) < 0.01, rec
        assert rec["max_abs_score_delta"] < 5e-4, rec
    old_split = run("f32x-bf16", 0.3, 0)
    assert abs(old_split["eer_delta_percent"]) < 0.03 and old_split["max_abs_score_delta"] < 5e-3, old_split
    # 16-bit ECAPA embeddings sit at 1.5-2 % (bf16) / ~0.2 % (f16) relative error from the f32 ones whatever the length (tools/
    # ecapa_precision_probe.py).  With synthetic weights the embeddings share a large common component that the scoring chain
    # subtracts (sub-mean), so that error is a large share of what is left: cosine scores move by up to 0.3 and the EER by
    # 0.1-0.5 % abs in bf16 on this stand-in (0.39 % at noise 0.1).  Reported by the script; bounded here only against gross regressions.
    bf16 = run("bf16", 0.1, 0)
    assert abs(bf16["eer_delta_percent"]) < 1.0, bf16
    f16 = run("f16", 0.1, 0)
    assert abs(f16["eer_delta_percent"]) < 0.25 and f16["max_abs_score_delta"] < 0.25 * bf16["max_abs_score_delta"] + 1e-3, (f16, bf16)

def test_c5_standin_stated_shape_single_gpu():
    """BASELINE configs[4] at its stated shape on one GPU (the 8-GPU run is the same script under torch.distributed.run):
    ResNet34-SE on 2 000 packed-ragged utterances of 200..1000 frames, PLDA trained with 10 EM iterations ON the extracted
    embeddings (plda_base.py:248-300), LLR scoring, EER.  The parity-grade f32x mode meets the north-star gates against the
    reference-equivalent chain (exact-f32 extraction, oracle-checked incl. the longest utterance); the 16-bit throughput
    modes are reported and bounded."""
    import json
    script = os.path.join(helpers.REPO, "tests", "c5_standin.py")

    def run(prec, checks):
        res = subprocess.run([sys.executable, script, "--precision", prec, "--oracle-checks", str(checks)], capture_output=True, text=True, timeout=1500)
        assert res.returncode == 0, res.stdout + res.stderr
        rec = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        print(rec)
        return rec

    r = run("f32x", 3)
    assert r["frames"] > 1_000_000 and 0.5 < r["eer_reference_equivalent_percent"] < 45.0, r
    assert r["oracle_max_rel_err_f32"] < 1e-4 and r["oracle_max_abs_llr_err_200_trials"] < 2e-3, r
    assert r["embedding_max_rel_err_vs_f32"] < 1e-4, r
    assert abs(r["eer_delta_percent"]) < 0.01, r
    for prec, bound in (("f16", 0.5), ("bf16", 2.0)):
        h = run(prec, 0)
        assert abs(h["eer_delta_percent"]) < bound, h
