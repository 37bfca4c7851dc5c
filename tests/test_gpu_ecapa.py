"""ECAPA-TDNN (SE-Res2Blocks, attentive statistics pooling) on the MI355X against the
reference's own outputs (tests/golden/ecapa_*.npz)."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4


def _run(name, precision):
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = precision
    return g, model, model.extract_embedding_batch(helpers.golden_feats(g)).numpy()


@pytest.mark.parametrize("name", ["ecapa_c3", "ecapa_launcher", "ecapa_c512_fc1_far", "ecapa_c512_near_affine"])
def test_ecapa_f32_vs_reference_golden(name):
    g, model, got = _run(name, "f32")
    assert got.shape == g["embeddings"].shape
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < TOL_F32, "%s: utterance of %d frames" % (name, T)


def test_ecapa_bf16_is_close():
    g, model, got = _run("ecapa_launcher", "bf16")
    ref = g["embeddings"]
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() > 0.999, cos


def test_ecapa_extract_embedding_whole_positions():
    g, sd, model = helpers.golden_model("ecapa_c3")
    model.cuda()
    model.amd_precision = "f32"
    x = helpers.golden_feats(g)[0]
    near = model.extract_embedding_whole(x, position="near", maxChunk=4000)
    assert rel_err(near.numpy(), g["embeddings"][0]) < TOL_F32
    assert model.embedding_dim() == 192
    # maxChunk smaller than T exercises the chunk-and-average rule on ECAPA
    from oracle import np_oracle as O
    want = O.extract_embedding(lambda c: O.ecapa_embed(c, sd, "near"), x, max_chunk=128)
    got = model.extract_embedding_whole(x, position="near", maxChunk=128)
    assert rel_err(got.numpy(), want) < TOL_F32


def test_ecapa_batch_composition_invariance():
    g, sd, model = helpers.golden_model("ecapa_c512_near_affine")
    model.cuda()
    model.amd_precision = "bf16"
    from libs.amd import synth
    mats = [synth.synth_feats(T, 80, 7000 + i) for i, T in enumerate([300, 211, 300, 64, 500, 300, 2, 129])]
    full = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(full).all()
    for i in (0, 3, 6):
        assert np.array_equal(model.extract_embedding(mats[i]).numpy(), full[i])
