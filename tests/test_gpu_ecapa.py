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


def test_ecapa_f32x_8phase_kernel_same_bits_and_range_status(monkeypatch):
    """Round 5: the f32x mode's wide plain layers (ECAPA's 1-tap C -> C layers, the 3C -> 1536 layer) go to kernels_tdnn_p8x.hip from
    one round of 256 x 256 tiles on; forced onto this small batch (ASV_AMD_P8X=2) the embeddings must be the bits tdnn_gemm_x3_kernel
    gives (the dispatch depends on the batch size: an utterance's embedding may not), inside the f32 gate against the reference's own
    outputs, and the range watch of the IEEE-half split - on the accumulators in this kernel - must raise the status word the
    scripts' guard reads."""
    import torch
    from libs.amd import capi, synth
    L = capi.lib()
    g, sd, model = helpers.golden_model("ecapa_c512_near_affine")
    model.cuda()
    model.amd_precision = "f32x"
    feats = helpers.golden_feats(g)
    mats = [synth.synth_feats(T, 80, 7100 + i) for i, T in enumerate([300, 211, 300, 64, 500, 300, 2, 129])]
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_P8X", "0")
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X)
    ref_g = model.extract_embedding_batch(feats).numpy()
    ref = model.extract_embedding_batch(mats).numpy()
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X) == n0
    monkeypatch.setenv("ASV_AMD_P8X", "2")
    got_g = model.extract_embedding_batch(feats).numpy()
    got = model.extract_embedding_batch(mats).numpy()
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X) > n0, "no layer went to the f32x 8-phase kernel"
    assert np.array_equal(got, ref) and np.array_equal(got_g, ref_g)
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got_g[i], g["embeddings"][i]) < TOL_F32
    eng = model._amd_engine()
    assert eng.status() == 0
    huge = [m.copy() for m in mats]
    huge[4] = (huge[4] * 3.0e5).astype(np.float32)
    dev = torch.device("cuda", eng.device_index)
    offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in huge])]).astype(np.int32)
    for p8x in ("2", "0"):
        monkeypatch.setenv("ASV_AMD_P8X", p8x)
        eng.extract_device(torch.from_numpy(np.concatenate(huge)).to(dev), offs)
        assert eng.status() & capi.STATUS_HALF_RANGE, p8x
        eng.extract_device(torch.from_numpy(np.concatenate(mats)).to(dev), offs)
        assert eng.status() == 0, p8x


def test_res2_block_kernel_matches_per_branch_layers(monkeypatch):
    """bf16 mode runs every Res2NetBlock (ecapa_tdnn_xvector.py:61-75) as one kernel with the running tensor in LDS
    (kernels_res2.hip); ASV_AMD_NO_FUSE=1 keeps one launch per branch.  Same bf16 operands and the same bf16 rounding of every
    intermediate: the embeddings agree to the f32 summation order - on ragged batches with tiny utterances, at all three
    dilations - and both agree with the reference."""
    from libs.amd import synth
    g, sd, model = helpers.golden_model("ecapa_launcher")
    model.cuda()
    model.amd_precision = "bf16"
    mats = helpers.golden_feats(g) + [synth.synth_feats(T, 80, 7100 + i) for i, T in enumerate([1, 2, 7, 33, 129, 300, 517, 64, 300])]
    fused = model.extract_embedding_batch(mats).numpy()
    assert "res2" in model._amd_engine().describe()
    monkeypatch.setenv("ASV_AMD_NO_FUSE", "1")
    plain = model.extract_embedding_batch(mats).numpy()
    assert "res2" not in model._amd_engine().describe()
    assert np.isfinite(fused).all()
    cos = (fused * plain).sum(1) / np.linalg.norm(fused, axis=1) / np.linalg.norm(plain, axis=1)
    assert cos.min() > 0.9999 and rel_err(fused, plain) < 1e-2, (cos.min(), rel_err(fused, plain))
    ref = g["embeddings"]
    n = len(ref)
    cos_ref = (fused[:n] * ref).sum(1) / np.linalg.norm(fused[:n], axis=1) / np.linalg.norm(ref, axis=1)
    assert cos_ref.min() > 0.999, cos_ref


@pytest.mark.parametrize("precision", ["bf16", "f16"])
def test_res2_block_kernel_window_forms_give_the_same_bits(precision, monkeypatch):
    """Round 5: res2_chain_kernel has two window sizes (6 / 7 row fragments: 128 / 160 output rows per workgroup, 4 + 3 fragments on
    the two row halves of the larger one, images B and X in one buffer); the launcher picks by the batch's tile count.  Every output row
    is computed from the same operands in the same order in both, so the embeddings must be bit-identical - on ragged batches whose
    row counts leave the last 160-row tile overhanging the matrix by different amounts (its loads clamp onto the last gap row, its
    stores write that row's zeros), at all three dilations of the model."""
    from libs.amd import synth
    g, sd, model = helpers.golden_model("ecapa_launcher")
    model.cuda()
    model.amd_precision = precision
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    lens_sets = [[300, 211, 300, 64, 500, 300, 2, 129], [1, 2, 7, 33], [200] * 9, [517, 640, 3, 300, 300, 300, 41], [160], [161, 159, 160, 1]]
    for k, lens in enumerate(lens_sets):
        mats = [synth.synth_feats(T, 80, 7300 + 17 * k + i) for i, T in enumerate(lens)]
        monkeypatch.setenv("ASV_AMD_RES2_FR", "6")
        a = model.extract_embedding_batch(mats).numpy()
        monkeypatch.setenv("ASV_AMD_RES2_FR", "7")
        b = model.extract_embedding_batch(mats).numpy()
        assert "res2" in model._amd_engine().describe()
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (precision, lens, int((a != b).sum()))
    monkeypatch.delenv("ASV_AMD_RES2_FR")
    c = model.extract_embedding_batch(mats).numpy()              # the launcher's own choice
    assert np.array_equal(a, c)


@pytest.mark.parametrize("precision", ["f32", "f32x"])
def test_ecapa_with_attentive_statistics_pooling_vs_reference_golden(precision):
    """Round 4: ECAPA_TDNN(pooling='attentive') - AttentiveStatisticsPooling with time context behind the MFA layer
    (ecapa_tdnn_xvector.py:275-281), the one further pooling option the reference's ECAPA constructor can build - against the
    reference's outputs (measured: 3.6e-6 in f32, 4.5e-6 in f32x, profiles/r4r_ecapa_attentive.txt)."""
    g, sd, model = helpers.golden_model("ecapa_attentive")
    model.cuda()
    model.amd_precision = precision
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < 1e-4, "utterance of %d frames" % T
