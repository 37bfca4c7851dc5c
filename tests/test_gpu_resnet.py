"""ResNet34(-SE) 2-D trunk + per-bin statistics pooling on the MI355X (BASELINE config C5
extractor) against the reference's own outputs (tests/golden/resnet34*.npz)."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32", "f32x"])
@pytest.mark.parametrize("name", ["resnet34se_c5", "resnet34_plain", "resnet34_cmvn", "resnet34_preact", "resnet34se_preact", "resnet_bottleneck_se",
                                  "resnet_bottleneck_preact"])
def test_resnet_f32_vs_reference_golden(name, precision):
    """incl. the blueprint's default configuration: full pre-activation blocks (resnet34_preact = ResNetXvector(80, 10, training=False))"""
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = precision
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    assert got.shape == g["embeddings"].shape
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < 1e-4, "%s: utterance of %d frames" % (name, T)


def test_resnet_bf16_is_close_and_batch_invariant():
    from libs.amd import synth
    g, sd, model = helpers.golden_model("resnet34se_c5")
    model.cuda()
    model.amd_precision = "bf16"
    mats = helpers.golden_feats(g)
    got = model.extract_embedding_batch(mats).numpy()
    ref = g["embeddings"]
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() > 0.998, cos
    # variable-length batch (config C5: T in [200, 1000]) - every utterance equals its stand-alone extraction
    lens = [200, 333, 1000, 201]
    mats = [synth.synth_feats(T, 80, 9000 + i) for i, T in enumerate(lens)]
    full = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(full).all()
    for i in (1, 2):
        assert np.array_equal(model.extract_embedding(mats[i]).numpy(), full[i])


def test_narrow_grid_conv_kernel_agrees_with_generic_gemm_tile(monkeypatch):
    """The C = 32 / 64 stages run on kernels_conv2d.hip in bf16 mode; ASV_AMD_SMALL_TILES=1 sends them back to the
    generic 128 x 128 implicit-GEMM kernel.  Both walk K as (tap, 16-channel group) with the same MFMA instruction,
    so the f32 accumulation order - and therefore every bit of the embeddings - is the same."""
    from libs.amd import synth
    g, sd, model = helpers.golden_model("resnet34se_c5")
    model.cuda()
    model.amd_precision = "bf16"
    mats = [synth.synth_feats(T, 80, 7000 + i) for i, T in enumerate([200, 257, 640])]
    fast = model.extract_embedding_batch(mats).numpy()
    monkeypatch.setenv("ASV_AMD_SMALL_TILES", "1")
    slow = model.extract_embedding_batch(mats).numpy()                     # new flags = new engine (framework.py caches per flag set)
    assert np.isfinite(fast).all() and np.array_equal(fast, slow)


def test_resnet_preactivation_bf16_is_close():
    g, sd, model = helpers.golden_model("resnet34_preact")
    model.cuda()
    model.amd_precision = "bf16"
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    ref = g["embeddings"]
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() > 0.998, cos


@pytest.mark.parametrize("precision", ["bf16", "f32x", "f32"])
def test_gather_with_elementwise_prologue_changes_no_bit(precision, monkeypatch):
    """Round 4: the pass that closes a ResNet stage - relu(y * se + identity), resnet.py:70-85 - is folded into the stride-2 gathers
    that are its only readers (Graph.fused_gather_ops -> asv_im2col_desc_t.b_buf / seg_scale_buf / act).  The prologue repeats the
    elementwise kernel's operations and roundings, so with ASV_AMD_NO_GATHER_FUSE=1 (two passes) every bit of the embeddings is the
    same, in every precision mode; ragged batch, SE and plain blocks."""
    from libs.amd import synth
    for name in ("resnet34se_c5", "resnet34_plain"):
        g, sd, model = helpers.golden_model(name)
        model.cuda()
        model.amd_precision = precision
        mats = [synth.synth_feats(T, int(g["dim"]), 4100 + i) for i, T in enumerate([200, 333, 517, 201])]
        monkeypatch.delenv("ASV_AMD_NO_GATHER_FUSE", raising=False)
        fused = model.extract_embedding_batch(mats).numpy()
        n_fused = len(model._amd_engine().ops)
        monkeypatch.setenv("ASV_AMD_NO_GATHER_FUSE", "1")
        two = model.extract_embedding_batch(mats).numpy()
        assert len(model._amd_engine().ops) > n_fused or name == "resnet34_plain"
        assert np.isfinite(fused).all() and np.array_equal(fused, two), (name, precision, float(np.abs(fused - two).max()))


@pytest.mark.parametrize("precision", ["f32", "f32x"])
@pytest.mark.parametrize("name", ["resnet_attentive", "resnet_multihead", "resnet_multires", "resnet_lde"])
def test_resnet_with_frame_weighting_poolings_vs_reference_golden(name, precision):
    """Round 4: ResNetXvector(pooling='attentive' | 'multi-head' | 'multi-resolution' | 'lde') - resnet_xvector.py:104-111 - on the
    device: asv_net_add_grid_flatten writes the trunk's output as [T'][c*F' + f] rows on a sequence domain (a width-1 grid at T / 8),
    the attention layers and attentive_pool / lde_pool kernels run there.  Reference outputs, 1e-4, in the exact and the default mode."""
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = precision
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    assert got.shape == g["embeddings"].shape
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < 1e-4, "%s: utterance of %d frames" % (name, T)
    assert any(op.kind == "flatten" for op in model._amd_engine().ops)


def test_resnet_attentive_pooling_16bit_modes_are_close_and_batch_invariant():
    from libs.amd import synth
    g, sd, model = helpers.golden_model("resnet_attentive")
    model.cuda()
    ref = g["embeddings"]
    for precision, floor in (("bf16", 0.995), ("f16", 0.9995)):
        model.amd_precision = precision
        got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
        cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
        assert cos.min() > floor, (precision, cos)
        mats = [synth.synth_feats(T, 40, 9100 + i) for i, T in enumerate([200, 77, 333])]
        full = model.extract_embedding_batch(mats).numpy()
        assert np.array_equal(model.extract_embedding(mats[1]).numpy(), full[1])
