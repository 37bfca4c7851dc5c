"""Build-container only: option combinations that no committed fixture covers, run through the REFERENCE itself (a subprocess
of oracle/gen_golden.py --adhoc: the reference's blueprint + its own extract_embedding() on CPU) and through this repository's
recorder + graph passes + numpy interpreter.  The committed goldens pin the BASELINE configurations and every pooling; this
widens the pinned option space of the blueprints (activation / BN placement / SE / skip / positions / trunk shapes) without
adding fixtures.  Skipped where /root/reference does not exist (the GPU box)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers
import ir_interp

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pytorch", "model")), reason="needs the reference tree (build container only)")

BN_AFFINE = "'bn_params':{'momentum':0.5,'affine':True,'track_running_stats':True}"
SMALL_ECAPA = "ecapa_params={'channels':256,'embd_dim':96,'mfa_conv':512}"
SMALL_RESNET = "'planes':[16,32,64,128]"
# (blueprint, creation, feature dim, [(frames, seed)], weight seed, tolerance)
CASES = [
    ("xvector.py", "Xvector(24,10,nonlinearity='tanh',training=False,extracted_embedding='near')", 24, [(50, 1), (7, 2)], 31, 1e-4),
    ("extended_xvector.py", "ExtendedXvector(24,10,nonlinearity='tanh',training=False)", 24, [(50, 1), (1, 2)], 32, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,tdnn_layer_params={'bn-relu':True},extracted_embedding='near')", 24, [(60, 1), (2, 2)], 33, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,tdnn_layer_params={'bn':False})", 24, [(60, 1)], 34, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,tdnn_layer_params={'nonlinearity':'tanh','nonlinearity_params':{},%s},"
                           "tdnn7_params={'nonlinearity':'','bn':True},extracted_embedding='near')" % BN_AFFINE, 24, [(60, 1), (5, 3)], 35, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,SE=True,se_ratio=8,extend=True,pooling_params={'stddev':False})", 24, [(60, 1), (4, 3)], 36, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,skip_connection=True,tdnn_layer_params={'bn-relu':True,%s})" % BN_AFFINE, 24, [(60, 1)], 37, 1e-4),
    # near-constant channels under a one-layer attention: the reference's own f32 result is 3e-5 .. 9e-5 from a float64 evaluation
    # of the same program, so 1e-4 against it is not attainable by construction - 3e-4 here
    ("snowdar_xvector.py", "Xvector(24,10,training=False,pooling='attentive',pooling_params={'affine_layers':1,'hidden_size':32},tdnn6=False,"
                           "extracted_embedding='near')", 24, [(60, 1), (2, 3)], 44, 3e-4),
    ("factored_xvector.py", "Xvector(24,10,nonlinearity='tanh',semi_orth=False,training=False)", 24, [(60, 1), (6, 3)], 38, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,ecapa_params={'channels':256,'embd_dim':128,'mfa_conv':768,'scale':4})", 40, [(80, 5), (9, 6)], 35, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,%s,pooling='ecpa-attentive',pooling_params={'hidden_size':64,'time_attention':False})" % SMALL_ECAPA,
     40, [(80, 5)], 36, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,%s,pooling='statistics')" % SMALL_ECAPA, 40, [(80, 5)], 37, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,fc1=True,extracted_embedding='near',%s,fc2_params={'nonlinearity':'relu','bn-relu':True})" % SMALL_ECAPA,
     40, [(80, 5), (3, 6)], 39, 1e-4),
    ("resnet_xvector.py", "ResNetXvector(24,10,training=False,resnet_params={'layers':[1,1,1,1],%s,'use_se':True,'se_ratio':8},fc1=True,"
                          "pooling_params={'stddev':False})" % SMALL_RESNET, 24, [(40, 5), (9, 6)], 41, 1e-4),
    ("resnet_xvector.py", "ResNetXvector(24,10,training=False,resnet_params={'layers':[1,2,1,1],%s,'zero_init_residual':True},"
                          "extracted_embedding='near_affine')" % SMALL_RESNET, 24, [(40, 5)], 42, 1e-4),
    # training-time wrappers that must vanish at extraction: context / hidden / input dropout, mixup, SpecAugment
    ("snowdar_xvector.py", "Xvector(24,10,training=False,context_dropout=0.1,hidden_dropout=0.2,aug_dropout=0.1,"
                           "dropout_params={'type':'random','start_p':0.1})", 24, [(60, 1)], 51, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,mixup=True,specaugment=True)", 24, [(60, 1)], 56, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,aug_dropout=0.1,tail_dropout=0.2,mixup=True,%s)" % SMALL_ECAPA, 40, [(80, 5)], 60, 1e-4),
    # poolings with non-default shapes
    ("snowdar_xvector.py", "Xvector(24,10,training=False,pooling='multi-head',pooling_params={'num_head':2,'stddev':False,'num_nodes':256})", 24, [(60, 1), (2, 2)], 52, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,pooling='attentive',pooling_params={'affine_layers':2,'hidden_size':16,'context':[-2,0,2],"
                           "'num_nodes':300},extend=True,SE=True)", 24, [(60, 1), (3, 2)], 54, 1e-4),
    ("snowdar_xvector.py", "Xvector(24,10,training=False,pooling='lde',pooling_params={'num_head':16,'num_nodes':128},tdnn6=False,"
                           "extracted_embedding='near')", 24, [(60, 1), (1, 2)], 55, 1e-4),
    # input normalisation variants in front of the 2-D trunk; ECAPA beyond maxChunk (framework.py:34-47: two chunks, weighted mean)
    ("resnet_xvector.py", "ResNetXvector(24,10,training=False,cmvn=True,cmvn_params={'mean_norm':True,'std_norm':False},resnet_params={'layers':[1,1,1,1],%s})"
                          % SMALL_RESNET, 24, [(40, 5), (1, 6)], 57, 1e-4),
    ("resnet_xvector.py", "ResNetXvector(24,10,training=False,cmvn=True,cmvn_params={'mean_norm':False,'std_norm':True},resnet_params={'layers':[1,1,1,1],%s,"
                          "'full_pre_activation':False})" % SMALL_RESNET, 24, [(40, 5)], 58, 1e-4),
    ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,%s)" % SMALL_ECAPA, 40, [(10003, 5)], 61, 1e-4),
]


@pytest.mark.parametrize("case", CASES, ids=[c[1][:60] for c in CASES])
def test_option_combinations_against_the_reference_itself(case, tmp_path):
    import torch
    from libs.amd import ir, synth
    blueprint, creation, dim, utts, wseed, tol = case
    out = str(tmp_path / "adhoc.npz")
    spec = dict(blueprint=blueprint, creation=creation, dim=dim, utts=[list(u) for u in utts], wseed=wseed)
    r = subprocess.run([sys.executable, os.path.join(helpers.REPO, "oracle", "gen_golden.py"), "--adhoc", json.dumps(spec), out],
                       capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    ref = np.load(out)["embeddings"]
    model = helpers.build_model(blueprint, creation)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, wseed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, dim)
    for (T, seed), e in zip(utts, ref):
        assert helpers.rel_err(ir_interp.extract(graph, synth.synth_feats(T, dim, seed)), e) < tol, (creation, T)
