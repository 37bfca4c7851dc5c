#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation itself.

Runs only in the build container (needs /root/reference, read-only).  It imports the
reference's PyTorch extractor with the three import shims of SURVEY.md section 8(c), loads
the deterministic synthetic weights of `libs/amd/synth.py` into the reference model, runs
the reference's own `model.extract_embedding()` per utterance on CPU (exactly what
pipeline/onestep/extract_embeddings.py:73-83 does) and stores the outputs.

The fixtures hold *outputs only* (plus the case description); weights and inputs are
regenerated from the seed recipe by the tests, so fixtures stay small.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [case ...]
"""

import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLDEN = os.path.join(REPO, "tests", "golden")


def install_shims():
    """stray imports in the reference tree."""
    tk = types.ModuleType("tkinter"); tk.N = None; tk.__path__ = []
    mb = types.ModuleType("tkinter.messagebox"); mb.NO = None
    tu = types.ModuleType("turtle"); tu.xcor = None
    sys.modules.setdefault("tkinter", tk)
    sys.modules.setdefault("tkinter.messagebox", mb)
    sys.modules.setdefault("turtle", tu)
    sys.modules.setdefault("scipye", types.ModuleType("scipye"))


def load_synth():
    import importlib.util
    p = os.path.join(REPO, "asv-subtools_amd", "pytorch", "libs", "amd", "synth.py")
    spec = importlib.util.spec_from_file_location("asv_synth", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# Each extractor case: blueprint file (under reference pytorch/model), creation string,
# feature dim, list of (num_frames, seed) utterances, weight seed.
LAUNCHER_FC2 = ("fc2_params={'nonlinearity':'','bn_params':{'momentum':0.5,'affine':False,"
                "'track_running_stats':True}}")
CASES = {
    # BASELINE config C1: 100 x [200,30], "far" embedding.
    "xvector_c1": dict(blueprint="xvector.py", creation="Xvector(30,10,training=False)", dim=30,
                       utts=[(200, i) for i in range(100)], wseed=0),
    # C2 model on ragged lengths incl. degenerate ones, "near" embedding.
    "xvector_near_ragged": dict(blueprint="xvector.py",
                                creation="Xvector(80,10,training=False,extracted_embedding='near')",
                                dim=80, utts=[(1, 1000), (2, 1001), (5, 1002), (37, 1003), (200, 1004),
                                              (201, 1005), (640, 1006), (1000, 1007)], wseed=1),
    # framework.py:34-47 chunking: T > maxChunk (10000) => 2 chunks of 5000/5001; 20001 => 3.
    "xvector_chunked": dict(blueprint="xvector.py", creation="Xvector(30,10,training=False)", dim=30,
                            utts=[(10001, 2000), (20001, 2001)], wseed=0),
    # C3 model, defaults (channels 1024, ecpa-attentive, "near" = fc2 incl. ReLU+BN).
    "ecapa_c3": dict(blueprint="ecapa_tdnn_xvector.py", creation="ECAPA_TDNN(80,10,training=False)",
                     dim=80, utts=[(300, 3000), (300, 3001), (2, 3002), (9, 3003), (123, 3004), (517, 3005)],
                     wseed=2),
    # launcher variant (runEcapaXvector_online.py:221-247): fc2 without ReLU, BN affine=False.
    "ecapa_launcher": dict(blueprint="ecapa_tdnn_xvector.py",
                           creation="ECAPA_TDNN(80,10,training=False,%s)" % LAUNCHER_FC2,
                           dim=80, utts=[(300, 3100), (411, 3101), (64, 3102)], wseed=3),
    # smaller ECAPA (C=512, the README's 6.5M model) with fc1 and "far"/"near_affine" positions.
    "ecapa_c512_fc1_far": dict(blueprint="ecapa_tdnn_xvector.py",
                               creation="ECAPA_TDNN(80,10,training=False,extracted_embedding='far',fc1=True,"
                                        "ecapa_params={'channels':512,'embd_dim':192,'mfa_conv':1536})",
                               dim=80, utts=[(200, 3200), (333, 3201)], wseed=4),
    "ecapa_c512_near_affine": dict(blueprint="ecapa_tdnn_xvector.py",
                                   creation="ECAPA_TDNN(80,10,training=False,extracted_embedding='near_affine',"
                                            "ecapa_params={'channels':512,'embd_dim':192,'mfa_conv':1536})",
                                   dim=80, utts=[(200, 3300), (150, 3301)], wseed=5),
    # the extended TDNN (E-TDNN) blueprint, both embedding positions, ragged lengths
    "extended_far": dict(blueprint="extended_xvector.py", creation="ExtendedXvector(40,10,training=False)", dim=40,
                         utts=[(200, 6000), (9, 6001), (333, 6002), (1, 6003)], wseed=9),
    "extended_near_plain": dict(blueprint="extended_xvector.py",
                                creation="ExtendedXvector(40,10,extend=False,training=False,extracted_embedding='near')", dim=40,
                                utts=[(150, 6100), (64, 6101)], wseed=10),
    # the composite ("snowdar") x-vector: defaults; every structural option at once; no tdnn6
    "snowdar_default": dict(blueprint="snowdar_xvector.py", creation="Xvector(40,10,training=False)", dim=40,
                            utts=[(200, 6200), (31, 6201)], wseed=11),
    "snowdar_full_near": dict(blueprint="snowdar_xvector.py",
                              creation="Xvector(40,10,training=False,extend=True,skip_connection=True,SE=True,extracted_embedding='near')", dim=40,
                              utts=[(200, 6300), (77, 6301), (3, 6302)], wseed=12),
    "snowdar_no_tdnn6": dict(blueprint="snowdar_xvector.py", creation="Xvector(40,10,training=False,tdnn6=False,extracted_embedding='near_affine')",
                             dim=40, utts=[(120, 6400)], wseed=13),
    # SURVEY 8(f) rank 3, alternative poolings: attentive statistics pooling (single shared head), two affine layers with a
    # time context in the attention / one affine layer and mean only
    "snowdar_attentive": dict(blueprint="snowdar_xvector.py",
                              creation="Xvector(40,10,training=False,pooling='attentive',pooling_params={'affine_layers':2,'hidden_size':64,'context':[-1,0,1]})",
                              dim=40, utts=[(200, 6700), (33, 6701), (2, 6702)], wseed=16),
    "snowdar_attentive_mean": dict(blueprint="snowdar_xvector.py",
                                   creation="Xvector(40,10,training=False,pooling='attentive',pooling_params={'affine_layers':1,'stddev':False},extracted_embedding='near')",
                                   dim=40, utts=[(120, 6800)], wseed=17),
    # multi-head poolings (pooling.py:371-587): heads over channel splits (shared: a logit per head; un-shared: per channel,
    # two grouped affine layers), global heads with fixed / learned temperatures (shared and un-shared)
    "snowdar_multihead": dict(blueprint="snowdar_xvector.py",
                              creation="Xvector(40,10,training=False,pooling='multi-head',pooling_params={'num_head':4})",
                              dim=40, utts=[(200, 6900), (41, 6901), (3, 6902)], wseed=18),
    "snowdar_multihead_unshared": dict(blueprint="snowdar_xvector.py",
                                       creation="Xvector(40,10,training=False,pooling='multi-head',pooling_params={'num_head':3,'share':False,"
                                                "'affine_layers':2,'hidden_size':32,'num_nodes':600,'context':[-1,0,1]},extracted_embedding='near')",
                                       dim=40, utts=[(150, 6910), (20, 6911)], wseed=19),
    "snowdar_multires": dict(blueprint="snowdar_xvector.py",
                             creation="Xvector(40,10,training=False,pooling='multi-resolution',pooling_params={'num_head':4,'temperature':True,"
                                      "'affine_layers':2,'hidden_size':16,'num_nodes':600})",
                             dim=40, utts=[(180, 6920), (29, 6921), (1, 6922)], wseed=20),
    "snowdar_multires_learned": dict(blueprint="snowdar_xvector.py",
                                     creation="Xvector(40,10,training=False,pooling='multi-resolution',pooling_params={'num_head':3,'temperature':True,"
                                              "'fixed':False,'share':False,'affine_layers':1,'num_nodes':300},extracted_embedding='near')",
                                     dim=40, utts=[(160, 6930), (12, 6931)], wseed=21),
    # learnable dictionary encoding pooling (pooling.py:130-162): 8 and 40 centres
    "snowdar_lde": dict(blueprint="snowdar_xvector.py",
                        creation="Xvector(40,10,training=False,pooling='lde',pooling_params={'num_head':8,'num_nodes':400})",
                        dim=40, utts=[(200, 6960), (23, 6961), (1, 6962)], wseed=24),
    "snowdar_lde40": dict(blueprint="snowdar_xvector.py",
                          creation="Xvector(40,10,training=False,pooling='lde',pooling_params={'num_head':40,'num_nodes':120},extracted_embedding='near')",
                          dim=40, utts=[(90, 6970), (11, 6971)], wseed=25),
    # xi-vector posterior pooling (pooling.py:165-218): posterior mean only / mean + std
    "snowdar_xi_mean": dict(blueprint="snowdar_xvector.py",
                            creation="Xvector(40,10,training=False,pooling='xi-postmean-softplus2',pooling_params={'hidden_size':64,'num_nodes':600})",
                            dim=40, utts=[(200, 6940), (37, 6941), (1, 6942)], wseed=22),
    "snowdar_xi_dist": dict(blueprint="snowdar_xvector.py",
                            creation="Xvector(40,10,training=False,pooling='xi-postdist-softplus2',pooling_params={'hidden_size':32},extracted_embedding='near')",
                            dim=40, utts=[(150, 6950), (9, 6951)], wseed=23),
    # the factorised TDNN (TDNN-F) x-vector with its dense skip wiring, both positions
    "factored_far": dict(blueprint="factored_xvector.py", creation="Xvector(40,10,training=False)", dim=40,
                         utts=[(200, 6500), (45, 6501), (7, 6502)], wseed=14),
    "factored_near": dict(blueprint="factored_xvector.py", creation="Xvector(40,10,training=False,extracted_embedding='near',embd_dim=256)", dim=40,
                          utts=[(150, 6600)], wseed=15),
    # BASELINE config C5 extractor: ResNet34-SE (32-64-128-256), launcher-style fc2 (runResnetXvector_online.py:221-260)
    "resnet34se_c5": dict(blueprint="resnet_xvector.py",
                          creation="ResNetXvector(80,10,training=False,resnet_params={'use_se':True,'se_ratio':4,'full_pre_activation':False},"
                                   "fc2_params={'nonlinearity':'','bn_params':{'momentum':0.5,'affine':False,'track_running_stats':True}})",
                          dim=80, utts=[(200, 5000), (203, 5001), (9, 5002), (64, 5003)], wseed=6),
    # cmvn=True: InputSequenceNormalization (mean and std) in front of the trunk
    "resnet34_cmvn": dict(blueprint="resnet_xvector.py",
                          creation="ResNetXvector(40,10,training=False,cmvn=True,cmvn_params={'mean_norm':True,'std_norm':True},"
                                   "resnet_params={'full_pre_activation':False})",
                          dim=40, utts=[(120, 5200), (45, 5201)], wseed=8),
    # the blueprint's DEFAULT resnet_params: full pre-activation blocks (BN-ReLU-conv), no SE, default fc2
    "resnet34_preact": dict(blueprint="resnet_xvector.py", creation="ResNetXvector(80,10,training=False)",
                            dim=80, utts=[(200, 5300), (131, 5301), (9, 5302)], wseed=9),
    # pre-activation + SE, odd feature dim
    "resnet34se_preact": dict(blueprint="resnet_xvector.py",
                              creation="ResNetXvector(45,10,training=False,resnet_params={'use_se':True,'se_ratio':4})",
                              dim=45, utts=[(120, 5400), (64, 5401)], wseed=10),
    # no SE, default fc2 (ReLU + affine BN), odd feature dim -> ceil division at every stride-2 stage
    # Bottleneck blocks (resnet.py:113-208), original form + SE, and the pre-activation form; small widths, odd feature dim
    "resnet_bottleneck_se": dict(blueprint="resnet_xvector.py",
                                 creation="ResNetXvector(40,10,training=False,resnet_params={'block':'Bottleneck','layers':[1,2,1,1],'planes':[16,32,64,128],"
                                          "'use_se':True,'se_ratio':4,'full_pre_activation':False})",
                                 dim=40, utts=[(90, 5500), (33, 5501)], wseed=11),
    "resnet_bottleneck_preact": dict(blueprint="resnet_xvector.py",
                                     creation="ResNetXvector(33,10,training=False,resnet_params={'block':'Bottleneck','layers':[1,1,1,1],'planes':[16,32,64,128]})",
                                     dim=33, utts=[(70, 5600), (21, 5601)], wseed=12),
    # the frame-weighting poolings behind the 2-D trunk (resnet_xvector.py:104-111, over the [B, C*F', T'] reshape of :193); small trunks
    "resnet_attentive": dict(blueprint="resnet_xvector.py",
                             creation="ResNetXvector(40,10,training=False,pooling='attentive',pooling_params={'hidden_size':32,'context':[-1,0,1]},"
                                      "resnet_params={'layers':[1,1,1,1],'planes':[16,32,64,128],'use_se':True,'full_pre_activation':False})",
                             dim=40, utts=[(150, 5700), (64, 5701), (9, 5702)], wseed=16),
    "resnet_multihead": dict(blueprint="resnet_xvector.py",
                             creation="ResNetXvector(33,10,training=False,pooling='multi-head',pooling_params={'num_head':4,'share':True,'affine_layers':1},"
                                      "resnet_params={'layers':[1,1,1,1],'planes':[16,32,64,128]})",
                             dim=33, utts=[(120, 5800), (41, 5801)], wseed=17),
    "resnet_multires": dict(blueprint="resnet_xvector.py",
                            creation="ResNetXvector(40,10,training=False,fc1=True,extracted_embedding='far',pooling='multi-resolution',"
                                     "pooling_params={'num_head':4,'share':True,'affine_layers':2,'hidden_size':32,'temperature':True},"
                                     "resnet_params={'layers':[1,1,1,1],'planes':[16,32,64,128],'full_pre_activation':False})",
                            dim=40, utts=[(130, 5900), (57, 5901)], wseed=18),
    "resnet_lde": dict(blueprint="resnet_xvector.py",
                       creation="ResNetXvector(40,10,training=False,pooling='lde',pooling_params={'num_head':8},"
                                "resnet_params={'layers':[1,1,1,1],'planes':[16,32,64,128],'full_pre_activation':False})",
                       dim=40, utts=[(140, 6000), (33, 6001)], wseed=19),
    # ECAPA with the reference's other constructible pooling: AttentiveStatisticsPooling with time context (ecapa_tdnn_xvector.py:275-281)
    "ecapa_attentive": dict(blueprint="ecapa_tdnn_xvector.py",
                            creation="ECAPA_TDNN(40,10,training=False,pooling='attentive',pooling_params={'hidden_size':64,'context':[-1,0,1]},"
                                     "ecapa_params={'channels':512,'embd_dim':128,'mfa_conv':768})",
                            dim=40, utts=[(150, 6100), (37, 6101)], wseed=20),
    "resnet34_plain": dict(blueprint="resnet_xvector.py",
                           creation="ResNetXvector(61,10,training=False,resnet_params={'full_pre_activation':False})",
                           dim=61, utts=[(150, 5100), (77, 5101)], wseed=7),
}


def run_extractor_case(name, case, synth, path=None):
    import numpy as np
    import torch
    import libs.support.utils as utils

    torch.set_num_threads(os.cpu_count() or 1)
    model = utils.create_model_from_py(os.path.join(REF, "pytorch", "model", case["blueprint"]),
                                       case["creation"])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, case["wseed"])
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    embs = []
    for (T, seed) in case["utts"]:
        x = synth.synth_feats(T, case["dim"], seed)
        e = model.extract_embedding(x)            # the reference's own wrapper + model
        embs.append(e.numpy().astype(np.float32))
    out = dict(
        embeddings=np.stack(embs),
        utts=np.asarray(case["utts"], dtype=np.int64),
        wseed=np.int64(case["wseed"]),
        dim=np.int64(case["dim"]),
        creation=np.array(case["creation"]),
        blueprint=np.array(case["blueprint"]),
        shape_keys=np.array(list(shapes.keys())),
        shape_vals=np.array([",".join(str(d) for d in s) for s in shapes.values()]),
        torch_version=np.array(torch.__version__),
    )
    path = path or os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: embeddings %s, %d params" % (path, out["embeddings"].shape,
                                                   sum(int(np.prod(s)) for s in shapes.values())))


def main(argv):
    if not os.path.isdir(REF):
        sys.exit("gen_golden.py needs the reference tree at %s (build container only)" % REF)
    install_shims()
    sys.path.insert(0, os.path.join(REF, "pytorch"))
    os.makedirs(GOLDEN, exist_ok=True)
    synth = load_synth()
    if argv and argv[0] == "--adhoc":
        # python oracle/gen_golden.py --adhoc '<json case>' <out.npz>: one extractor case that is not a committed fixture (the
        # build-container differential test tests/test_reference_differential.py drives the reference through this)
        import json
        case = json.loads(argv[1])
        case["utts"] = [tuple(u) for u in case["utts"]]
        run_extractor_case("adhoc", case, synth, path=argv[2])
        return
    names = argv or list(CASES) + list(EXTRA_CASES)
    for n in names:
        if n in CASES:
            run_extractor_case(n, CASES[n], synth)
        elif n in EXTRA_CASES:
            EXTRA_CASES[n](n, synth)
        else:
            sys.exit("unknown case %s (have: %s)" % (n, ", ".join(list(CASES) + list(EXTRA_CASES))))


def run_scoring_plda(name, synth):
    """PLDA EM + transform + LLR of score/pyplda/plda_base.py, the two-covariance scorer of
    gaussian-plda-scoring.py and the EER of computeEER-like-Bosaris.py, all executed from the
    reference sources on a planted-speaker embedding set."""
    import importlib.util
    import numpy as np
    import libs.support.kaldi_io as ref_kaldi_io
    sys.modules["kaldi_io"] = ref_kaldi_io                       # SURVEY.md 8(c) shim 4
    sys.path.insert(0, os.path.join(REF, "score", "pyplda"))
    import plda_base as PB

    def load(path, modname):
        spec = importlib.util.spec_from_file_location(modname, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    argv, sys.argv = sys.argv, ["x"]
    try:
        bosaris = load(os.path.join(REF, "computeEER-like-Bosaris.py"), "ref_bosaris")
        twocov = load(os.path.join(REF, "score", "pyplda", "gaussian-plda-scoring.py"), "ref_twocov")
    finally:
        sys.argv = argv

    dim, n_spk, per_spk = 48, 120, 6
    train, labels = synth.synth_speaker_embeddings(n_spk, per_spk, dim, seed=11, within=1.0, between=0.8)
    train = train.astype(np.float64)
    stats = PB.PldaStats(dim)
    for spk in range(n_spk):
        stats.add_samples(1.0, train[labels == spk])
    assert stats.is_sorted()
    est = PB.PldaEstimation(stats)
    est.estimate(num_em_iters=5)
    plda = est.get_output()
    # plda_base.py keeps `offset` as a [dim,1] column (152-156); with the 1-D vectors that give
    # `self.dim = ivector.shape[-1]` its intended value (Kaldi's dim) it must be 1-D too, otherwise
    # numpy broadcasts T.x + offset to a [dim,dim] matrix.  Same numbers, intended shapes.
    plda.offset = np.asarray(plda.offset).reshape(-1)

    ev, ev_labels = synth.synth_speaker_embeddings(40, 5, dim, seed=12, within=1.0, between=0.8)
    ei, ti, tgt = synth.synth_trials(ev_labels, 3000, seed=13)
    ev64 = ev.astype(np.float64)
    tr = np.stack([plda.transform_ivector(v, 1) for v in ev64])
    llr = np.array([float(plda.log_likelihood_ratio(tr[a], 1, tr[b])) for a, b in zip(ei, ti)])
    gamma, lam, c, k = twocov.CalculateVar(est.between_var, est.within_var + 5e-5 * np.eye(dim), est.mean)
    tc = np.array([twocov.PLDAScoring(ev64[a].reshape(-1, 1), ev64[b].reshape(-1, 1), gamma, lam, c, k) for a, b in zip(ei, ti)])
    rows = [[float(s), "target" if t else "nontarget"] for s, t in zip(llr, tgt)]
    eer, thr = bosaris.compute_eer(rows)
    out = dict(dim=np.int64(dim), n_spk=np.int64(n_spk), per_spk=np.int64(per_spk),
               mean=np.asarray(est.mean).reshape(-1), within_var=est.within_var, between_var=est.between_var,
               transform=plda.transform, psi=plda.psi, offset=np.asarray(plda.offset).reshape(-1),
               transformed=tr, llr=llr, two_cov=tc, trials_e=ei, trials_t=ti, trials_tgt=tgt,
               eer=np.float64(eer), eer_threshold=np.float64(thr))
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d trials, EER %.4f%% thr %.5f" % (path, len(llr), 100 * eer, thr))


def run_score_norm(name, synth):
    """S-norm / AS-norm of score/ScoreNormalization.py:70-179 executed from the reference source (pandas) on
    cosine scores of a planted-speaker set: text score files in, text score files out, exactly as
    recipe/voxcelebSRC/gather_results_from_epochs.sh:103-183 drives it.  Inputs are stored as the f32 values
    that were written (repr-exact), outputs as the f64 the reference printed."""
    import importlib.util
    import tempfile
    import types
    import numpy as np
    spec = importlib.util.spec_from_file_location("ref_score_norm", os.path.join(REF, "score", "ScoreNormalization.py"))
    sn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sn)

    dim = 32
    emb, labels = synth.synth_speaker_embeddings(30, 4, dim, seed=21, within=1.0, between=0.9)
    emb = emb / np.linalg.norm(emb, axis=1, keepdims=True)
    enroll, test, cohort = emb[:24], emb[24:72], emb[72:]            # 24 enrol, 48 test, 48 cohort vectors
    ec = (enroll @ cohort.T).astype(np.float32)
    tc = (test @ cohort.T).astype(np.float32)
    rng = np.random.RandomState(5)
    n_trials = 400
    # distinct (enrol, test) pairs, like a real trial list: the cross-select merge of the reference multiplies the
    # cohort rows of a trial that is listed twice (and so changes its ddof=1 std)
    pairs = rng.permutation(len(enroll) * len(test))[:n_trials]
    ei = (pairs // len(test)).astype(np.int32)
    ti = (pairs % len(test)).astype(np.int32)
    sc = np.einsum("ij,ij->i", enroll[ei], test[ti]).astype(np.float32)
    out = dict(enroll_cohort=ec, test_cohort=tc, trials_e=ei, trials_t=ti, scores=sc)
    with tempfile.TemporaryDirectory() as td:
        def write(path, rows):
            with open(path, "w") as f:
                for a, b, v in rows:
                    f.write("%s %s %s\n" % (a, b, repr(float(np.float32(v)))))
        write(os.path.join(td, "et"), [("e%d" % a, "t%d" % b, v) for a, b, v in zip(ei, ti, sc)])
        write(os.path.join(td, "ec"), [("e%d" % a, "c%d" % c, ec[a, c]) for a in range(ec.shape[0]) for c in range(ec.shape[1])])
        write(os.path.join(td, "tc"), [("t%d" % b, "c%d" % c, tc[b, c]) for b in range(tc.shape[0]) for c in range(tc.shape[1])])
        for tag, method, top_n, cross in (("snorm", "snorm", 0, "false"), ("asnorm10", "asnorm", 10, "false"), ("asnorm10x", "asnorm", 10, "true"),
                                          ("asnorm_all", "asnorm", 300, "false")):
            args = types.SimpleNamespace(method=method, top_n=top_n, second_cohort="true", cross_select=cross, input_score=os.path.join(td, "et"),
                                         enroll_cohort_score=os.path.join(td, "ec"), test_cohort_score=os.path.join(td, "tc"),
                                         output_score=os.path.join(td, "out_" + tag))
            (sn.snorm if method == "snorm" else sn.asnorm)(args)
            vals = [float(line.split()[2]) for line in open(args.output_score)]
            keys = [tuple(line.split()[:2]) for line in open(args.output_score)]
            assert keys == [("e%d" % a, "t%d" % b) for a, b in zip(ei, ti)], "the reference keeps the trial order"
            out[tag] = np.asarray(vals, dtype=np.float64)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d trials, %d cohort vectors; mean |snorm| %.3f" % (path, n_trials, ec.shape[1], float(np.abs(out["snorm"]).mean())))


def run_plda_ragged(name, synth):
    """plda_base.py PldaStats + PldaEstimation on classes of different sizes (2..9 examples), stats added in ascending size."""
    import numpy as np
    import libs.support.kaldi_io as ref_kaldi_io
    sys.modules["kaldi_io"] = ref_kaldi_io                       # SURVEY.md 8(c) shim 4
    sys.path.insert(0, os.path.join(REF, "score", "pyplda"))
    import plda_base as PB
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers
    meta = dict(n_spk=150, per_spk_max=9, dim=40, seed=31, num_iters=6)
    train, labels = helpers.plda_ragged_set(meta, synth)
    train = train.astype(np.float64)
    stats = PB.PldaStats(meta["dim"])
    counts = np.bincount(labels)
    for spk in np.unique(labels)[np.argsort(counts, kind="stable")]:
        stats.add_samples(1.0, train[labels == spk])
    assert stats.is_sorted()
    est = PB.PldaEstimation(stats)
    est.estimate(num_em_iters=meta["num_iters"])
    out = dict(meta, mean=np.asarray(est.mean).reshape(-1), within_var=est.within_var, between_var=est.between_var)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d vectors, %d classes, sizes %s" % (path, len(labels), len(counts), sorted(set(counts.tolist()))))


def run_plda_adapt(name, synth):
    """PldaUnsupervisedAdaptor (plda_base.py:344-485) on a shifted / stretched in-domain set, and the ZCA class of
    score/whiten/{train,do}_ZCA_Whitening.py - the reference code itself."""
    import importlib.util
    import numpy as np
    import libs.support.kaldi_io as ref_kaldi_io
    sys.modules["kaldi_io"] = ref_kaldi_io
    sys.path.insert(0, os.path.join(REF, "score", "pyplda"))
    import plda_base as PB
    dim, seed = 32, 51
    train, labels = synth.synth_speaker_embeddings(100, 5, dim, seed=seed, within=1.0, between=0.8)
    train = train.astype(np.float64)
    stats = PB.PldaStats(dim)
    for spk in range(100):
        stats.add_samples(1.0, train[labels == spk])
    est = PB.PldaEstimation(stats)
    est.estimate(num_em_iters=4)
    plda = est.get_output()
    base = dict(mean=np.asarray(plda.mean).reshape(-1).copy(), transform=np.array(plda.transform), psi=np.array(plda.psi))
    adapt, _ = synth.synth_speaker_embeddings(60, 4, dim, seed=seed + 1, within=1.3, between=1.1)
    adapt = (adapt * np.linspace(0.8, 1.6, dim)[None, :] + 0.3).astype(np.float32)           # a different domain
    ad = PB.PldaUnsupervisedAdaptor(mean_diff_scale=1.0, within_covar_scale=0.3, between_covar_scale=0.7)
    for v in adapt.astype(np.float64):
        ad.add_stats(1.0, v)
    plda.mean = np.asarray(plda.mean).reshape(-1, 1)
    ad.update_plda(plda)
    plda.mean = np.asarray(plda.mean).reshape(-1)
    plda.compute_derived_vars()
    plda.offset = np.asarray(plda.offset).reshape(-1)
    ev, ev_labels = synth.synth_speaker_embeddings(30, 4, dim, seed=seed + 2, within=1.3, between=1.1)
    ev = (ev * np.linspace(0.8, 1.6, dim)[None, :] + 0.3).astype(np.float32)
    ei, ti, tgt = synth.synth_trials(ev_labels, 1500, seed=seed + 3)
    tr = np.stack([plda.transform_ivector(v.astype(np.float64), 1) for v in ev])
    llr = np.array([float(plda.log_likelihood_ratio(tr[a], 1, tr[b])) for a, b in zip(ei, ti)])

    def load(path, modname):
        """The scripts run their command line at import time: execute the class definition part only (up to their own
        '## class defined end ##' marker), from the reference file where it lies."""
        import types
        src = open(path, encoding="utf8").read().split("## class defined end ##")[0]
        m = types.ModuleType(modname)
        exec(compile(src, path, "exec"), m.__dict__)
        return m
    z1 = load(os.path.join(REF, "score", "whiten", "train_ZCA_Whitening.py"), "ref_zca_train").ZCA().fit(adapt.astype(np.float64))
    z2 = load(os.path.join(REF, "score", "whiten", "do_ZCA_Whitening.py"), "ref_zca_do").ZCA().fit(adapt.astype(np.float64))
    out = dict(dim=dim, seed=seed, base_mean=base["mean"], base_transform=base["transform"], base_psi=base["psi"],
               adapted_mean=np.asarray(plda.mean).reshape(-1), adapted_psi_sorted=np.sort(np.real(np.asarray(plda.psi))), llr=llr,
               trials_e=ei, trials_t=ti, zca_train_whiten=z1.whiten_, zca_do_whiten=z2.whiten_, zca_do_mean=z2.mean_)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %s: psi %s..%s" % (path, out["adapted_psi_sorted"][0], out["adapted_psi_sorted"][-1]))


EXTRA_CASES = {"plda_adapt": run_plda_adapt, "scoring_plda": run_scoring_plda, "score_norm": run_score_norm, "plda_ragged": run_plda_ragged}


if __name__ == "__main__":
    main(sys.argv[1:])
