"""The C ABI library loads, exports every declared symbol, and rejects malformed programs
without a GPU; the recorder produces the expected programs for the target blueprints."""

import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers
import ir_interp
from helpers import rel_err


def test_library_exports_every_declared_symbol(repo_root):
    from libs.amd import capi
    lib = capi.lib()
    hdr = open(os.path.join(repo_root, "include", "asv_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(asv_[a-z0-9_]+)\s*\(", hdr)))
    assert declared and set(declared) == set(capi.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.asv_version() >= 100
    assert isinstance(lib.asv_last_error(), bytes)


def test_struct_layouts_match_the_c_header(repo_root, tmp_path):
    """sizeof / field offsets of every ABI struct as a C compiler sees include/asv_amd.h == ctypes."""
    import subprocess
    from libs.amd import capi
    src = tmp_path / "sizes.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "asv_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(asv_tdnn_desc_t), sizeof(asv_pool_desc_t), sizeof(asv_attpool_desc_t), sizeof(asv_eltwise_desc_t), sizeof(asv_kernel_time_t));
  printf("%zu %zu %zu %zu\n", offsetof(asv_tdnn_desc_t, weight), offsetof(asv_tdnn_desc_t, scale), offsetof(asv_tdnn_desc_t, res_ch_off), offsetof(asv_eltwise_desc_t, scale));
  printf("%zu %zu\n", offsetof(asv_kernel_time_t, total_ms), offsetof(asv_kernel_time_t, flops));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(asv_lde_desc_t), sizeof(asv_res2_desc_t), sizeof(asv_grid_input_desc_t), sizeof(asv_grid_flatten_desc_t),
         sizeof(asv_im2col_desc_t), sizeof(asv_fbank_opts_t));
  printf("%zu %zu %zu %zu\n", offsetof(asv_im2col_desc_t, df), offsetof(asv_im2col_desc_t, b_buf), offsetof(asv_im2col_desc_t, act), offsetof(asv_grid_flatten_desc_t, out_buf));
  return 0;
}''')
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(repo_root, "include"), str(src), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)], text=True).split("\n")
    sizes = [int(v) for v in lines[0].split()]
    assert sizes == [C.sizeof(capi.TdnnDesc), C.sizeof(capi.PoolDesc), C.sizeof(capi.AttPoolDesc), C.sizeof(capi.EltwiseDesc), C.sizeof(capi.KernelTime)]
    offs = [int(v) for v in lines[1].split()]
    assert offs == [capi.TdnnDesc.weight.offset, capi.TdnnDesc.scale.offset, capi.TdnnDesc.res_ch_off.offset, capi.EltwiseDesc.scale.offset]
    assert [int(v) for v in lines[2].split()] == [capi.KernelTime.total_ms.offset, capi.KernelTime.flops.offset]
    # every other struct of the header, incl. the ones round 4 touched (the strided gather's elementwise prologue, grid_flatten)
    assert [int(v) for v in lines[3].split()] == [C.sizeof(capi.LdeDesc), C.sizeof(capi.Res2Desc), C.sizeof(capi.GridInputDesc), C.sizeof(capi.GridFlattenDesc),
                                                  C.sizeof(capi.Im2colDesc), C.sizeof(capi.FbankOpts)]
    assert [int(v) for v in lines[4].split()] == [capi.Im2colDesc.df.offset, capi.Im2colDesc.b_buf.offset, capi.Im2colDesc.act.offset, capi.GridFlattenDesc.out_buf.offset]


def test_no_gpu_calls_fail_loudly_not_silently():
    from libs.amd import capi
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only behaviour")
    lib = capi.lib()
    net = C.c_void_p()
    rc = lib.asv_net_create(C.byref(net), 0, capi.PREC_F32, 0, 30)
    assert rc < 0 and lib.asv_last_error()
    with pytest.raises(capi.AsvError):
        capi.check(rc, "asv_net_create")


def test_xvector_program_shape():
    from libs.amd import ir
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    g = ir.trace(model, type(model).extract_embedding.__wrapped_body__, 80)
    kinds = [op.kind for op in g.ops]
    assert kinds == ["tdnn"] * 5 + ["pool", "tdnn"]
    assert [op.taps for op in g.ops if op.kind == "tdnn"][:3] == [[-2, -1, 0, 1, 2], [-2, 0, 2], [-3, 0, 3]]
    per_frame, per_utt = g.flops_per_frame()
    assert per_frame == 2 * 2807808 and per_utt == 2 * 3000 * 512       # SURVEY.md 8(d) MAC breakdown


def test_ecapa_program_has_no_copies_but_the_res2_passthrough():
    from libs.amd import ir
    model = helpers.build_model("ecapa_tdnn_xvector.py", "ECAPA_TDNN(80,10,training=False)")
    g = ir.trace(model, type(model).extract_embedding.__wrapped_body__, 80)
    kinds = [op.kind for op in g.ops]
    assert "cat" not in kinds
    copies = [op for op in g.ops if op.kind == "eltwise" and op.b is None and op.seg_scale is None and op.scale is None]
    assert len(copies) == 3 and all(op.a.channels == 128 for op in copies)      # group 0 of each Res2Net block
    # the 3C->128 attention conv was split: per-frame part reads 1536 channels + a per-utterance bias
    att = [op for op in g.ops if op.kind == "tdnn" and op.seg_bias is not None]
    assert len(att) == 1 and att[0].inp.channels == 1536 and att[0].act2 == "tanh"
    # Res2 branches take (previous branch + next group) as a fused second input
    assert sum(1 for op in g.ops if op.kind == "tdnn" and op.inp2 is not None) == 18


@pytest.mark.parametrize("name,limit", [("ecapa_c3", 3), ("ecapa_launcher", 2), ("ecapa_c512_fc1_far", 2), ("ecapa_c512_near_affine", 2)])
def test_ecapa_traced_program_reproduces_reference_on_cpu(name, limit):
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    for x, ref in list(zip(helpers.golden_feats(g), g["embeddings"]))[:limit]:
        assert rel_err(ir_interp.extract(graph, x), ref) < 2e-5


TDNN_FAMILY_GOLDENS = ["xvector_c1", "xvector_chunked", "xvector_near_ragged", "extended_far", "extended_near_plain", "factored_far", "factored_near",
                       "snowdar_default", "snowdar_full_near", "snowdar_no_tdnn6", "snowdar_attentive", "snowdar_attentive_mean", "snowdar_multihead",
                       "snowdar_multihead_unshared", "snowdar_multires", "snowdar_multires_learned", "snowdar_xi_mean", "snowdar_xi_dist", "snowdar_lde",
                       "snowdar_lde40"]


@pytest.mark.parametrize("name", TDNN_FAMILY_GOLDENS)
def test_tdnn_family_traced_programs_reproduce_the_reference_on_cpu(name):
    """Every TDNN-family blueprint and pooling of SURVEY.md 8(a)/(f3) - standard / extended / composite / factorised x-vector with
    statistics, attentive, multi-head, multi-resolution, xi-vector and LDE pooling; far / near positions; chunked long inputs and
    one-frame utterances - traced into the layer program and interpreted with the numpy oracle's layer functions: what the device
    is handed is the reference's computation (north_star tolerance 1e-4), before any kernel runs."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    feats = helpers.golden_feats(g)
    pairs = list(zip(feats, g["embeddings"]))
    picks = pairs[:2] + ([min(pairs, key=lambda p: len(p[0]))] if len(pairs) > 2 else [])        # + the shortest utterance of the set
    for x, ref in picks:
        assert rel_err(ir_interp.extract(graph, x), ref) < 1e-4, (name, len(x))


@pytest.mark.parametrize("name", ["ecapa_c3", "ecapa_c512_fc1_far"])
def test_ecapa_late_fusion_passes_preserve_the_program(name):
    """The engine-level passes (one 'res2' op per Res2NetBlock, the running block sum as a second eltwise output) are
    rewrites of the op list only: interpreted on CPU, the fused list gives the unfused list's embedding."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    fused = graph.fused_add_ops(graph.fused_res2_ops())
    kinds = [op.kind for op in fused]
    n_res2 = 3 if name == "ecapa_c3" else 0                  # kernels_res2.hip is written for 128-channel groups (C = 1024, scale 8)
    assert kinds.count("res2") == n_res2 and len(fused) <= len(graph.ops) - n_res2 * 7 - 2
    assert sum(1 for op in fused if getattr(op, "out2", None) is not None) >= 2
    x = helpers.golden_feats(g)[0]
    plain = ir_interp.extract(graph, x)
    assert np.array_equal(ir_interp.extract(graph, x, ops=graph.fused_add_ops()), plain)       # same arithmetic, same order
    assert rel_err(ir_interp.extract(graph, x, ops=fused), plain) < 1e-6


@pytest.mark.parametrize("name", ["resnet34se_c5", "resnet34_plain", "resnet34_preact", "resnet_bottleneck_se"])
def test_resnet_gather_fusion_pass_preserves_the_program(name):
    """Graph.fused_gather_ops: the elementwise pass that closes a ResNet stage (relu(se(y) + identity)) becomes the prologue of the
    stride-2 gathers that are its only readers.  A rewrite of the op list only: same operations in the same order, so the
    interpreted embedding is the unfused one bit for bit; every folded pass disappears from the list."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    fused = graph.fused_gather_ops()
    folded = [op for op in fused if op.kind == "im2col" and (getattr(op, "b", None) is not None or getattr(op, "act", None) is not None)]
    assert len(fused) < len(graph.ops) or not folded
    if name == "resnet34se_c5":
        assert len(graph.ops) - len(fused) == 3 and len(folded) == 4           # three stage transitions; the last one feeds two gathers
        assert all(op.seg_scale is not None and op.b is not None and op.act == "relu" for op in folded)
    written = {op.out.tid for op in fused}
    for op in fused:
        for v in op.inputs():
            assert v.tid == 0 or v.tid in written, "an input of %s lost its producer" % op.kind
    x = helpers.golden_feats(g)[0]
    assert np.array_equal(ir_interp.extract(graph, x, ops=fused), ir_interp.extract(graph, x))


@pytest.mark.parametrize("name,idx", [("resnet34se_c5", 2), ("resnet34se_c5", 3), ("resnet34_plain", 1), ("resnet34_cmvn", 1), ("resnet34_preact", 2),
                                      ("resnet34se_preact", 1), ("resnet_bottleneck_se", 0), ("resnet_bottleneck_se", 1), ("resnet_bottleneck_preact", 0)])
def test_resnet_traced_program_reproduces_reference_on_cpu(name, idx):
    """2-D trunk: row-flattened (time, frequency) grids, BN folded into the convolutions, im2col for the
    stride-2 convolutions, SE with the pitch/width factor folded, per-bin pooling + permuted fc2 columns."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    # the 32 -> 64 and 64 -> 128 transitions: one space-to-depth gather feeds the strided 3x3 and the strided 1x1 shortcut; 128 -> 256:
    # im2col of all taps + 1x1 gather (the pre-activation block feeds its two strided convolutions from different tensors: no sharing)
    if "bottleneck" not in name:
        assert sum(1 for op in graph.ops if op.kind == "im2col") == (6 if "preact" in name else 4)
    if name == "resnet34_cmvn":
        assert [op.kind for op in graph.ops[:3]] == ["pool", "eltwise", "grid_input"]   # InputSequenceNormalization
    per_frame, _ = graph.flops_per_frame()
    if name == "resnet34se_c5":
        assert abs(per_frame - 45.27e6) < 0.02e6
    x = helpers.golden_feats(g)[idx]
    assert rel_err(ir_interp.extract(graph, x), g["embeddings"][idx]) < 2e-5


@pytest.mark.parametrize("name,kinds", [("resnet_attentive", ["flatten", "attpool"]), ("resnet_multihead", ["flatten", "attpool"]),
                                        ("resnet_multires", ["flatten"] + ["attpool"] * 4), ("resnet_lde", ["flatten", "lde"])])
def test_resnet_frame_weighting_poolings_reproduce_reference_on_cpu(name, kinds):
    """ResNetXvector with the reference's other pooling options (resnet_xvector.py:104-111): the [B, C*F', T'] reshape (:193) is
    materialised ONCE, in the reference's channel order c*F' + f, on a sequence domain at the trunk's frame rate (T / 8); the
    attention layers (time context included) and the pooling kernels then run on it like on the frames domain.  Golden vectors:
    the reference's own outputs (oracle/gen_golden.py), every utterance incl. the 9-frame one (two trunk frames)."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model(name)
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    tail = [op.kind for op in graph.ops if op.kind in ("flatten", "attpool", "lde", "pool")]
    assert tail[-len(kinds):] == kinds
    assert ("seq", 3) in graph.domains
    flat = [op for op in graph.ops if op.kind == "flatten"][0]
    assert flat.out.channels == model.stats.input_dim == flat.inp.channels * graph.grid_spec(flat.inp.tid)[2]
    for x, ref in zip(helpers.golden_feats(g), g["embeddings"]):
        assert rel_err(ir_interp.extract(graph, x), ref) < 3e-5, (name, len(x))


def test_ecapa_with_attentive_statistics_pooling_reproduces_reference_on_cpu():
    """ECAPA_TDNN(pooling='attentive') - the one further pooling option the reference's ECAPA constructor can build
    (ecapa_tdnn_xvector.py:275-281; 'multi-head' / 'global-multi' / 'multi-resolution' die there with a TypeError and raise the same
    here): AttentiveStatisticsPooling with time context behind the MFA layer, eval BatchNorm over the pooled statistics.  Golden
    vectors: the reference's outputs (oracle/gen_golden.py ecapa_attentive)."""
    from libs.amd import ir
    g, sd, model = helpers.golden_model("ecapa_attentive")
    graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
    assert [op.kind for op in graph.ops if op.kind in ("attpool", "pool")][-1] == "attpool"
    for x, ref in zip(helpers.golden_feats(g), g["embeddings"]):
        assert rel_err(ir_interp.extract(graph, x), ref) < 2e-5


def test_batch_norm_over_a_pooled_tensor_in_device_column_order():
    """F.batch_norm applied to a pooled tensor whose device columns are a permutation (with alignment gaps) of the reference's -
    the global multi-head poolings' [mean_h | std_h] blocks - permutes its constants with the columns and keeps the gaps at zero:
    pooling -> BatchNorm1d -> affine traced in one graph equals the same affine applied to the hand-normalised pooled statistics."""
    import torch
    from libs.amd import ir
    from libs.nnet import GlobalMultiHeadAttentionPooling, TdnnAffine
    from libs.amd import synth
    C, H = 24, 3

    class Probe(torch.nn.Module):
        def __init__(self, with_bn):
            super().__init__()
            self.pool = GlobalMultiHeadAttentionPooling(C, num_head=H, share=True, affine_layers=1)
            self.bn = torch.nn.BatchNorm1d(2 * C * H) if with_bn else None
            self.out = TdnnAffine(2 * C * H, 8)

        def body(self, x):
            y = self.pool(x)
            if self.bn is not None:
                y = self.bn(y)
            return self.out(y)

    rs = np.random.RandomState(3)
    with_bn, plain = Probe(True), Probe(False)
    sd = {k: torch.from_numpy(rs.randn(*v.shape).astype(np.float32) * 0.3) for k, v in with_bn.state_dict().items() if "num_batches" not in k}
    sd["bn.running_var"] = sd["bn.running_var"].abs() + 0.5
    with_bn.load_state_dict(sd, strict=False)
    plain.load_state_dict({k: v for k, v in sd.items() if not k.startswith("bn.")}, strict=False)
    x = synth.synth_feats(57, C, 77)
    got = ir_interp.extract(ir.trace(with_bn, Probe.body, C), x)
    # the same by hand: pooled statistics in the REFERENCE's column order = what an identity affine reads
    eye = Probe(False)
    eye.load_state_dict({k: v for k, v in sd.items() if k.startswith("pool.")}, strict=False)
    eye.out = TdnnAffine(2 * C * H, 2 * C * H)
    with torch.no_grad():
        eye.out.weight.copy_(torch.eye(2 * C * H)[:, :, None]); eye.out.bias.zero_()
    stats = ir_interp.extract(ir.trace(eye, Probe.body, C), x).astype(np.float64)
    bn = {k[3:]: v.numpy().astype(np.float64) for k, v in sd.items() if k.startswith("bn.")}
    normed = (stats - bn["running_mean"]) / np.sqrt(bn["running_var"] + 1e-5) * bn["weight"] + bn["bias"]
    want = sd["out.weight"].numpy()[:, :, 0].astype(np.float64) @ normed + sd["out.bias"].numpy().reshape(-1)
    assert rel_err(got, want) < 1e-5


REF_MODEL_DIR = "/root/reference/pytorch/model"


@pytest.mark.skipif(not os.path.isdir(REF_MODEL_DIR), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("blueprint,creation,golden", [("xvector.py", "Xvector(30,10,training=False)", "xvector_c1"),
                                                       ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(80,10,training=False)", "ecapa_c3"),
                                                       ("resnet_xvector.py", "ResNetXvector(61,10,training=False,resnet_params={'full_pre_activation':False})", "resnet34_plain"),
                                                       ("resnet_xvector.py", "ResNetXvector(40,10,training=False,pooling='attentive',pooling_params={'hidden_size':32,'context':[-1,0,1]},"
                                                        "resnet_params={'layers':[1,1,1,1],'planes':[16,32,64,128],'use_se':True,'full_pre_activation':False})", "resnet_attentive"),
                                                       ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(40,10,training=False,pooling='attentive',pooling_params={'hidden_size':64,'context':[-1,0,1]},"
                                                        "ecapa_params={'channels':512,'embd_dim':128,'mfa_conv':768})", "ecapa_attentive"),
                                                       ("extended_xvector.py", "ExtendedXvector(40,10,training=False)", "extended_far"),
                                                       ("snowdar_xvector.py", "Xvector(40,10,training=False,extend=True,skip_connection=True,SE=True,extracted_embedding='near')",
                                                        "snowdar_full_near"),
                                                       ("factored_xvector.py", "Xvector(40,10,training=False)", "factored_far")])
def test_unmodified_reference_blueprints_run_on_this_libs_nnet(blueprint, creation, golden):
    """Drop-in check: the reference's OWN blueprint files import this package's `libs.nnet`,
    build, load the checkpoint keys and record to a program that reproduces the reference."""
    import subprocess
    import sys
    code = r'''
import sys, os, types
sys.dont_write_bytecode = True
for n, attrs in (("tkinter", {"N": None}), ("tkinter.messagebox", {"NO": None}), ("turtle", {"xcor": None})):
    m = types.ModuleType(n); m.__dict__.update(attrs); m.__path__ = []; sys.modules[n] = m
sys.path[:0] = [%(repo)r + "/tests", %(repo)r, %(repo)r + "/asv-subtools_amd/pytorch"]
import numpy as np, torch
import helpers, ir_interp
import libs.support.utils as utils
from libs.amd import ir
g, sd = helpers.golden_state_dict(%(golden)r)
model = utils.create_model_from_py(%(ref)r + "/" + %(bp)r, %(creation)r)
assert type(model).__module__ in ("xvector", "ecapa_tdnn_xvector", "resnet_xvector", "extended_xvector", "snowdar_xvector", "factored_xvector") and %(ref)r in sys.modules[type(model).__module__].__file__
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
graph = ir.trace(model, type(model).extract_embedding.__wrapped_body__, int(g["dim"]))
x = helpers.golden_feats(g)[-1]
err = helpers.rel_err(ir_interp.extract(graph, x), g["embeddings"][-1])
print("ERR", err)
assert err < 2e-5
''' % dict(repo=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ref=REF_MODEL_DIR, bp=blueprint, creation=creation, golden=golden)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPYCACHEPREFIX="/tmp/pyc_ref")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr


def test_space_to_depth_form_of_a_stride2_convolution_equals_the_convolution():
    """libs/nnet/resnet.py lowers 3 x 3 / stride 2 / pad 1 convolutions as a gather of the four input phases into the output
    grid followed by a 2 x 2 stride-1 convolution with re-arranged weights (s2d_kernel).  Plain numpy on the grid row layout
    (rows = (time, frequency), pitch = width + 1, zero gap column and gap rows): the form equals the convolution itself, odd
    sizes included."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch"))
    from libs.nnet.resnet import s2d_kernel
    r = np.random.RandomState(5)
    for T, F in ((7, 10), (8, 9), (1, 4), (5, 5)):
        cin, cout = 3, 4
        x = r.standard_normal((cin, F, T)).astype(np.float32)                 # [C, F, T] like the reference's [B, C, F, T]
        w = r.standard_normal((cout, cin, 3, 3)).astype(np.float32)            # [Cout, Cin, kF, kT]
        want = torch.nn.functional.conv2d(torch.from_numpy(x)[None], torch.from_numpy(w), stride=2, padding=1)[0].numpy()   # [Cout, F', T']
        Fo, To = (F + 1) // 2, (T + 1) // 2
        assert want.shape == (cout, Fo, To)
        pitch = Fo + 1
        # the gather (im2col_kernel with the four phase taps): rows (t', f'), zero outside the input / in the gap column
        halo = pitch + 2
        cols = np.zeros((halo + To * pitch + halo, 4 * cin), dtype=np.float32)
        for tp in range(To):
            for fp in range(Fo):
                for pt in (0, 1):
                    for pf in (0, 1):
                        t, f = 2 * tp + pt, 2 * fp + pf
                        if t < T and f < F:
                            cols[halo + tp * pitch + fp, (pt * 2 + pf) * cin:(pt * 2 + pf + 1) * cin] = x[:, f, t]
        taps, left, dense = s2d_kernel(w, pitch)
        got = np.zeros((cout, Fo, To), dtype=np.float32)
        for tp in range(To):
            for fp in range(Fo):
                row = halo + tp * pitch + fp
                acc = np.zeros(cout, dtype=np.float64)
                for tap in taps:
                    acc += dense[:, :, tap - left].astype(np.float64) @ cols[row + tap].astype(np.float64)
                got[:, fp, tp] = acc
        assert np.abs(got - want).max() < 1e-5, (T, F, np.abs(got - want).max())


def test_free_statistics_pooling_and_weight_normalised_affine_trace_like_their_plain_forms():
    """FreeStatisticsPooling (pooling.py:92-127) = StatisticsPooling without a declared width; TdnnAffine(norm_w=True)
    (components.py:139-143) = the affine with every (output, tap) weight column scaled to unit norm over its input channels:
    both are checked on the CPU interpreter of the layer program against their definition in torch."""
    import torch
    from libs.amd import ir
    from libs.nnet import TopVirtualNnet, TdnnAffine, FreeStatisticsPooling, for_extract_embedding

    class Tiny(TopVirtualNnet):
        def init(self, dim):
            self.a = TdnnAffine(dim, 24, context=[-2, 0, 2], norm_w=True)
            self.pool = FreeStatisticsPooling(stddev=True, unbiased=True)

        @for_extract_embedding(maxChunk=10000, isMatrix=True)
        def extract_embedding(self, inputs):
            return self.pool(self.a(inputs))

    torch.manual_seed(3)
    m = Tiny(10)
    with torch.no_grad():
        m.a.weight.normal_(0, 1.0); m.a.bias.normal_(0, 0.1)
    graph = ir.trace(m, type(m).extract_embedding.__wrapped_body__, 10)
    x = np.random.RandomState(0).standard_normal((37, 10)).astype(np.float32)
    got = ir_interp.extract(graph, x)
    with torch.no_grad():
        w = torch.nn.functional.normalize(m.a.weight * m.a.mask, dim=1)
        xin = torch.nn.functional.pad(torch.from_numpy(x.T)[None], (2, 2))
        y = torch.nn.functional.conv1d(xin, w, m.a.bias)
        want = torch.cat([y.mean(2), y.std(2, unbiased=True)], dim=1)[0].numpy()
    assert rel_err(got, want) < 1e-5


def test_host_half_and_bf16_conversions_match_numpy(tmp_path):
    """The weight packer's f32 -> IEEE half / bf16 conversions (csrc/host_convert.h, plain C++: compiled here with g++) against
    numpy's float16 (round-to-nearest-even, subnormals, overflow to infinity) and a bit-level bf16 reference, on random values
    of every magnitude plus the edge cases around the half range."""
    import subprocess
    src = tmp_path / "conv.cc"
    src.write_text('#include <stdio.h>\n#include "host_convert.h"\nint main() { float f; while (fread(&f, 4, 1, stdin) == 1) { unsigned short h = asv::f32_to_f16_host(f), '
                   'b = asv::f32_to_bf16_host(f); float back = asv::f16_to_f32_host(h); fwrite(&h, 2, 1, stdout); fwrite(&b, 2, 1, stdout); fwrite(&back, 4, 1, stdout); } return 0; }\n')
    exe = tmp_path / "conv"
    subprocess.check_call(["g++", "-O1", "-I", os.path.join(helpers.REPO, "asv-subtools_amd", "csrc"), str(src), "-o", str(exe)])
    r = np.random.RandomState(0)
    vals = np.concatenate([
        (r.randn(20000) * np.exp2(r.randint(-30, 18, 20000))).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 65536.0, -65520.0, 1e9, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 2.0 ** -25 * 1.0000001, 2.0 ** -26, 3 * 2.0 ** -25,
                  6.1e-5, 5.96e-8, np.inf, -np.inf, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20], dtype=np.float32),
        np.float16(r.randn(2000) * np.exp2(r.randint(-24, 15, 2000))).astype(np.float32)])      # exactly representable halves round-trip
    out = subprocess.run([str(exe)], input=vals.tobytes(), capture_output=True, check=True).stdout
    rec = np.frombuffer(out, dtype=np.dtype([("h", "<u2"), ("b", "<u2"), ("back", "<f4")]))
    with np.errstate(over="ignore"):
        want_h = vals.astype(np.float16)
    assert np.array_equal(rec["h"], want_h.view(np.uint16)), np.flatnonzero(rec["h"] != want_h.view(np.uint16))[:5]
    assert np.array_equal(rec["back"].view(np.uint32), want_h.astype(np.float32).view(np.uint32))
    u = vals.view(np.uint32).astype(np.uint64)
    want_b = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)                                # RNE on the upper 16 bits (no NaN in the set)
    assert np.array_equal(rec["b"], want_b)


def test_host_e4m3_conversion_matches_torch(tmp_path):
    """The f32m form's weight packer rounds the two weight corrections to OCP e4m3 on the host (csrc/host_convert.h f32_to_e4m3_host:
    round-to-nearest-even, subnormals down to 2^-9, saturating at +-448) - against torch's float8_e4m3fn on values inside its range,
    the saturation and every one of the 254 finite codes round-tripping."""
    import subprocess
    import torch
    src = tmp_path / "conv8.cc"
    src.write_text('#include <stdio.h>\n#include "host_convert.h"\nint main() { float f; while (fread(&f, 4, 1, stdin) == 1) { unsigned char b = asv::f32_to_e4m3_host(f); '
                   'float back = asv::e4m3_to_f32_host(b); fwrite(&b, 1, 1, stdout); fwrite(&back, 4, 1, stdout); } return 0; }\n')
    exe = tmp_path / "conv8"
    subprocess.check_call(["g++", "-O1", "-I", os.path.join(helpers.REPO, "asv-subtools_amd", "csrc"), str(src), "-o", str(exe)])
    r = np.random.RandomState(1)
    codes = np.array([c for c in range(256) if (c & 0x7f) != 0x7f], dtype=np.uint8)
    exact = torch.from_numpy(codes).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    vals = np.concatenate([
        np.clip((r.randn(20000) * np.exp2(r.randint(-14, 9, 20000))), -447.9, 447.9).astype(np.float32),
        np.array([0.0, -0.0, 448.0, 447.0, 464.0, 1e9, -1e9, np.inf, 2.0 ** -9, 2.0 ** -10, 2.0 ** -10 * 1.0001, 3 * 2.0 ** -10, 2.0 ** -6, 2.0 ** -6 - 2.0 ** -11,
                  1.0625, 1.1875, 0.0019, 15.5, 17.0], dtype=np.float32),
        exact])
    out = subprocess.run([str(exe)], input=vals.tobytes(), capture_output=True, check=True).stdout
    rec = np.frombuffer(out, dtype=np.dtype([("b", "u1"), ("back", "<f4")]))
    want = torch.from_numpy(np.clip(vals, -448.0, 448.0)).to(torch.float8_e4m3fn)
    assert np.array_equal(rec["b"], want.view(torch.uint8).numpy()), np.flatnonzero(rec["b"] != want.view(torch.uint8).numpy())[:8]
    assert np.array_equal(rec["back"], want.to(torch.float32).numpy())
    assert np.array_equal(rec["b"][-len(codes):] & 0x7f, codes & 0x7f) and np.array_equal(rec["back"][-len(codes):], exact)
