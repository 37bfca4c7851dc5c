#!/usr/bin/env python3
"""Diagnostic (GPU): ResNet34-SE in the 16-bit modes on utterances of growing input scale, alone and beside ordinary neighbours,
against the f64 oracle - where does the bf16 error of tests/test_gpu_neighbour_independence.py's 1e5-scaled utterance come from?"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "asv-subtools_amd", "pytorch"), os.path.join(REPO, "tests")]
import helpers
from helpers import rel_err
from libs.amd import synth
from oracle import np_oracle as O

g, sd, model = helpers.golden_model("resnet34se_c5")
model.cuda()
sd64 = O.cast_state_dict(sd, np.float64)
fn = lambda c: O.resnet_embed(c, sd64, "near", "", preact=False)
base = synth.synth_feats(64, 80, 8802)
other = synth.synth_feats(88, 80, 8801)
for scale in (1.0, 1e1, 1e2, 1e3, 1e4, 1e5):
    m = (base * np.float32(scale)).astype(np.float32)
    want = O.extract_embedding(fn, m, dtype=np.float64)
    taps = {}
    line = "scale %-8g |want| %.3g " % (scale, np.abs(want).max())
    for prec in ("f32", "bf16", "f16", "f32x-bf16"):
        model.amd_precision = prec
        eng = model._amd_engine()
        alone = eng._extract_batch([m]).numpy()[0]
        beside = eng._extract_batch([other, m, other]).numpy()[1]
        cos = float((alone * want).sum() / np.linalg.norm(alone) / np.linalg.norm(want))
        line += " | %s alone %.3g (cos %.5f) beside-vs-alone %.3g" % (prec, rel_err(alone, want), cos, rel_err(beside, alone))
    print(line, flush=True)
