#!/usr/bin/env python3
"""Developer aid: in-process interleaved A/B of kernel variants selected by an environment variable the runtime reads at every
launch (boxes of the pool differ by +-5 %, run-to-run noise hides anything smaller; cdna_hip_programming.md 5.4 rule 24).

    tools/chain_ab.py ENVVAR v1 v2 ... [--batch 640] [--rounds 7] [--model xvector|ecapa]"""
import os, sys, time
os.environ["ASV_AMD_LIVE_TUNE"] = "1"            # the runtime then reads its developer switches at every launch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
import helpers
from libs.amd import synth
args = sys.argv[1:]
def opt(name, default):
    if name in args:
        i = args.index(name); v = args[i + 1]; del args[i:i + 2]; return v
    return default
B, rounds, kind = int(opt("--batch", 640)), int(opt("--rounds", 7)), opt("--model", "xvector")
T = int(opt("--frames", 300 if kind == "ecapa" else 200))
env, variants = args[0], args[1:]
bp, creation = {"xvector": ("xvector.py", "Xvector(80,10,training=False)"), "ecapa": ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(80,10,training=False)"),
                "resnet": ("resnet_xvector.py", "ResNetXvector(80,10,training=False,resnet_params={'use_se':True,'se_ratio':4,'full_pre_activation':False})")}[kind]
model = helpers.build_model(bp, creation)
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.cuda(); model.amd_precision = os.environ.get("AB_PRECISION", "bf16")
eng = model._amd_engine()
feats = torch.from_numpy(np.concatenate([synth.synth_feats(T, 80, i) for i in range(B)])).cuda()
offs = (np.arange(B + 1) * T).astype(np.int32)
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.extract_device(feats, offs)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
run(300)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        os.environ[env] = v
        run(20); res[v].append(run(150))
for v in variants:
    a = sorted(res[v]); print("%s=%s: median %.1f us/step  min %.1f  max %.1f  (%.0f utt/s)" % (env, v, 1e6 * a[len(a) // 2], 1e6 * a[0], 1e6 * a[-1], B / a[len(a) // 2]))
# the variants must agree bit for bit unless the experiment says otherwise
outs = {}
for v in variants:
    os.environ[env] = v
    outs[v] = eng.extract_device(feats, offs).float().cpu().numpy().copy()
for v in variants[1:]:
    d = np.abs(outs[v] - outs[variants[0]])
    print("%s=%s vs %s: max |diff| %.3g (%s)" % (env, v, variants[0], d.max(), "bit-identical" if np.array_equal(outs[v], outs[variants[0]]) else "DIFFERENT"))
