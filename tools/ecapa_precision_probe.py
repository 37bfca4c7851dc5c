import sys, os
sys.path[:0] = ["/root/repo/asv-subtools_amd/pytorch", "/root/repo", "/root/repo/tests"]
import numpy as np, torch, helpers
from libs.amd import synth
model = helpers.build_model("ecapa_tdnn_xvector.py", "ECAPA_TDNN(80,10,training=False)")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.cuda()
base = synth.synth_feats(1500, 80, 700000)
for noise in (0.2, 1.0):
    mats = []
    for T in (300, 600, 1000, 1500):
        for k in range(3):
            r = np.random.RandomState(T + k)
            mats.append((base[:T] + noise * r.standard_normal((T, 80))).astype(np.float32))
    out = {}
    for prec in ("f32", "f32x", "bf16"):
        model.amd_precision = prec
        out[prec] = model.extract_embedding_batch(mats).numpy()
    for p in ("f32x", "bf16"):
        cos = (out[p] * out["f32"]).sum(1) / np.linalg.norm(out[p], axis=1) / np.linalg.norm(out["f32"], axis=1)
        rel = np.abs(out[p] - out["f32"]).max(1) / np.abs(out["f32"]).max(1)
        print("noise", noise, p, "cos", np.round(cos, 5).tolist(), "rel", np.round(rel, 4).tolist())
