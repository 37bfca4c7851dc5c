import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for q in (R, os.path.join(R, "asv-subtools_amd", "pytorch"), os.path.join(R, "tests")): sys.path.insert(0, q)
os.environ["ASV_AMD_LIVE_TUNE"] = "1"
import numpy as np, torch
import helpers
from libs.amd import capi, synth
L = capi.lib()
model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
model.cuda(); model.amd_precision = "f32m"
eng = model._amd_engine()
ids = dict(P8=1, BIG3=2, P8X=3, CHAINM=4, X3M=5, IMG=6)
def counts(): return {k: int(L.asv_kernel_launch_count(v)) for k, v in ids.items()}
for lens in ([200] * 256, [200] * 255 + [37]):
    mats = [synth.synth_feats(T, 80, 5000 + i) for i, T in enumerate(lens)]
    for img in ("1", "0", "1"):
        os.environ["ASV_AMD_X3M_IMAGE"] = img
        a = counts()
        out = eng._extract_batch(mats).numpy()
        b = counts()
        print(len(lens), lens[-1], "image", img, {k: b[k] - a[k] for k in ids}, float(np.abs(out).max()), flush=True)
