# round 5 diagnostic: which engine / order makes utt003 of the huge-feature batch wrong in the sharded order?
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "tests"), root, os.path.join(root, "asv-subtools_amd", "pytorch")]
import numpy as np, warnings
import helpers
from oracle import np_oracle as O
g, sd, model = helpers.golden_model("xvector_near_ragged")
model.cuda()
mats = [m.copy() for m in helpers.golden_feats(g)]
for i in (1, 4, len(mats) - 1):
    mats[i] = (mats[i] * 1.0e5).astype(np.float32)
want = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in mats])
for order in ([0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [4, 3], [3, 4], [5, 3], [4, 3, 2]):
    for prec in ("f32", "f32x-bf16", "f32x", "bf16"):
        model.amd_precision = prec
        eng = model._amd_engine()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = eng._extract_batch([mats[i] for i in order]).numpy()
        print(order, prec, ["%d:%.2g" % (i, helpers.rel_err(got[j], want[i])) for j, i in enumerate(order)], flush=True)
