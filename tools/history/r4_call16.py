# round 4, call 16 (the last seconds of the GPU budget): ECAPA_TDNN(pooling='attentive') on the device against the reference's golden vectors
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "tests"), root, os.path.join(root, "asv-subtools_amd", "pytorch")]
import numpy as np
import helpers
g, sd, model = helpers.golden_model("ecapa_attentive")
model.cuda()
for prec in ("f32", "f32x", "bf16"):
    model.amd_precision = prec
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    ref = g["embeddings"]
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    print(prec, "rel_err", [helpers.rel_err(got[i], ref[i]) for i in range(len(ref))], "min cos", float(cos.min()))
