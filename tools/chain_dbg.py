#!/usr/bin/env python3
"""Developer aid: phase durations of the layer-chain kernel (ASV_AMD_CHAIN_DBG=1 makes the runtime print the mean shader-clock
cycles between the kernel's phase stamps after every chain launch)."""
import os, sys
os.environ["ASV_AMD_CHAIN_DBG"] = sys.argv[2] if len(sys.argv) > 2 else "1"      # 2: + raw per-wave timelines of two workgroups
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
import helpers
from libs.amd import synth
model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); model.cuda(); model.amd_precision = "bf16"
eng = model._amd_engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 640
feats = torch.from_numpy(np.concatenate([synth.synth_feats(200, 80, i) for i in range(B)])).cuda()
offs = (np.arange(B + 1) * 200).astype(np.int32)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 200): eng.extract_device(feats, offs)      # warm clocks (prints each time; keep the last lines)
torch.cuda.synchronize()
