#!/usr/bin/env python3
"""Developer tool: registers / spills / occupancy / LDS of every kernel of one translation unit, from hipcc's
-Rpass-analysis=kernel-resource-usage (build container; no GPU needed).    tools/kernel_resources.py kernels_tdnn_chain [filter]"""
import re, subprocess, sys, os
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asv-subtools_amd", "csrc")
unit = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] if unit in ("kernels_tdnn_x3", "kernels_tdnn_chainx", "kernels_tdnn_chain4") else []
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(csrc, "..", "..", "include"), "-I" + csrc] + extra + \
      ["-c", os.path.join(csrc, unit + ".hip") if os.path.exists(os.path.join(csrc, unit + ".hip")) else os.path.join(csrc, "tools", unit + ".hip"),
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + (["-DASV_WITH_ABLATION"] if unit == "kernels_tdnn_chain4" else [])
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"asv::\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][\w \[\]/]*?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-70s VGPR %3d AGPR %3d spill %3d SGPR %3d occ %d LDS %6d scratch %d" % (k[:70], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1),
              v.get("TotalSGPRs", -1), v.get("Occupancy [waves/SIMD]", -1), v.get("LDS Size [bytes/block]", -1), v.get("ScratchSize [bytes/lane]", -1)))
