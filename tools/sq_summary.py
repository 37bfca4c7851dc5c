#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 SQ-counter pass (tools/collect_profiles.sh):

    tools/sq_summary.py <sq_counter_collection.csv> <kernel_trace.csv> <out.json>

Normalisation used on gfx950 (MI355X: 8 XCDs x 32 CUs x 4 SIMDs):
  * GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: busy shader cycles of the dispatch = value / 8,
    and value / 8 / duration is the average shader clock while the kernel ran;
  * SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over all SIMDs (32 per
    v_mfma_f32_32x32x16_bf16): utilisation = value / (1024 SIMDs x busy shader cycles);
  * SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY are quad-cycles summed over waves: reported as fractions
    of SQ_WAVE_CYCLES (parked at s_waitcnt / barrier, issue-stalled).
"""
import collections
import csv
import json
import sys


def main():
    cc, trace, out = sys.argv[1:4]
    dur = {}
    with open(trace) as f:
        for r in csv.DictReader(f):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3   # us
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(cc) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[name]["_us"].append(dur.get(r["Dispatch_Id"], float("nan")))
    res = {}
    for name, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if "GRBM_GUI_ACTIVE" not in m or "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
            continue
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        wave = m.get("SQ_WAVE_CYCLES", float("nan"))
        res[name] = {
            "dispatches": len(c["GRBM_GUI_ACTIVE"]),
            "avg_us_under_counters": round(m["_us"], 2),
            "shader_clock_ghz": round(cycles / m["_us"] * 1e-3, 3),
            "mfma_util": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles), 4),
            "mfma_util_vs_2p4ghz_peak": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["_us"] * 2400.0), 4),
            "wave_parked_frac": round(m.get("SQ_WAIT_ANY", float("nan")) / wave, 4),
            "wave_issue_stall_frac": round(m.get("SQ_WAIT_INST_ANY", float("nan")) / wave, 4),
            "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
            "lds_active_frac_per_cu": round(m.get("SQ_LDS_IDX_ACTIVE", float("nan")) / (256.0 * cycles), 4),
        }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
