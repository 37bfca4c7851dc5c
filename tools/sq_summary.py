#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 SQ-counter pass (tools/r6_collect.sh):

    tools/sq_summary.py <sq_counter_collection.csv> <kernel_trace.csv> <out.json> [--clock <kernel substring>=<MHz> ...]

What the counters can and cannot say on gfx950 (VERDICT r5 item 8 - the round-5 version of this tool printed two derived columns that were
not measurements):

  * SQ_VALU_MFMA_BUSY_CYCLES is, to the last count, 32 x the number of v_mfma_f32_32x32x16 instructions the dispatch issued (64 per
    scaled K = 64 instruction): a WORK count, summed over all SIMDs.  Dividing it by (SIMDs x duration x 2.4 GHz) is FLOP-rate / peak by
    construction - reported here as `matrix_cycles_per_us` (raw) and nothing more.
  * GRBM_GUI_ACTIVE / 8 / duration is NOT the shader clock (it gave 2.8 GHz for a matrix kernel and 13.5 GHz for a copy kernel on a 2.4 GHz
    part): not printed any more.
  * The shader clock under load comes from INSIDE a kernel: s_memtime (shader cycles) against s_memrealtime (100 MHz) around the whole
    workgroup - ASV_AMD_CHAIN_DBG=1 prints it for tdnn_chain_kernel / tdnn_chainm_kernel ("shader clock 1840 MHz"), tools_p8_probe for the
    8-phase kernels.  Pass it with --clock and the matrix pipe's busy share of the kernel's own cycles is
        matrix_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock);
    without a measured clock that column is absent.
  * SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY are quad-cycles summed over waves: reported as fractions of SQ_WAVE_CYCLES (parked at
    s_waitcnt / barrier, issue-stalled); SQ_LDS_BANK_CONFLICT as a share of SQ_LDS_IDX_ACTIVE.
"""
import collections
import csv
import json
import sys


def main():
    args = sys.argv[1:]
    clocks = {}
    while "--clock" in args:
        i = args.index("--clock")
        name, mhz = args[i + 1].split("=")
        clocks[name] = float(mhz)
        del args[i:i + 2]
    cc, trace, out = args[:3]
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
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in m:
            continue
        wave = m.get("SQ_WAVE_CYCLES", float("nan"))
        rec = {
            "dispatches": len(c["SQ_VALU_MFMA_BUSY_CYCLES"]),
            "avg_us_under_counters": round(m["_us"], 2),
            "matrix_cycles": round(m["SQ_VALU_MFMA_BUSY_CYCLES"]),
            "matrix_instructions_32x32x16_equivalent": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0),
            "matrix_cycles_per_us": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["_us"], 1),
            "wave_parked_frac": round(m.get("SQ_WAIT_ANY", float("nan")) / wave, 4),
            "wave_issue_stall_frac": round(m.get("SQ_WAIT_INST_ANY", float("nan")) / wave, 4),
        }
        if m.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share_of_lds_active"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        for sub, mhz in clocks.items():
            if sub in name:
                rec["shader_clock_mhz_from_in_kernel_stamps"] = mhz
                rec["matrix_pipe_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["_us"] * mhz), 4)
        res[name] = rec
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
