#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - they do not fit one pass on gfx950,
MI355X_MICROARCH.md "rocprofv3 PMC slots") into profiles/pmc_summary.json.

    tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-name-substring> <out.json> [<libasv_amd.so> [<algorithmic bytes>]]

The optional library path stamps the summary with the build the counters were taken on (sha256 of the file: the same hash
__graft_entry__.smoke() prints); the optional byte count is recorded beside the measured traffic.

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  On gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read (guide, section HBM), so the read side is doubled; WRITE_SIZE is taken as is
(uncalibrated per the guide)."""

import csv
import json
import sys


def per_dispatch(path, counter, name_sub):
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter and name_sub in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_csv, write_csv, name_sub, out = sys.argv[1:5]
    lib = sys.argv[5] if len(sys.argv) > 5 else None
    alg = float(sys.argv[6]) if len(sys.argv) > 6 else None
    fetch = per_dispatch(fetch_csv, "FETCH_SIZE", name_sub)
    write = per_dispatch(write_csv, "WRITE_SIZE", name_sub)
    if not fetch or not write:
        sys.exit("no dispatches of %r with the counters found" % name_sub)
    fetch_b = 2.0 * 1024.0 * sum(fetch) / len(fetch)
    write_b = 1024.0 * sum(write) / len(write)
    info = {
        "kernel": name_sub, "dispatches": len(fetch),
        "fetch_bytes_per_launch_corrected": fetch_b, "write_bytes_per_launch": write_b,
        "traffic_bytes_per_launch": fetch_b + write_b,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), averaged over the kernel's dispatches",
    }
    if lib:
        import hashlib
        with open(lib, "rb") as f:
            info["library_sha256_16"] = hashlib.sha256(f.read()).hexdigest()[:16]
    if alg:
        info["algorithmic_bytes_per_launch"] = alg
        info["traffic_over_algorithmic"] = round((fetch_b + write_b) / alg, 3)
    with open(out, "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps(info))


if __name__ == "__main__":
    main()
