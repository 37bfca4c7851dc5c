#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection csv (one row per dispatch) into one row per kernel:
kernel, dispatches, average counter value per dispatch as reported (FETCH_SIZE / WRITE_SIZE: KiB; FETCH_SIZE x 2 on gfx950 for bytes,
/opt/skills/guides/MI355X_MICROARCH.md).  The raw files are MBs each and stay in gpurun_out/; profiles/ keeps the aggregate.

    python tools/pmc_per_kernel.py gpurun_out/r5z_pmc_xvector_bf16_FETCH_SIZE.csv profiles/r5z_pmc_xvector_bf16_FETCH_SIZE_per_kernel.csv
"""
import csv
import re
import sys
from collections import OrderedDict


def main(src, dst):
    acc = OrderedDict()
    counter = None
    with open(src, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\((asv::|int|float|void|unsigned|const|long|char|bool|\)).*$", "", name)      # drop the argument list, keep the template arguments
            counter = row["Counter_Name"]
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    unit = "_KiB" if counter in ("FETCH_SIZE", "WRITE_SIZE") else ""
    with open(dst, "w") as g:
        g.write("kernel,dispatches,avg_%s%s_per_dispatch_as_reported\n" % (counter, unit))
        for name, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            g.write('"%s",%d,%.1f\n' % (name, n, tot / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
