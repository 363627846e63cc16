#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

    python tools/pmc_summary.py gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE \
        > profiles/rNN_bench_hbm_traffic.json

Per-launch means.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024
is exact for 16-byte streaming stores.  Both counters are in KiB."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:90]


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    out = {"note": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        if k.startswith(("at::", "__amd")) or fe[k][0] == 0:
            continue
        n = fe[k][0]
        f_kb = fe[k][1] / n
        w_kb = wr[k][1] / max(1, wr[k][0]) if k in wr else 0.0
        out["kernels"][k] = {"launches": n, "FETCH_SIZE_KB": round(f_kb), "WRITE_SIZE_KB": round(w_kb),
                             "hbm_bytes_corrected": int(2 * f_kb * 1024 + w_kb * 1024)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
