#!/bin/bash
# Fabric read traffic (L2 misses) of the weight-gradient kernels, five launches against one: tools/probes/pmc_tn.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tn
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_tn.py 4 > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_tn_img" in r["Kernel_Name"]:
            k = r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "")
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k}: launches {n}, mean FETCH_SIZE x2 = {2 * v / n / 1024:.1f} MiB")
PY
