#!/bin/bash
# timing, then fabric read traffic per gang size: tools/probes/chain_gang.sh 1,2,3,4,8
set -e
G=${1:-1,2,4}
timeout -k 10 300 python3 $GRAFT_REPO_ROOT/tools/probes/chain_gang.py $G
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gang
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probes/chain_gang.py $G 4 > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "mlp_chain" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
gs = "$G".split(",")
for i, g in enumerate(gs):
    v = [x for _, x in rows[4 * i: 4 * i + 4]]
    print(f"gang {g}: FETCH_SIZE x2 = {[round(2 * x / 1024) for x in v]} MiB")
PY
