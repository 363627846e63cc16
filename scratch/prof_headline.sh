#!/bin/bash
# rocprofv3 kernel-trace summary of the headline steps only (no secondary legs)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-exact-leg --no-overlap-leg --train-steps 0 > $OUT/log.txt 2>&1 || exit 1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    n=re.sub(r"\(anonymous namespace\)::|void |hd::","",r["Name"])[:70]
    print(f'{n:70s} calls {int(r["Calls"]):5d} avg_us {float(r["AverageNs"])/1e3:8.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f} {float(r["Percentage"]):5.1f}%')
PY
