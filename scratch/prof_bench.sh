#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py (headline steps + training leg)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-overlap-leg --train-steps 0 --no-exact-leg > $OUT/log.txt 2>&1 || exit 1
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
head -40 $OUT/kernel_stats.csv | cut -c1-200
