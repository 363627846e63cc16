#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py's headline steps: tools/prof_bench.sh <tag> [extra bench args]
# (both phases on one stream, so a kernel's duration is its own; run on the GPU box through gpurun; output under gpurun_out/prof_<tag>/, summary = kernel_stats.csv)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --one-stream "$@" > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
# steps in the trace: 5 warm-up + 50 timed + 3 + 50 of the instrumented pass = 108 (bench.py --steps 50 --warmup 5 --one-stream: no priming)
python3 $GRAFT_REPO_ROOT/tools/statsum.py $OUT/kernel_stats.csv 108 24
