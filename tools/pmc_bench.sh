#!/bin/bash
# HBM traffic counters of bench.py: one rocprofv3 --pmc pass per counter (FETCH_SIZE takes 3 of the 4
# TCC slots, WRITE_SIZE 2), kernel trace only.  tools/pmc_bench.sh <tag> [extra bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$c
  rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --one-stream "$@" > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_WRITE_SIZE > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_hbm_traffic.json
