#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$c
  mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1 || exit 1
done
