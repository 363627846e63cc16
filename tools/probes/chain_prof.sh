#!/bin/bash
# rocprofv3 kernel durations of the one-launch chain and the per-layer kernels (tools/probes/chain_scaling.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_chain
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/probes/chain_scaling.py > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
f=$(ls $OUT/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
tail -16 $OUT/log.txt
