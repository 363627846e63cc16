set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_train.py tests/test_mlp_chain.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3
for r in 1 2; do timeout -k 10 200 python3 tools/prof_train.py 40 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train5
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_train.py 40 > $OUT/log.txt 2>&1 || { tail -20 $OUT/log.txt; exit 1; }
cp $(ls $OUT/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/statsum.py $OUT/kernel_stats.csv 45 24
