set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_mlp_chain.py tests/test_train.py -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/s9_tn_test.log 2>&1 || { tail -30 gpurun_out/s9_tn_test.log; exit 1; }
tail -3 gpurun_out/s9_tn_test.log
timeout -k 10 200 python3 tools/bench_tn.py 50 > gpurun_out/s9_bench_tn.log 2>&1 || { tail -30 gpurun_out/s9_bench_tn.log; exit 1; }
tail -8 gpurun_out/s9_bench_tn.log
for r in 1 2; do
for m in 1 0; do
HD_TN_MULTI=$m timeout -k 10 200 python3 tools/prof_train.py 40 2>&1 | tail -1 | sed "s/^/multi=$m /"
done; done
bash tools/probes/pmc_tn.sh
