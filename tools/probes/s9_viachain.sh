set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_mlp_chain.py -x -q --timeout 120 --timeout-method thread -m gpu -k "chain_of_one" > gpurun_out/s9_vc_test.log 2>&1 || { tail -30 gpurun_out/s9_vc_test.log; exit 1; }
tail -3 gpurun_out/s9_vc_test.log
bash tools/probes/bench_ab.sh 2 HD_GEMM_VIA_CHAIN=0 HD_GEMM_VIA_CHAIN=1
