set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_mlp_chain.py tests/test_pipeline_gpu.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -2
timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_v5.json 2> gpurun_out/r03_bench_v5.err
python3 -c "
import json; j=json.load(open('gpurun_out/r03_bench_v5.json')); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['achieved'], j['one_stream']['ms_per_step'], j['train']['ms_per_step'], j['sharded']['ms_per_step'], j['cpu_baseline']['value'])"
bash tools/prof_bench.sh r03d > gpurun_out/r03d_prof_summary.txt 2>&1; tail -14 gpurun_out/r03d_prof_summary.txt
bash tools/pmc_bench.sh r03d_bench; python3 -c "
import json; j=json.load(open('gpurun_out/r03d_bench_hbm_traffic.json')); [print(k[:50], v) for k,v in j['kernels'].items() if 'chain' in k or 'gemm' in k]"
