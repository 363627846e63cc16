#!/bin/bash
# The round-end measurement set on one box: GPU suite with the margin table, smoke, bench (defaults and the driver's flags),
# rocprofv3 kernel stats, PMC traffic.  gpurun -- bash tools/final_state.sh
set -e
cd $GRAFT_REPO_ROOT
bash tools/parity_margin.sh
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python3 bench.py > gpurun_out/r03_bench_v8.json 2> gpurun_out/r03_bench_v8.err
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_v8_driver_flags.json 2>> gpurun_out/r03_bench_v8.err
python3 -c "
import json
for f in ('gpurun_out/r03_bench_v8.json','gpurun_out/r03_bench_v8_driver_flags.json'):
    j=json.load(open(f)); print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['achieved'], j['one_stream']['ms_per_step'], j['train']['ms_per_step'], j['sharded']['ms_per_step'], j['cpu_baseline']['value'])"
bash tools/prof_bench.sh r03g > gpurun_out/r03g_prof_summary.txt 2>&1; head -12 gpurun_out/r03g_prof_summary.txt | cut -c1-130
bash tools/pmc_bench.sh r03g_bench; python3 -c "
import json; j=json.load(open('gpurun_out/r03g_bench_hbm_traffic.json')); [print(k[:50], v) for k,v in j['kernels'].items() if 'chain' in k or 'gemm' in k]"
