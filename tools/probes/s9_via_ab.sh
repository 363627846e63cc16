cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in 0 1; do
HD_GEMM_VIA_CHAIN=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary --one-stream > /tmp/ab.json 2>/dev/null || exit 1
python3 -c "
import json,sys; j=json.load(open('/tmp/ab.json')); print('via_chain=$v one-stream', round(j['ms_per_step'],4), 'encode', round(j['breakdown']['encode_ms'],4), 'p_sample', round(j['breakdown']['p_sample_ms'],4), {k: round(v['ms_per_step'],4) for k,v in j['roofline']['other_gemm_kernels'].items()})"
HD_GEMM_VIA_CHAIN=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/dev/null || exit 1
python3 -c "
import json,sys; j=json.load(open('/tmp/ab.json')); print('via_chain=$v pipelined ', round(j['ms_per_step'],4))"
done; done
