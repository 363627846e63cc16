#!/bin/bash
# Same-box A/B of library builds (or of one environment setting, VAR=VALUE) in the bench step:
# tools/probes/bench_ab.sh <rounds> <lib | default | VAR=VALUE> ...
# (each run: bench.py --no-cpu-baseline --no-secondary; prints ms per step, roofline frac, TF, p_sample ms)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    unset HD_LIB_PATH; E=""
    case "$v" in default) ;; *=*) E="$v" ;; *) export HD_LIB_PATH=$GRAFT_REPO_ROOT/$v ;; esac
    env $E timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > /tmp/ab.json 2>/dev/null || exit 1
    python3 -c "
import json,sys; j=json.load(open('/tmp/ab.json')); print(sys.argv[1].ljust(34), j['ms_per_step'], round(j['roofline']['frac'],4), round(j['roofline']['achieved'],1), round(j['breakdown']['p_sample_ms'],4))" "$v"
  done
done
