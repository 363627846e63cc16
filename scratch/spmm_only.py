import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, "tools/bench_kernels.py"], capture_output=True, text=True).stdout
for l in out.splitlines():
    if "spmm" in l or "agg" in l or "hyp_act" in l: print(l[:170])
