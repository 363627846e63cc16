import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hypdiff_amd import ops
torch.cuda.set_device(0)
w = bench.build_workload(512, 1000, 1432, "cuda:0")
def run(steps, ev):
    with torch.no_grad():
        for i in range(5): bench.one_step(w, 999 - i)
        torch.cuda.synchronize()
        ops.GEMM_EVENTS = [] if ev else None
        ph = [] if ev == 2 else None
        t0 = time.perf_counter()
        for i in range(steps): bench.one_step(w, 999 - i, ph)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ops.GEMM_EVENTS = None
    return dt / steps * 1e3
for ev in (0, 1, 2, 0, 2):
    print("events", ev, "ms/step %.4f" % run(200, ev), flush=True)
