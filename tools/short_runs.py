import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hypdiff_amd import ops
torch.cuda.set_device(0)
w = bench.build_workload(512, 1000, 1432, "cuda:0")
def barrier(): torch.cuda.synchronize()
T = 1000
el, _, _ = bench.timed_steps(w, T, 100, 3, barrier, ops, None, "cuda:0")
print("one-stream 100:", round(el / 100 * 1e3, 4))
w["side_stream"] = torch.cuda.Stream()
for i in range(4):
    el, _, _ = bench.timed_steps(w, T, 20, 5, barrier, ops, None, "cuda:0")
    print("two-stream 5+20 #%d:" % i, round(el / 20 * 1e3, 4), flush=True)
el, _, _ = bench.timed_steps(w, T, 200, 20, barrier, ops, None, "cuda:0")
print("two-stream 20+200:", round(el / 200 * 1e3, 4))
for i in range(2):
    el, _, _ = bench.timed_steps(w, T, 20, 5, barrier, ops, None, "cuda:0")
    print("two-stream 5+20 again #%d:" % i, round(el / 20 * 1e3, 4), flush=True)
