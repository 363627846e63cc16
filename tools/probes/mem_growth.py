#!/usr/bin/env python3
"""Does anything keep allocating?  Device allocations (hipMalloc count, reserved MiB) of the caching allocator over 300
training steps, 300 one-stream and 300 pipelined bench steps.  tools/probes/mem_growth.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def stats():
    st = torch.cuda.memory_stats()
    return st.get("num_device_alloc", 0), st.get("reserved_bytes.all.current", 0) >> 20, st.get("allocated_bytes.all.current", 0) >> 20


w = bench.build_workload(512, 1000, 1432, "cuda:0")
print("start", stats(), flush=True)
with torch.no_grad():
    for phase in ("one stream", "pipelined"):
        if phase == "pipelined":
            w["side_stream"] = torch.cuda.Stream()
        for i in range(300):
            bench.one_step(w, 900 - (i % 800), None, i)
            if i in (0, 9, 99, 299):
                print(phase, "step", i, stats(), flush=True)
        torch.cuda.synchronize()
w.pop("side_stream", None)
r = bench.train_leg(w, 512, "cuda:0", 10, 3, torch.cuda.synchronize, None)
print("train 13 steps", stats(), flush=True)
r = bench.train_leg(w, 512, "cuda:0", 300, 3, torch.cuda.synchronize, None)
print("train 300 more", stats(), r["ms_per_step"], flush=True)
