#!/usr/bin/env python3
"""Per-step duration of the headline (two-stream, pipelined across two batches) arrangement right after 100 one-stream
steps, as bench.py orders its legs: HIP events on the main stream at every step boundary.  Shows how long the
arrangement takes to reach its steady state.  tools/probes/two_stream_ramp.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
w = bench.build_workload(512, 1000, 1432, "cuda:0")
early = len(sys.argv) > 2 and sys.argv[2] in ("early", "keepalive")   # create the side stream (and touch it) before the one-stream steps
keepalive = len(sys.argv) > 2 and sys.argv[2] == "keepalive"           # ... and launch a tiny kernel on it in every one-stream step
if early:
    side0 = torch.cuda.Stream()
    with torch.cuda.stream(side0):
        torch.zeros(1024, device="cuda:0").add_(1)
with torch.no_grad():
    tiny = torch.zeros(1024, device="cuda:0")
    for i in range(100):
        if keepalive:
            with torch.cuda.stream(side0):
                tiny.add_(1)
        bench.one_step(w, 500 - (i % 400), None, i)
    torch.cuda.synchronize()
    w["side_stream"] = side0 if early else torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    mallocs = []
    for i in range(steps):
        bench.one_step(w, 500 - (i % 400), None, i)
        ev[i + 1].record()
        st = torch.cuda.memory_stats()
        mallocs.append((st.get("num_device_alloc", 0), st.get("reserved_bytes.all.current", 0) >> 20))
    torch.cuda.synchronize()
d = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
for a in range(0, steps, 10):
    print(f"steps {a:3d}-{a + 9:3d}: " + " ".join(f"{v:.3f}" for v in d[a:a + 10]), flush=True)
print("device allocations (count, reserved MiB) after steps 0, 1, 2, 5, 10, 20, 40, last:",
      [mallocs[i] for i in (0, 1, 2, 5, 10, 20, 40, steps - 1) if i < steps])
