#!/usr/bin/env python3
"""Host profile of p_sample at config 1's size (7850 x 14 latents, ball, T = 200): 1000 steps.  tools/probes/host_profile_ps.py"""
import cProfile
import os
import pstats
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402
from hypdiff_amd.models.direction_diffusion import diffusion  # noqa: E402

dev = "cuda"
n, d, T = 7850, 14, 200
rs = np.random.RandomState(0)
diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), d, num_timesteps=T).to(dev).eval()
img = torch.from_numpy(rs.standard_normal((n, d))).to(dev)
labels = ops.labels_dev(rs.randint(0, 3, size=n).astype(np.int64), dev)
u0 = torch.from_numpy((rs.standard_normal((3, d)) * 0.05).astype(np.float32)).to(dev)
with torch.no_grad():
    direction = diff.directions(u0)
    for i in range(20):
        diff.p_sample(img, None, 1, direction, labels, u0, 100, None, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        diff.p_sample(img, None, 1, direction, labels, u0, 100, None, True)
    e1.record()
    torch.cuda.synchronize()
    print("ms per step (no profiler):", e0.elapsed_time(e1) / 200)
    pr = cProfile.Profile()
    pr.enable()
    for i in range(1000):
        diff.p_sample(img, None, 1, direction, labels, u0, 100, None, True)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
