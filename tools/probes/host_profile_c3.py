#!/usr/bin/env python3
"""Where the host's time goes in config 3's eager training step (ddpm.Model, 200 rows, ~180 launches): cProfile of 300 steps.
tools/probes/host_profile_c3.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd.ddpm import Model  # noqa: E402
from hypdiff_amd.dist import DataParallelTrainer  # noqa: E402

dev = "cuda"
B, D, Hd, L, T = 200, 64, 512, 10, 1000
model = Model(D, Hd, L, T, 1e-4, 0.02).to(dev).train()
x = torch.randn(B, D, device=dev)
tr = DataParallelTrainer(model, lr=1e-3)
for _ in range(20):
    tr.step(lambda: model.loss_fn(x))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    tr.step(lambda: model.loss_fn(x))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
