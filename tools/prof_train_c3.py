#!/usr/bin/env python3
"""25 eager training steps of ddpm.Model at config 3's shape (for rocprofv3 --kernel-trace --stats): B=200 tools/prof_train_c3.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd.ddpm import Model
from hypdiff_amd.dist import DataParallelTrainer
dev = "cuda"
B, D, Hd, L, T = int(os.environ.get("B", 200)), 64, 512, 10, 1000
torch.manual_seed(0)
model = Model(D, Hd, L, T, 1e-4, 0.02).to(dev).train()
tr = DataParallelTrainer(model, lr=1e-3)
x = torch.randn(B, D, device=dev)
for _ in range(25): tr.step(lambda: model.loss_fn(x))
torch.cuda.synchronize()
