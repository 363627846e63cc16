#!/usr/bin/env python3
"""The denoiser training step of bench.py's `train` leg alone (for rocprofv3 --kernel-trace --stats):
tools/prof_train.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    w = bench.build_workload(512, 1000, 1432, "cuda:0")
    sync = torch.cuda.synchronize
    r = bench.train_leg(w, 512, "cuda:0", steps, 5, sync, None)
    print({k: r[k] for k in ("ms_per_step", "value", "final_loss")})


if __name__ == "__main__":
    main()
