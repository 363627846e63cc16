#!/usr/bin/env python3
"""hd_limb_pack_multi_f32 on the denoiser's weight shapes: all ten images, the five W images, the five W^T images.
tools/probes/pack_multi_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402

dims = [513, 800, 900, 800, 700, 512]
ws = [torch.randn(dims[i + 1], dims[i], device="cuda") for i in range(5)]


def timeit(items, reps=200):
    for _ in range(5):
        ops.limb_pack_multi(items)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.limb_pack_multi(items)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("all ten   : %.1f us" % timeit([(w, False) for w in ws] + [(w, True) for w in ws]))
print("five W    : %.1f us" % timeit([(w, False) for w in ws]))
print("five W^T  : %.1f us" % timeit([(w, True) for w in ws]))
print("one W     : %.1f us" % timeit([(ws[1], False)]))
print("one W^T   : %.1f us" % timeit([(ws[1], True)]))
