#!/usr/bin/env python3
"""Is the limb GEMM of the encode (15 777 x 512 x 512) slower when its A image is not in the Infinity Cache?  Events around
single launches, with and without a 1 GiB fill between them, and with the image freshly written by limb_pack just before
(as row_pack does in the step).  tools/probes/gemm_cold.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402

M, K, N = 15777, 512, 512
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
wl, _ = ops.weight_limbs(w, "bf16x3")
al = ops.limb_pack(x)
out = torch.empty(M, N, device="cuda")
junk = torch.empty(256 << 20, device="cuda")


def run(mode, reps=30):
    ts = []
    for _ in range(reps):
        a = al
        if mode == "flush":
            junk.fill_(1.0)
        elif mode == "fresh":
            junk.fill_(1.0)
            a = ops.limb_pack(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_limbs(a, wl, None, epi=0, want_f32=True, want_limbs=False, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for via in (False, True):
    ops.GEMM_VIA_CHAIN = via
    for mode in ("warm", "flush", "fresh"):
        run(mode, 5)
        print(f"via_chain={via} {mode:6s}: median {run(mode):6.1f} us", flush=True)
