#!/usr/bin/env python3
"""HIP-event timing of the three GEMM forms on the denoiser shapes (M = 15 777 rows):
forward NT (x @ W^T, bias+GELU), input gradient (dz @ W with the GELU' epilogue), weight
gradient TN (dz^T @ x).  One JSON line per shape; peak = 157.3 TFLOP/s dense fp32 matrix."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import autograd as hag, ops  # noqa: E402

PEAK = 157.3


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 15777
    dev = "cuda"
    shapes = [(516, 800), (800, 900), (900, 800), (800, 700), (700, 512), (512, 512), (4096, 4096)]
    for K, N in shapes:
        m = 4096 if K == 4096 else M
        x = torch.randn(m, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        z = torch.randn(m, K, device=dev)
        dz = torch.randn(m, N, device=dev)
        out = torch.empty(m, N, device=dev)
        g = torch.empty(m, K, device=dev)
        wt = w.t().contiguous()
        fl = 2.0 * m * N * K
        t_nt = timeit(lambda: hag.gemm_ex(x, w, b, None, None, out, None, 1))
        t_tr = timeit(lambda: hag.gemm_ex(x, w, b, None, None, out, g.new_empty(m, N), 1))
        t_dg = timeit(lambda: hag.gemm_ex(dz, wt, None, None, z, None, g, 3))
        t_tn = timeit(lambda: hag.gemm_tn(dz, x))
        row = {"M": m, "K": K, "N": N}
        for k, t in (("fwd_nt", t_nt), ("fwd_nt_train", t_tr), ("dgrad", t_dg), ("wgrad_tn", t_tn)):
            row[k] = {"us": round(t * 1e6, 1), "TF": round(fl / t / 1e12, 1), "frac": round(fl / t / 1e12 / PEAK, 3)}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
