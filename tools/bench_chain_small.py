#!/usr/bin/env python3
"""ddpm.Backbone at config 3's size (B latents of width 64, hidden 512, 10 layers): one cooperative launch
(hd_linear_chain_small_f32) against one GEMM launch per layer.  tools/bench_chain_small.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops  # noqa: E402
from hypdiff_amd.ddpm import Model  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    torch.manual_seed(0)
    model = Model(64, 512, 10, 1000, 1e-4, 0.02).cuda().eval()
    for B in [int(v) for v in sys.argv[1:]] or [200, 16, 512]:
        x = torch.randn(B, 64, device="cuda")
        idx = torch.randint(0, 1000, (B,), device="cuda")
        with torch.no_grad():
            ops._NO_CHAIN_SMALL = False
            t1 = timeit(lambda: model.backbone(x, idx))
            ops._NO_CHAIN_SMALL = True
            t2 = timeit(lambda: model.backbone(x, idx))
            ops._NO_CHAIN_SMALL = False
        print(f"B={B}: one launch {t1:7.1f} us | launch per layer {t2:7.1f} us", flush=True)


def depth_sweep():
    """time against the number of layers: slope = cost of one layer hand-over, intercept = launch"""
    torch.manual_seed(0)
    ws = [torch.randn(512, 512, device="cuda") / 512 ** 0.5 for _ in range(24)]
    bs = [torch.randn(512, device="cuda") for _ in range(24)]
    for B in (16, 200):
        x = torch.randn(B, 512, device="cuda")
        line = []
        for n in (1, 2, 4, 8, 16, 24):
            layers = [(ws[i], bs[i], 1 if i == 0 else 2) for i in range(n)]
            line.append(f"{n}: {timeit(lambda: ops.linear_chain_small(x, layers)):6.1f}")
        print(f"B={B} layers -> us  " + " | ".join(line), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "depth":
        depth_sweep()
    else:
        main()
