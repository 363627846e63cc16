#!/usr/bin/env python3
"""Per-layer time of the mlp denoiser chain's GEMMs (M = 15 777, limb-image operands) for each limb format and
tile configuration: tools/bench_chain.py [--fmt bf16x3|f16x2] [--tunes N,W,X,...]"""
import argparse
import os
import sys

import torch

os.environ["HD_GEMM_TUNE_RELOAD"] = "1"      # the library re-reads the tile configuration on every launch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops  # noqa: E402


def timeit(fn, iters=30):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--fmt", default="bf16x3")
    ap.add_argument("--tunes", default="N,W,X")
    ap.add_argument("--rows", type=int, default=15777)
    ap.add_argument("--phase", default="0", help="HD_GEMM_LIMBS_PHASE values, comma separated")
    ap.add_argument("--delay", default="500", help="HD_GEMM_LIMBS_DELAY_NS values (ns per K slab), comma separated")
    a = ap.parse_args()
    M = a.rows
    dims = [513, 800, 900, 800, 700, 512]
    torch.manual_seed(0)
    env = "HD_GEMM_LIMBS_TUNE" if a.fmt == "bf16x3" else "HD_GEMM_F16X2_TUNE"
    for tune, phase, delay in [(t, p, d) for t in a.tunes.split(",") for p in a.phase.split(",")
                               for d in (a.delay.split(",") if p != "0" else ["0"])]:
        os.environ[env] = tune
        os.environ["HD_GEMM_LIMBS_PHASE"], os.environ["HD_GEMM_LIMBS_DELAY_NS"] = phase, delay
        tot, flops = 0.0, 0.0
        line = []
        for l in range(5):
            K, N = dims[l], dims[l + 1]
            x = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") / K ** 0.5
            b = torch.randn(N, device="cuda")
            al, wl = ops.limb_pack(x, a.fmt), ops.limb_pack(w, a.fmt)
            last = l == 4
            t = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=0 if last else 1, want_f32=last, want_limbs=not last,
                                              w_norm=1.0, bias_max=4.0))
            tot += t
            flops += 2.0 * M * N * K
            line.append(f"L{l + 1} {t:6.1f} us {2.0 * M * N * K / t / 1e6:5.0f} TF")
        print(f"{a.fmt} tune={tune} phase={phase} delay={delay}: " + " | ".join(line) + f" || chain {tot:6.1f} us {flops / tot / 1e6:5.0f} TF", flush=True)


if __name__ == "__main__":
    main()
