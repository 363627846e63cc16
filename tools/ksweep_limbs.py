#!/usr/bin/env python3
"""K sweep of the bf16x3 limb GEMM at M = 15 777: separates the per-slab cost from the fixed cost per tile round."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

fmt = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
M = int(os.environ.get("KSWEEP_M", 15777))
for N in [int(v) for v in os.environ.get("KSWEEP_N", "768,800,512").split(",")]:
    for K in [int(v) for v in os.environ.get("KSWEEP_K", "32,64,128,256,512,1024,2048").split(",")]:
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        al, wl = ops.limb_pack(a, fmt), ops.limb_pack(w, fmt)
        t0 = timeit(lambda: ops.gemm_limbs(al, wl, None, epi=0, want_f32=True))
        t1 = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=True))
        t2 = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=False, want_limbs=True, w_norm=1.0, bias_max=3.0))
        print(f"{fmt} N={N} K={K:5d} slabs={K//32:3d}: f32 plain {t0:7.1f} us | f32 bias+gelu {t1:7.1f} | limb bias+gelu {t2:7.1f}", flush=True)
