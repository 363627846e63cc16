#!/usr/bin/env python3
"""Per-shape sweep of the bf16x3 limb GEMM's tile configurations at the headline's six launches
(M = 15 777; limb bias+gelu write-out as in the denoiser): tools/cfg_sweep_limbs.py [configs]"""
import os, sys, torch
os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops


def timeit(fn, iters=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


cfgs = sys.argv[1] if len(sys.argv) > 1 else "TSNWX"
M = 15777
tot = {c: 0.0 for c in cfgs}
for N, K in [(512, 512), (800, 516), (900, 800), (800, 900), (700, 800), (512, 700)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    al, wl = ops.limb_pack(a, "bf16x3"), ops.limb_pack(w, "bf16x3")
    line = f"N={N} K={K}:"
    for c in cfgs:
        os.environ["HD_GEMM_LIMBS_TUNE"] = c
        last = (N, K) == (512, 700)
        t = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=0 if last else 1, want_f32=last, want_limbs=not last))
        tot[c] += t
        line += f"  {c} {t:6.1f}"
    print(line, flush=True)
print({k: round(v, 1) for k, v in tot.items()})
