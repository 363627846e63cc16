#!/usr/bin/env python3
"""Weight-gradient kernels at the denoiser's shapes (15 777 rows): tools/bench_tn.py [reps]
HIP events around `reps` launches each; prints us and algorithmic TFLOP/s per shape and GEMM mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import autograd as hag, ops  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    R = int(os.environ.get("TN_ROWS", 15777))
    shapes = [(800, 516), (900, 800), (800, 900), (700, 800), (512, 700)]
    g = torch.Generator(device="cuda").manual_seed(0)
    tot = {}
    for mode in ("bf16x3", "exact"):
        ops.GEMM_MODE = mode
        for P, Q in shapes:
            a = torch.randn(R, P, device="cuda", generator=g)
            b = torch.randn(R, Q, device="cuda", generator=g)
            out = torch.empty(P, Q, device="cuda")
            cs = torch.empty(P, device="cuda")
            for _ in range(5):
                hag.gemm_tn(a, b, out=out, colsum_out=cs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                hag.gemm_tn(a, b, out=out, colsum_out=cs)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            tot[mode] = tot.get(mode, 0.0) + us
            print(f"{mode:7s} {R}x{P}x{Q}: {us:7.1f} us  {2.0 * R * P * Q / us * 1e-6:6.1f} TFLOP/s", flush=True)
    for P, Q in shapes:                                   # the same products from limb images (hd_gemm_tn_limbs_f32)
        ai = ops.limb_pack(torch.randn(R, P, device="cuda", generator=g), "bf16x3")
        bi = ops.limb_pack(torch.randn(R, Q, device="cuda", generator=g), "bf16x3")
        Pp, Qp = (P + 3) // 4 * 4, (Q + 3) // 4 * 4
        out = torch.empty(Pp, Qp, device="cuda")
        cs = torch.empty(Pp, device="cuda")
        for _ in range(5):
            hag.gemm_tn_limbs(ai, bi, out=out, colsum_out=cs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hag.gemm_tn_limbs(ai, bi, out=out, colsum_out=cs)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tot["images"] = tot.get("images", 0.0) + us
        print(f"images  {R}x{P}x{Q}: {us:7.1f} us  {2.0 * R * P * Q / us * 1e-6:6.1f} TFLOP/s", flush=True)
    # the five image products as ONE launch (hd_gemm_tn_limbs_multi_f32)
    jobs = []
    for P, Q in shapes:
        ai = ops.limb_pack(torch.randn(R, P, device="cuda", generator=g), "bf16x3")
        bi = ops.limb_pack(torch.randn(R, Q, device="cuda", generator=g), "bf16x3")
        Pp, Qp = (P + 3) // 4 * 4, (Q + 3) // 4 * 4
        jobs.append((ai, bi, torch.empty(Pp, Qp, device="cuda"), torch.empty(Pp, device="cuda")))
    for _ in range(5):
        hag.gemm_tn_limbs_multi(jobs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hag.gemm_tn_limbs_multi(jobs)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = sum(2.0 * R * P * Q for P, Q in shapes)
    tot["images, one launch"] = us
    print(f"images, one launch: {us:7.1f} us  {fl / us * 1e-6:6.1f} TFLOP/s", flush=True)
    print({k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
