#!/usr/bin/env python3
"""The mlp denoiser's five GEMMs at M = 15 777: one persistent launch (hd_mlp_chain_limbs_f32) against five
hd_gemm_limbs_f32 launches, same operands; bit-compare, then time both (alternating, HIP events).
tools/bench_mlp_chain.py [--rows N] [--reps R]"""
import argparse
import os
import sys

import torch

os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops  # noqa: E402
ops.GEMM_VIA_CHAIN = False   # per-layer launches = the per-layer kernel (hd_gemm_limbs_f32)


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=15777)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--flags", default="8,40", help="HD_MLP_CHAIN_FLAGS values to time")
    a = ap.parse_args()
    M, dims = a.rows, [513, 800, 900, 800, 700, 512]
    torch.manual_seed(0)
    x = torch.randn(M, dims[0], device="cuda")
    layers = []
    for i in range(5):
        k, n = dims[i], dims[i + 1]
        layers.append((torch.randn(n, k, device="cuda") / k ** 0.5, torch.randn(n, device="cuda"), 0 if i == 4 else 1))
    al = ops.limb_pack(x)
    flops = sum(2.0 * M * dims[i] * dims[i + 1] for i in range(5))

    def per_layer():
        xf, cur = None, al
        for i, (w, b, epi) in enumerate(layers):
            wl, _ = ops.weight_limbs(w, "bf16x3")
            xf, cur = ops.gemm_limbs(cur, wl, b, epi=epi, want_f32=i == 4, want_limbs=i < 4)
        return xf

    ref = per_layer()
    got = ops.mlp_chain(al, layers)
    torch.cuda.synchronize()
    print("bit-identical:", bool(torch.equal(ref, got)), "max |diff|", float((ref - got).abs().max()), "failed flag:",
          ops.mlp_chain_failed(), flush=True)
    # single layers through the chain kernel (its K loop, write-out and protocol without the cross-layer schedule)
    cur = al
    for i, (w, b, epi) in enumerate(layers):
        wl, _ = ops.weight_limbs(w, "bf16x3")
        t5 = timeit(lambda: ops.gemm_limbs(cur, wl, b, epi=epi, want_f32=i == 4, want_limbs=i < 4), 30)
        outs = []

        def one():
            img = None if i == 4 else ops.limb_empty(M, wl.rows, "cuda")
            c = torch.empty(M, wl.rows, device="cuda") if i == 4 else None
            arr = (ops._lib.HdMlpChainLayer * 1)(ops._lib.HdMlpChainLayer(ops._p(wl.buf), ops._p(b), ops._p(img.buf) if img else None,
                                                                         ops._p(c), wl.rows, wl.cols, wl.rows, epi))
            ws = one.ws
            ops._lib.call("hd_mlp_chain_limbs_f32", ops._p(cur.buf), M, arr, 1, ops._p(ws), ws.numel() * 4, ops._stream())
        one.ws = torch.zeros(4096, dtype=torch.int32, device="cuda")
        t1 = timeit(one, 30)
        fl = 2.0 * M * wl.rows * wl.cols
        print(f"layer {i + 1}: hd_gemm_limbs_f32 {t5:6.1f} us {fl / t5 / 1e6:4.0f} TF | chain kernel, this layer alone {t1:6.1f} us {fl / t1 / 1e6:4.0f} TF", flush=True)
        _, cur = ops.gemm_limbs(cur, wl, b, epi=epi, want_f32=False, want_limbs=True)
    for flags in a.flags.split(","):
        flags, _, pc = flags.partition(":")
        os.environ["HD_MLP_CHAIN_PER_CU"] = pc or "2"
        os.environ["HD_MLP_CHAIN_FLAGS"] = flags
        ops._mlp_chain_ws.clear()
        got = ops.mlp_chain(al, layers)
        torch.cuda.synchronize()
        t1 = timeit(lambda: ops.mlp_chain(al, layers))
        ws = list(ops._mlp_chain_ws.values())[0]
        print(f"flags {flags} per_cu {pc or 2}: one launch {t1:6.1f} us {flops / t1 / 1e6:5.0f} TF, identical {bool(torch.equal(ref, got))}, "
              f"stats (slow paths, dep not ready, queue dry) {ws[260:263].tolist()} over {55 + 1} launches", flush=True)
    os.environ["HD_MLP_CHAIN_FLAGS"] = "0"
    os.environ["HD_MLP_CHAIN_PER_CU"] = "2"
    for r in range(a.reps):
        t5 = timeit(per_layer)
        t1 = timeit(lambda: ops.mlp_chain(al, layers))
        print(f"rep {r}: five launches {t5:6.1f} us {flops / t5 / 1e6:5.0f} TF | one launch {t1:6.1f} us {flops / t1 / 1e6:5.0f} TF "
              f"({flops / t1 / 1e6 / (2500 / 6):.3f} of the limb-adjusted peak)", flush=True)


if __name__ == "__main__":
    main()
