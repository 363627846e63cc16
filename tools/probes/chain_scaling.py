#!/usr/bin/env python3
"""fixed overhead and slope of the one-launch chain against per-layer launches: rows sweep, one layer and five"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops
ops.MLP_CHAIN_MIN_ROWS = 1

def timeit(fn, iters=40):
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

torch.manual_seed(0)
for dims in ([700, 512], [513, 800, 900, 800, 700, 512]):
    layers = [(torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5, torch.randn(dims[i + 1], device="cuda"),
               0 if i == len(dims) - 2 else 1) for i in range(len(dims) - 1)]
    for M in (128, 1024, 4096, 8192, 15777, 31554, 65536):
        x = torch.randn(M, dims[0], device="cuda")
        al = ops.limb_pack(x)
        wls = [ops.weight_limbs(w, "bf16x3")[0] for w, _, _ in layers]
        outs = [None if i == len(layers) - 1 else ops.limb_empty(M, wls[i].rows, "cuda") for i in range(len(layers))]
        c = torch.empty(M, dims[-1], device="cuda")
        arr = (ops._lib.HdMlpChainLayer * len(layers))()
        for i, (w, b, epi) in enumerate(layers):
            last = i == len(layers) - 1
            arr[i] = ops._lib.HdMlpChainLayer(ops._p(wls[i].buf), ops._p(b), None if last else ops._p(outs[i].buf),
                                              ops._p(c) if last else None, wls[i].rows, wls[i].cols, dims[-1], epi)
        ws = torch.zeros(8192, dtype=torch.int32, device="cuda")
        st = ops._stream()
        def chain():
            ops._lib.call("hd_mlp_chain_limbs_f32", ops._p(al.buf), M, arr, len(layers), ops._p(ws), ws.numel() * 4, st)
        def per_layer():
            cur = al
            for i, (w, b, epi) in enumerate(layers):
                last = i == len(layers) - 1
                ops._lib.call("hd_gemm_limbs_f32", ops._p(cur.buf), ops._p(wls[i].buf), ops._p(b), None, ops._p(c) if last else None,
                              None if last else ops._p(outs[i].buf), M, wls[i].rows, wls[i].cols, dims[-1] if last else wls[i].rows, epi, st)
                cur = outs[i]
        t5, t1 = timeit(per_layer), timeit(chain)
        print(f"layers {len(layers)} rows {M:6d}: per-layer launches {t5:7.1f} us | one launch {t1:7.1f} us", flush=True)
