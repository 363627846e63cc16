#!/usr/bin/env python3
"""wall-clock stamps inside hd_mlp_chain_limbs_f32 (probe build knob HD_MLP_CHAIN_DEBUG_PTR): where a workgroup's time goes"""
import os, sys
import numpy as np
import torch
os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops

torch.manual_seed(0)
for dims, M in (([700, 512], 8192), ([700, 512], 15777), ([513, 800, 900, 800, 700, 512], 15777)):
    layers = [(torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5, torch.randn(dims[i + 1], device="cuda"),
               0 if i == len(dims) - 2 else 1) for i in range(len(dims) - 1)]
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
    dbg = torch.zeros(512 * 8 + 512 * 24 * 4 + 512 * 8 + 512 * 2 * 32, dtype=torch.int64, device="cuda")
    st = ops._stream()
    os.environ.pop("HD_MLP_CHAIN_DEBUG_PTR", None)
    for _ in range(5):
        ops._lib.call("hd_mlp_chain_limbs_f32", ops._p(al.buf), M, arr, len(layers), ops._p(ws), ws.numel() * 4, st)
    os.environ["HD_MLP_CHAIN_DEBUG_PTR"] = hex(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(60):                          # steady state: the stamps of the LAST launch of a back-to-back run
        if it == 10:
            e0.record()
        ops._lib.call("hd_mlp_chain_limbs_f32", ops._p(al.buf), M, arr, len(layers), ops._p(ws), ws.numel() * 4, st)
    e1.record()
    torch.cuda.synchronize()
    print(f"    (stamped launches, back to back: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us each)")
    full = dbg.cpu().numpy()
    d = full[:512 * 8].reshape(512, 8)
    tl = full[512 * 8:512 * 8 + 512 * 24 * 4].reshape(512, 24, 4)
    phases = full[512 * 8 + 512 * 24 * 4:512 * 8 + 512 * 24 * 4 + 512 * 8].reshape(512, 8)[:, :6].astype(np.float64) / 100.0 / 60.0   # us per launch (60 launches accumulate)
    t0 = d[:, 0].min()
    us = lambda a: (a - t0) / 100.0
    names = ["entry", "first item ready", "first tile K loop done", "first tile written", "loop left", "exit barrier", "exit atomic done"]
    print(f"--- layers {len(layers)} rows {M}: stamps in us after the first workgroup's entry (min / median / max over 512 workgroups)")
    for k, nm in enumerate(names):
        v = us(d[:, k].astype(np.float64))
        print(f"  {nm:<26s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")
    tiles = d[:, 7] >> 8
    print("  tiles per workgroup: min %d median %d max %d, total %d" % (tiles.min(), np.median(tiles), tiles.max(), tiles.sum()))
    # per layer: K loop time, write-out time, gap to the next tile's start (all workgroups, all tiles)
    rows = []
    for w in range(512):
        n = int(tiles[w])
        for k in range(min(n, 24)):
            t_start, t_k, t_w, tag = tl[w, k]
            gap = (tl[w, k + 1, 0] - t_w) / 100.0 if k + 1 < min(n, 24) else np.nan
            rows.append((int(tag >> 32), (t_k - t_start) / 100.0, (t_w - t_k) / 100.0, gap, (t_start - t0) / 100.0))
    rows = np.array(rows)
    for l in range(len(layers)):
        r = rows[rows[:, 0] == l]
        if len(r):
            print(f"  layer {l + 1}: {len(r):5d} tiles, K loop {np.median(r[:, 1]):6.2f} us median ({np.percentile(r[:, 1], 10):.2f}-{np.percentile(r[:, 1], 90):.2f}), "
                  f"write-out {np.median(r[:, 2]):5.2f} ({np.percentile(r[:, 2], 10):.2f}-{np.percentile(r[:, 2], 90):.2f}), gap to next {np.nanmedian(r[:, 3]):5.2f} "
                  f"({np.nanpercentile(r[:, 3], 10):.2f}-{np.nanpercentile(r[:, 3], 90):.2f}); starts {r[:, 4].min():.0f}-{r[:, 4].max():.0f} us")
    clk = full[512 * 8 + 512 * 24 * 4:512 * 8 + 512 * 24 * 4 + 512 * 8].reshape(512, 8)[:, 6:8].astype(np.float64)
    if clk[:, 1].sum() > 0:
        ghz = clk[:, 0] / np.maximum(clk[:, 1], 1) * 0.1
        print(f"  core clock inside the K loops (s_memtime against the 100 MHz s_memrealtime): median {np.median(ghz):.3f} GHz "
              f"({ghz.min():.3f}-{ghz.max():.3f})")
    slabs = full[512 * 8 + 512 * 24 * 4 + 512 * 8:].reshape(1024, 32).astype(np.float64)
    if slabs.sum() > 0 and len(layers) > 1:
        # slab-to-slab intervals of the 4th / 5th tile of every workgroup, grouped by the tile's slab count
        by = {}
        for row in slabs:
            n = int((row > 0).sum())
            if n >= 6:
                by.setdefault(n, []).append(np.diff(row[:n]) / 100.0)
        for n in sorted(by):
            d = np.median(np.array(by[n]), axis=0)
            print(f"  tiles of {n} slabs ({len(by[n])}): median us between slab tops: first {np.round(d[:4], 2).tolist()} ... middle "
                  f"{np.round(np.median(d[4:n - 7]), 2) if n > 12 else '-'} ... last {np.round(d[-6:], 2).tolist()}")
    if phases.sum() > 0:
        names = ["wait for the slab's LDS-DMA (vmcnt)", "barrier 1", "18 fragment reads + lgkmcnt(0)", "barrier 2", "staging issue + protocol", "48 MFMAs issued (+ write-out tail of the tile)"]
        tot = phases.sum(1)
        print("  wave 0 of a workgroup, per launch (median over workgroups), us and share of its K loops:")
        for k, nm in enumerate(names):
            print(f"    {nm:<50s} {np.median(phases[:, k]):8.1f}  {np.median(phases[:, k] / tot) * 100:5.1f} %")
