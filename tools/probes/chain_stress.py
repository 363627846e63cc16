#!/usr/bin/env python3
"""Repeat the five-layer chain and compare every result with the per-layer launches: counts mismatching runs and
names the row blocks / column ranges that differ.  tools/probes/chain_stress.py [runs] [rows]"""
import os
import sys

import torch

os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
M = int(sys.argv[2]) if len(sys.argv) > 2 else 15777
dims = [513, 800, 900, 800, 700, 512]
torch.manual_seed(0)
x = torch.randn(M, dims[0], device="cuda")
layers = [(torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5, torch.randn(dims[i + 1], device="cuda"),
           0 if i == 4 else 1) for i in range(5)]
al = ops.limb_pack(x)
xf, cur = None, al
for i, (w, b, epi) in enumerate(layers):
    wl, _ = ops.weight_limbs(w, "bf16x3")
    xf, cur = ops.gemm_limbs(cur, wl, b, epi=epi, want_f32=i == 4, want_limbs=i < 4)
ref = xf
for flags in os.environ.get("STRESS_FLAGS", "0").split(","):
    os.environ["HD_MLP_CHAIN_FLAGS"] = flags
    bad = 0
    for r in range(runs):
        got = ops.mlp_chain(al, layers)
        if r % 3 == 0:
            torch.cuda.synchronize()
        ne = got != ref
        if bool(ne.any()):
            bad += 1
            if bad <= 5:
                rows = ne.any(1).nonzero().flatten()
                cols = ne.any(0).nonzero().flatten()
                print(f"flags {flags} run {r}: {int(ne.sum())} elements differ, rows {int(rows.min())}..{int(rows.max())} "
                      f"({rows.numel()} rows, row blocks {sorted(set((rows // 128).tolist()))[:12]}), cols {int(cols.min())}..{int(cols.max())} "
                      f"({cols.numel()}), nan {int(torch.isnan(got).sum())}, max |diff| {float((got - ref)[ne].abs().max()):.3g}", flush=True)
    print(f"flags {flags}: {bad} of {runs} runs differ; failed flag {ops.mlp_chain_failed()}", flush=True)
