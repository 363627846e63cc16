#!/usr/bin/env python3
"""Time the five-layer chain against HD_MLP_CHAIN_TAIL (percent of the last layer's row blocks cut into narrow tiles),
alternating the settings over several rounds; every result is compared with the per-layer launches."""
import os
import sys

import torch

os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_mlp_chain import timeit  # noqa: E402

M, dims = 15777, [513, 800, 900, 800, 700, 512]
torch.manual_seed(0)
x = torch.randn(M, dims[0], device="cuda")
layers = [(torch.randn(dims[i + 1], dims[i], device="cuda") / dims[i] ** 0.5, torch.randn(dims[i + 1], device="cuda"),
           0 if i == 4 else 1) for i in range(5)]
al = ops.limb_pack(x)
xf, cur = None, al
for i, (w, b, epi) in enumerate(layers):
    wl, _ = ops.weight_limbs(w, "bf16x3")
    xf, cur = ops.gemm_limbs(cur, wl, b, epi=epi, want_f32=i == 4, want_limbs=i < 4)
tails = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "0,15,25,33,50,100").split(",")]
best = {t: [] for t in tails}
for rnd in range(4):
    for t in tails:
        os.environ["HD_MLP_CHAIN_TAIL"] = str(t)
        got = ops.mlp_chain(al, layers)
        torch.cuda.synchronize()
        assert torch.equal(got, xf), t
        best[t].append(timeit(lambda: ops.mlp_chain(al, layers), 40))
for t in tails:
    print(f"tail {t:3d} %: {min(best[t]):6.1f} us (of {[round(v, 1) for v in best[t]]})", flush=True)
print("failed flag", ops.mlp_chain_failed())
