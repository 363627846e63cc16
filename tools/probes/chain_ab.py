#!/usr/bin/env python3
"""Time the five-layer chain (15 777 rows) with the library HD_LIB_PATH names: min of several rounds; also with a start
stagger of the second workgroup of a CU (HD_MLP_CHAIN_DELAY_NS).  tools/probes/chain_ab.py [delay_ns,...]"""
import os
import sys

import torch

os.environ["HD_GEMM_TUNE_RELOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402
ops.GEMM_VIA_CHAIN = False   # per-layer launches = the per-layer kernel (hd_gemm_limbs_f32)
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
delays = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
res = {d: [] for d in delays}
for rnd in range(4):
    for d in delays:
        os.environ["HD_MLP_CHAIN_DELAY_NS"] = str(d)
        got = ops.mlp_chain(al, layers)
        torch.cuda.synchronize()
        assert os.environ.get("HD_AB_NOCHECK") or torch.equal(got, xf), d
        res[d].append(timeit(lambda: ops.mlp_chain(al, layers), 40))
for d in delays:
    print(f"{os.environ.get('HD_LIB_PATH', 'default')[-14:]} delay {d:5d} ns: {min(res[d]):6.1f} us (of {[round(v, 1) for v in res[d]]})", flush=True)
