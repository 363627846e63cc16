#!/usr/bin/env python3
"""The five-layer chain (15 777 rows) with its wide row blocks handed out in gangs of G (HD_MLP_CHAIN_GANG): the G tiles
that read one slice of a W image start together.  tools/probes/chain_gang.py [G,G,...] [launches-only | check]
Timing: min of several alternating rounds, results bit-compared.  With a second argument: that many launches per G and
nothing else (for a rocprofv3 --pmc FETCH_SIZE pass; the launches appear in the order of the list)."""
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
gangs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]
if len(sys.argv) > 2 and sys.argv[2] == "check":       # results only: every gang size against G = 1, bit for bit
    os.environ["HD_MLP_CHAIN_GANG"] = "1"
    ref = ops.mlp_chain(al, layers)
    for g in gangs:
        os.environ["HD_MLP_CHAIN_GANG"] = str(g)
        for _ in range(3):
            assert torch.equal(ops.mlp_chain(al, layers), ref), g
    torch.cuda.synchronize()
    assert not ops.mlp_chain_failed()
    print("gang sizes", gangs, "bit-identical to G = 1")
    sys.exit(0)
if len(sys.argv) > 2:
    for g in gangs:
        os.environ["HD_MLP_CHAIN_GANG"] = str(g)
        for _ in range(int(sys.argv[2])):
            ops.mlp_chain(al, layers)
        torch.cuda.synchronize()
    sys.exit(0)
os.environ["HD_MLP_CHAIN_GANG"] = "1"
ref = ops.mlp_chain(al, layers)
res = {g: [] for g in gangs}
for rnd in range(4):
    for g in gangs:
        os.environ["HD_MLP_CHAIN_GANG"] = str(g)
        got = ops.mlp_chain(al, layers)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), g
        res[g].append(timeit(lambda: ops.mlp_chain(al, layers), 40))
assert not ops.mlp_chain_failed()
for g in gangs:
    print(f"gang {g:2d}: {min(res[g]):6.1f} us (of {[round(v, 1) for v in res[g]]})", flush=True)
