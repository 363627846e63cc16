#!/usr/bin/env python3
"""One limb GEMM (the HypLinear of the encode: 15 777 x 512 x 512) through hd_gemm_limbs_f32 and through the chain
kernel with one layer (128 x 128 tiles, work queue).  tools/probes/single_gemm.py [M] [K] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops  # noqa: E402
ops.GEMM_VIA_CHAIN = False   # ops.gemm_limbs = hd_gemm_limbs_f32 here
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_mlp_chain import timeit  # noqa: E402

M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (15777, 512, 512)))
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") / K ** 0.5
al = ops.limb_pack(x)
wl, _ = ops.weight_limbs(w, "bf16x3")
ref, _ = ops.gemm_limbs(al, wl, None, epi=0, want_f32=True, want_limbs=False)
got = ops._mlp_chain_launch(al, [{"w": wl, "bias": None, "epi": 0, "f32": True}])[0][1]
torch.cuda.synchronize()
print("identical", bool(torch.equal(ref, got)))
fl = 2.0 * M * N * K
for dly in (int(v) for v in os.environ.get("DELAYS", "0").split(",")):
  os.environ["HD_MLP_CHAIN_DELAY_NS"] = str(dly)      # start stagger of every second workgroup (bench_mlp_chain sets HD_GEMM_TUNE_RELOAD)
  print("delay", dly, "ns")
  for rep in range(3):
    t0 = timeit(lambda: ops.gemm_limbs(al, wl, None, epi=0, want_f32=True, want_limbs=False), 50)
    t1 = timeit(lambda: ops._mlp_chain_launch(al, [{"w": wl, "bias": None, "epi": 0, "f32": True}]), 50)
    print(f"gemm_limbs {t0:6.1f} us {fl / t0 / 1e6:4.0f} TF | chain kernel, one layer {t1:6.1f} us {fl / t1 / 1e6:4.0f} TF", flush=True)
