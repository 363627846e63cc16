#!/usr/bin/env python3
"""HypAgg + HypAct on the power-law graph (10^6 x 256) and the ego set (15 777 x 512), both manifolds:
HD_SPMM_ROWS4=0/1 python tools/bench_agg.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops, workloads
from tools.bench_kernels import timeit

def main():
    dev = "cuda"
    cases = (("powerlaw_m5", (1_000_000, 256)), ("ego_set", (15777, 512)))
    if len(sys.argv) > 1:
        cases = (("powerlaw_m5", (500_000, 512)), ("ego_set", (15777, 256)), ("powerlaw_m5", (500_000, 128)))
    for tag, (n, d) in cases:
        if tag == "powerlaw_m5":
            nadj = workloads.normalize_adj(workloads.power_law_graph(n, 5)).tocoo()
            idx = np.vstack((nadj.row, nadj.col)).astype(np.int64); val = nadj.data.astype(np.float32)
        else:
            mats, sizes = workloads.ego_graph_set()
            idx, val, _, _ = workloads.batch_graph_set(mats, sizes, 20)
        csr = ops.csr_from_coo(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), (n, n))
        xt = torch.randn(n, d, device=dev) * 0.1
        for m, c in (("pb", 0.6), ("hb", 1.0)):
            print(tag, (n, d), m, round(timeit(lambda: ops.agg(m, csr, xt, c, c, 1), 20) * 1e6, 1), "us", flush=True)

if __name__ == "__main__":
    main()
