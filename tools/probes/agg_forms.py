#!/usr/bin/env python3
"""Hyperboloid / ball fused aggregate on the power-law graph (10^6 x 256) in both forms of the row tail (1 = one row per
wavefront, 2 = four rows per wavefront), a few launches each, for rocprofv3 (kernel trace + --pmc): which resource the
four-row form loses on hub-heavy graphs.  tools/probes/agg_forms.py [rows] [dim]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda"
nadj = workloads.normalize_adj(workloads.power_law_graph(n, 5)).tocoo()
idx = np.vstack((nadj.row, nadj.col)).astype(np.int64)
csr = ops.csr_from_coo(torch.from_numpy(idx).to(dev), torch.from_numpy(nadj.data.astype(np.float32)).to(dev), (n, n))
xt = torch.randn(n, d, device=dev) * 0.1
print("rows", n, "dim", d, "nnz", idx.shape[1], "long rows", csr.n_long, flush=True)
real_form = ops.agg_form
for m, c in (("hb", 1.0), ("pb", 0.6)):
    for form in (1, 2):
        ops.agg_form = lambda *a, f=form: f
        csr.whole = (n, csr.n_long)
        for _ in range(3):
            ops.agg(m, csr, xt, c, c, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.agg(m, csr, xt, c, c, 1)
        e1.record()
        torch.cuda.synchronize()
        print(m, "form", form, round(e0.elapsed_time(e1) / 5 * 1e3, 1), "us", flush=True)
ops.agg_form = real_form
