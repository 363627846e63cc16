#!/usr/bin/env python3
"""Where the host's time goes in a launch-bound path: cProfile of 2000 HGCN encodes at config 1's size (7850 x 14, ball).
tools/probes/host_profile.py"""
import cProfile
import os
import pstats
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hypdiff_amd import ops, workloads  # noqa: E402
from hypdiff_amd.models.encoders import HGCN  # noqa: E402

dev = "cuda"
mats, sizes = workloads.community_graph_set(seed=1) if hasattr(workloads, "community_graph_set") else workloads.ego_graph_set(seed=1)
idx, val, shape, feats = workloads.batch_graph_set(mats, sizes, 20)
n = shape[0]
x = torch.from_numpy(feats).to(dev)
adj = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), shape, check_invariants=False)
args = types.SimpleNamespace(manifold="PoincareBall", c=1.0, cuda=0, device=dev, feat_dim=x.shape[1], dim=14, num_layers=2,
                             act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
enc = HGCN(torch.tensor([1.0], device=dev), args).to(dev).eval()
with torch.no_grad():
    for _ in range(50):
        enc.encode(x, adj)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(2000):
        enc.encode(x, adj)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
