#!/usr/bin/env python3
"""30 HGCN encodes at config 4's shape (N = 2708, identity features, PoincareBall c = 0.6, dim 512) for
rocprofv3 --kernel-trace --stats: tools/prof_encode_c4.py"""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import workloads
from hypdiff_amd.models.base_models import LPModel
import scipy.sparse as sp
dev = "cuda"
n, dim, c = 2708, 512, 0.6
rs = np.random.RandomState(1432)
src = np.repeat(np.arange(2, n), 2)
dst = np.floor(src * rs.random_sample(src.shape[0]) ** 2).astype(np.int64)
a = sp.coo_matrix((np.ones(src.shape[0]), (src, dst)), shape=(n, n)).tocsr()
a = a + a.T
a.data[:] = 1.0
nadj = workloads.normalize_adj(a).tocoo()
idx = np.vstack((nadj.row, nadj.col)).astype(np.int64)
adj = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(nadj.data.astype(np.float32)).to(dev), (n, n),
                              check_invariants=False)
args = types.SimpleNamespace(manifold="PoincareBall", c=c, cuda=0, device=dev, feat_dim=n, n_nodes=n, model="HGCN", r=2.0,
                             t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, num_layers=2, act="relu", dropout=0.0, bias=0,
                             use_att=0, local_agg=0)
lp = LPModel(args).to(dev).eval()
feats = torch.eye(n, device=dev)
with torch.no_grad():
    for _ in range(30):
        lp.encode(feats, adj)
torch.cuda.synchronize()
