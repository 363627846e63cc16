#!/usr/bin/env python3
"""End-to-end `lptask` of the reference (main.py:10-17 -> lp_train.train -> diff.hyperdiff) on a
synthetic citation-shaped graph, every stage on the MI355X through hypdiff_amd:

    HGCN encoder training (link prediction, lp_train.py:66-82: Adam, per-parameter grad clipping)
 -> hyperbolic k-means on the embeddings (hyperkmeans.py:42-44: k=3, 5 epochs)
 -> direction-diffusion training (diff.py:107-118, mlp or gcndec denoiser)
 -> p_sample_loop (diff.py:126-127) -> link-prediction score of the sampled embeddings

    python examples/lptask_synthetic.py [--nodes 2708] [--dim 64] [--epochs 10] [--diff-epochs 20] [--gcn]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import workloads  # noqa: E402
from hypdiff_amd.hyperbolic_learning.hyperbolic_kmeans.hkmeans import HyperbolicKMeans  # noqa: E402
from hypdiff_amd.models.base_models import LPModel  # noqa: E402
from hypdiff_amd.models.direction_diffusion import diffusion  # noqa: E402


def citation_graph(n, seed, n_comm=8, deg_in=8.0, deg_out=1.0):
    """Sparse planted-partition graph (a citation network's community structure): a node has on
    average deg_in neighbours inside its community and deg_out outside, so held-out edges are
    predictable from the observed ones."""
    rs = np.random.RandomState(seed)
    comm = rs.randint(0, n_comm, size=n)
    m_in, m_out = int(n * deg_in / 2), int(n * deg_out / 2)
    order = np.argsort(comm, kind="stable")
    start = np.searchsorted(comm[order], np.arange(n_comm))
    size = np.bincount(comm, minlength=n_comm)
    u = rs.randint(0, n, size=m_in)
    v = order[start[comm[u]] + (rs.random_sample(m_in) * size[comm[u]]).astype(np.int64)]
    src = np.concatenate([u, rs.randint(0, n, size=m_out)])
    dst = np.concatenate([v, rs.randint(0, n, size=m_out)])
    keep = src != dst
    a = sp.coo_matrix((np.ones(keep.sum()), (src[keep], dst[keep])), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    return a


def run(nodes=2708, dim=64, epochs=10, diff_epochs=20, timesteps=200, gcn=False, seed=1432, dropout=0.2, log=print):
    dev = "cuda:0"
    torch.manual_seed(seed)
    np.random.seed(seed)
    rs = np.random.RandomState(seed)
    a = citation_graph(nodes, seed)
    tri = np.vstack(sp.triu(a, 1).nonzero()).T.astype(np.int64)
    rs.shuffle(tri)
    n_test = max(1, tri.shape[0] // 10)
    test_pos, train_pos = tri[:n_test], tri[n_test:]
    a_train = sp.coo_matrix((np.ones(train_pos.shape[0]), (train_pos[:, 0], train_pos[:, 1])), shape=a.shape).tocsr()
    a_train = a_train + a_train.T
    nadj = workloads.normalize_adj(a_train).tocoo()                     # utils/data_utils.py:103-133
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((nadj.row, nadj.col)).astype(np.int64)).to(dev),
                                  torch.from_numpy(nadj.data.astype(np.float32)).to(dev), a.shape, check_invariants=False)
    neg_all = rs.randint(0, nodes, size=(4 * tri.shape[0], 2)).astype(np.int64)
    neg_all = neg_all[(neg_all[:, 0] != neg_all[:, 1]) & (np.asarray(a[neg_all[:, 0], neg_all[:, 1]]).ravel() == 0)]
    test_neg, train_neg = neg_all[:n_test], neg_all[n_test:]
    feats = torch.eye(nodes, device=dev)                                 # data_utils.py:171 identity features
    # c = 1: the diffusion stage hard-codes curvature 1 (models/direction_diffusion.py:119), so the
    # embeddings must live in the unit ball (with config.py's c = 0.6 rows of norm > 1 make its
    # logmap at the cluster centre NaN -- in the reference as well)
    args = types.SimpleNamespace(manifold="PoincareBall", c=1.0, cuda=0, device=dev, feat_dim=nodes, n_nodes=nodes,
                                 model="HGCN", r=2.0, t=1.0, nb_false_edges=train_neg.shape[0],
                                 nb_edges=train_pos.shape[0], dim=dim, num_layers=2, act="relu", dropout=dropout, bias=0,
                                 use_att=0, local_agg=0)
    model = LPModel(args).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    pos_t, test_pos_t, test_neg_t = (torch.from_numpy(x).to(dev) for x in (train_pos, test_pos, test_neg))
    res = {}

    def auc(h):
        with torch.no_grad():
            p = model.decode(h, test_pos_t).cpu().numpy().ravel()
            q = model.decode(h, test_neg_t).cpu().numpy().ravel()
        return float((p[:, None] > q[None, :]).mean() + 0.5 * (p[:, None] == q[None, :]).mean())

    t0 = time.perf_counter()
    losses = []
    for ep in range(epochs):                                            # lp_train.py:66-82
        model.train()
        opt.zero_grad()
        h = model.encode(feats, adj)[0]
        neg = torch.from_numpy(train_neg[rs.randint(0, train_neg.shape[0], train_pos.shape[0])]).to(dev)
        loss = model.lp_loss(h, pos_t, neg)
        loss.backward()
        for p in model.parameters():
            if p.grad is not None:
                torch.nn.utils.clip_grad_norm_(p, 10.0)
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    res["encoder_losses"], res["encoder_s"] = losses, time.perf_counter() - t0
    model.eval()
    with torch.no_grad():
        h0 = model.encode(feats, adj)[0]
    res["auc_encoder"] = auc(h0)
    log(f"encoder: loss {losses[0]:.4f} -> {losses[-1]:.4f}, test AUC {res['auc_encoder']:.3f}, {res['encoder_s']:.2f} s")

    km = HyperbolicKMeans(n_clusters=3)                                 # hyperkmeans.py:42-44
    centres, labels = km.fit(h0.double(), max_epochs=5)
    res["cluster_sizes"] = np.bincount(labels, minlength=3).tolist()
    log(f"k-means: cluster sizes {res['cluster_sizes']}")

    diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=timesteps)
    dopt = torch.optim.Adam(diff.parameters(), lr=1e-3)                 # diff.py:107
    u0 = torch.from_numpy(centres).to(device=dev, dtype=torch.float32)
    t0 = time.perf_counter()
    dl = []
    for ep in range(diff_epochs):                                       # diff.py:113-118
        dopt.zero_grad()
        loss = diff(h0, adj, labels, u0, restrict=True, graphset=not gcn)
        loss.backward()
        dopt.step()
        dl.append(float(loss.detach()))
    torch.cuda.synchronize()
    res["diffusion_losses"], res["diffusion_s"] = dl, time.perf_counter() - t0
    log(f"diffusion ({'gcndec' if gcn else 'mlp'}): loss {dl[0]:.4f} -> {dl[-1]:.4f}, {res['diffusion_s']:.2f} s")
    diff.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        sa = diff.p_sample_loop(h0.clone(), adj, labels, u0, graphset=not gcn)      # diff.py:126-127
    torch.cuda.synchronize()
    res["sample_s"] = time.perf_counter() - t0
    res["sample_finite"] = bool(torch.isfinite(sa).all())
    res["auc_sampled"] = auc(sa.float().contiguous())
    log(f"sampling: {timesteps} steps x {nodes} rows in {res['sample_s']:.2f} s "
        f"({nodes * timesteps / res['sample_s']:.3g} node-steps/s), finite={res['sample_finite']}, "
        f"test AUC of sampled embeddings {res['auc_sampled']:.3f}")
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=2708)
    ap.add_argument("--dim", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--diff-epochs", type=int, default=20)
    ap.add_argument("--timesteps", type=int, default=200)
    ap.add_argument("--gcn", action="store_true")
    a = ap.parse_args()
    run(a.nodes, a.dim, a.epochs, a.diff_epochs, a.timesteps, a.gcn)
