#!/usr/bin/env python3
"""Throughput of the hot path on the shapes of BASELINE.json's configs 1-4, one GPU
(config 2 is the bench.py headline; config 5 needs 8 GPUs and is covered per kernel by
tools/bench_kernels.py).  One JSON line per measurement; HIP-event timing, inputs resident."""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops, workloads  # noqa: E402
from hypdiff_amd.ddpm import Model  # noqa: E402
from hypdiff_amd.ddpm_utils import DDPMSampler  # noqa: E402
from hypdiff_amd.dist import DataParallelTrainer  # noqa: E402
from hypdiff_amd.models.base_models import LPModel  # noqa: E402
from hypdiff_amd.models.direction_diffusion import diffusion  # noqa: E402
from hypdiff_amd.models.encoders import HGCN  # noqa: E402

dev = "cuda"


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def out(cfg, what, units, sec, **kw):
    print(json.dumps(dict(config=cfg, what=what, ms=round(sec * 1e3, 4), per_s=round(units / sec, 1), **kw)), flush=True)


def graph_set(manifold, c, n_graphs, total_nodes, total_edges, n_min, n_max, feat, dim, T, tag):
    mats, sizes = workloads.ego_graph_set(n_graphs, total_nodes, total_edges, n_min, n_max, seed=1432)
    idx, val, shape, feats = workloads.batch_graph_set(mats, sizes, feat)
    n = shape[0]
    hb = manifold == "Hyperboloid"
    args = types.SimpleNamespace(manifold=manifold, c=c, cuda=0, device=dev, feat_dim=feat + (1 if hb else 0), dim=dim,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    enc = HGCN(torch.tensor([c], device=dev), args).to(dev).eval()
    diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T).eval()
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), shape,
                                  check_invariants=False)
    x = torch.from_numpy(feats).to(dev)
    if hb:
        x = torch.cat([torch.zeros_like(x[:, :1]), x], 1).contiguous()
    rs = np.random.RandomState(0)
    labels = rs.randint(0, 3, size=n)
    u0 = torch.from_numpy((rs.standard_normal((3, dim)) * 0.05).astype(np.float32)).to(dev)
    img = torch.from_numpy(rs.standard_normal((n, dim))).to(dev)
    emb = rs.standard_normal((n, dim)).astype(np.float32)
    emb = torch.from_numpy(emb * 0.5 / np.linalg.norm(emb, axis=1, keepdims=True)).to(dev)
    with torch.no_grad():
        direction = diff.directions(u0)
        ops.csr_of(adj)
        out(tag, "HGCN encode (nodes/s)", n, timeit(lambda: enc.encode(x, adj), 30), rows=n, dim=dim)
        out(tag, "p_sample step, mlp denoiser (node-steps/s)", n,
            timeit(lambda: diff.p_sample(img, None, 1, direction, labels, u0, T // 2, None, True), 30), rows=n, dim=dim)
    diff.train()
    tr = DataParallelTrainer(diff.mlp, lr=1e-3)
    out(tag, "denoiser train step fwd+bwd+Adam (node-steps/s)", n,
        timeit(lambda: tr.step(lambda: diff(emb, None, labels, u0, restrict=True, graphset=True)), 15), rows=n, dim=dim)
    try:
        if os.environ.get("HD_BENCH_NO_GRAPH"):
            raise RuntimeError("skipped (HD_BENCH_NO_GRAPH)")
        replay = tr.capture(lambda: diff(emb, None, labels, u0, restrict=True, graphset=True))
        out(tag, "denoiser train step fwd+bwd+Adam as ONE HIP graph (node-steps/s)", n, timeit(replay, 30), rows=n, dim=dim)
    except Exception as e:                                   # report, keep the other lines
        print(json.dumps(dict(config=tag, what="graphed train step", error=repr(e)[:300])), flush=True)


def config3():
    B, D, Hd, L, T = 200, 64, 512, 10, 1000
    model = Model(D, Hd, L, T, 1e-4, 0.02).to(dev)
    x = torch.randn(B, D, device=dev)
    s = DDPMSampler(1e-4, 0.02, T, model.eval(), dev, (D,))
    with torch.no_grad():
        nz = torch.randn_like(x)
        out("3: BA graph-set latent DDPM", "DDPMSampler step, Backbone 24 Linear (latent-steps/s)", B,
            timeit(lambda: s.one_step(x, 500, nz), 50), rows=B, hidden=Hd)
    model.train()
    tr = DataParallelTrainer(model, lr=1e-3)
    out("3: BA graph-set latent DDPM", "ddpm.Model train step fwd+bwd+Adam (latent-steps/s)", B,
        timeit(lambda: tr.step(lambda: model.loss_fn(x)), 20), rows=B, hidden=Hd)
    replay = tr.capture(lambda: model.loss_fn(x))
    out("3: BA graph-set latent DDPM", "ddpm.Model train step fwd+bwd+Adam as ONE HIP graph (latent-steps/s)", B,
        timeit(replay, 50), rows=B, hidden=Hd)
    for Bb in (4096, 32768):
        xb = torch.randn(Bb, D, device=dev)
        out("3: BA graph-set latent DDPM", "ddpm.Model train step fwd+bwd+Adam (latent-steps/s)", Bb,
            timeit(lambda: tr.step(lambda: model.loss_fn(xb)), 10), rows=Bb, hidden=Hd)


def config3_again():
    B, D, Hd, L, T = 200, 64, 512, 10, 1000
    model = Model(D, Hd, L, T, 1e-4, 0.02).to(dev).eval()
    x = torch.randn(B, D, device=dev)
    s = DDPMSampler(1e-4, 0.02, T, model, dev, (D,))
    with torch.no_grad():
        nz = torch.randn_like(x)
        out("3: BA graph-set latent DDPM", "DDPMSampler step again, at the end of this process (after the captured steps)", B,
            timeit(lambda: s.one_step(x, 500, nz), 50), rows=B, hidden=Hd)


def config4():
    n, dim, c = 2708, 512, 0.6
    import scipy.sparse as sp
    rs = np.random.RandomState(1432)
    src = np.repeat(np.arange(2, n), 2)
    dst = np.floor(src * rs.random_sample(src.shape[0]) ** 2).astype(np.int64)
    a = sp.coo_matrix((np.ones(src.shape[0]), (src, dst)), shape=(n, n)).tocsr()
    a = a + a.T
    a.data[:] = 1.0
    nadj = workloads.normalize_adj(a).tocoo()
    idx = np.vstack((nadj.row, nadj.col)).astype(np.int64)
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(nadj.data.astype(np.float32)).to(dev),
                                  (n, n), check_invariants=False)
    args = types.SimpleNamespace(manifold="PoincareBall", c=c, cuda=0, device=dev, feat_dim=n, n_nodes=n, model="HGCN",
                                 r=2.0, t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, num_layers=2, act="relu",
                                 dropout=0.0, bias=0, use_att=0, local_agg=0)
    lp = LPModel(args).to(dev)
    feats = torch.eye(n, device=dev)
    tri = np.vstack(sp.triu(a, 1).nonzero()).T.astype(np.int64)
    pos = torch.from_numpy(tri).to(dev)
    neg = torch.from_numpy(rs.randint(0, n, size=tri.shape).astype(np.int64)).to(dev)
    tag = "4: citation-shaped link prediction (N=2708, identity features, nnz=%d)" % idx.shape[1]
    lp.eval()
    with torch.no_grad():
        out(tag, "HGCN encode (nodes/s)", n, timeit(lambda: lp.encode(feats, adj), 30), rows=n, dim=dim)
    lp.train()
    opt = torch.optim.Adam(lp.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        h = lp.encode(feats, adj)[0]
        lp.lp_loss(h, pos, neg).backward()
        opt.step()
    out(tag, "lp_train epoch: encode + decode + BCE + backward + Adam (nodes/s)", n, timeit(step, 20), rows=n, dim=dim,
        edges=int(tri.shape[0]))
    try:
        tr = DataParallelTrainer(lp, lr=1e-4)
        replay = tr.capture(lambda: lp.lp_loss(lp.encode(feats, adj)[0], pos, neg))
        out(tag, "lp_train epoch as ONE HIP graph (nodes/s)", n, timeit(replay, 30), rows=n, dim=dim, edges=int(tri.shape[0]))
    except Exception as e:
        import traceback
        print(json.dumps(dict(config=tag, what="graphed lp_train epoch", error=traceback.format_exc()[-700:])), flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    # config 3 first: its one-launch (cooperative) sampler step measured 0.32 ms instead of 0.19 when the process's FIRST
    # cooperative launch came after the captured mlp steps of configs 1 / 2; once it has run early, it stays at 0.19
    # after them too (config3_again() records that).  Not reproduced in four isolation scripts; cause open.
    slow = "--order-slow" in sys.argv      # the order in which the observation was made: config 3 AFTER the captured mlp steps
    only3 = "--only-3" in sys.argv
    if not slow:
        config3()
    if not only3:
        graph_set("PoincareBall", 1.0, 1000, 7850, 4684, 2, 14, 14, 14, 200, "1: SynCommunity1000-shaped (sum N=7850), PoincareBall, D=14, T=200")
        graph_set("Hyperboloid", 1.0, 1197, 15777, 16608, 4, 20, 20, 512, 1000, "2: SynEgo1000-shaped (sum N=15777), Hyperboloid, D=512, T=1000")
    if slow:
        config3()
    if not only3:
        config4()
    config3_again()
