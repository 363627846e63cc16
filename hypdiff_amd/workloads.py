"""Synthetic graph-set workloads shaped like the reference's datasets (host-side numpy).

The shipped pickles are never unpickled; graph sets are regenerated with the same
statistics: SynEgo1000 = 1197 ego graphs, 4-20 nodes,
sum N = 15 777, sum E = 16 608.  Adjacency preprocessing restates
utils/data_utils.py:103-133 (`normalize(A + I)` = D^-1 (A+I), COO, fp32 values) and
GraphDataset's identity features (diff.py:38-39), batched block-diagonally so the whole
set is one CSR.
"""
import numpy as np
import scipy.sparse as sp


def ego_graph_set(n_graphs=1197, total_nodes=15777, total_edges=16608, n_min=4, n_max=20, seed=1432):
    rs = np.random.RandomState(seed)
    sizes = np.clip(np.rint(rs.normal(total_nodes / n_graphs, 4.0, n_graphs)), n_min, n_max).astype(np.int64)
    while sizes.sum() != total_nodes:                       # nudge to the exact node count
        i = rs.randint(n_graphs)
        step = 1 if sizes.sum() < total_nodes else -1
        if n_min <= sizes[i] + step <= n_max:
            sizes[i] += step
    extra = max(0, total_edges - int((sizes - 1).sum()))    # chords beyond the ego star
    chord_p = extra / max(1, int((sizes - 1).sum()))
    mats = []
    for n in sizes:
        r, c = [], []
        for i in range(1, n):
            r += [0, i]; c += [i, 0]
            if rs.rand() < chord_p and n > 2:
                j = 1 + rs.randint(n - 1)
                if j != i:
                    r += [i, j]; c += [j, i]
        a = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n))
        a.data[:] = 1.0
        mats.append(a)
    return mats, sizes


def normalize_adj(adj):
    """utils/data_utils.py:109-122: D^-1 (A + I) in fp64, zero for empty rows."""
    mx = adj + sp.eye(adj.shape[0])
    rowsum = np.asarray(mx.sum(1), dtype=np.float64).reshape(-1)
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def batch_graph_set(mats, sizes, feat_dim):
    """One block-diagonal normalised COO (int64 indices, fp32 values, row-sorted) and the
    per-node identity features of width feat_dim (node i of a graph -> e_i)."""
    big = sp.block_diag([normalize_adj(a) for a in mats], format="csr").tocoo()
    idx = np.vstack((big.row, big.col)).astype(np.int64)
    val = big.data.astype(np.float32)
    n = int(sizes.sum())
    feats = np.zeros((n, feat_dim), dtype=np.float32)
    off = 0
    for s in sizes:
        feats[np.arange(off, off + s), np.arange(s)] = 1.0
        off += s
    return idx, val, (n, n), feats


def power_law_graph(n, m, seed=1432):
    """Config 5's graph: mean-field Barabasi-Albert.  Node i attaches m times to floor(i * u^2),
    u ~ U(0,1) -- P(j) ~ 1 / (2 sqrt(i j)), the BA attachment kernel -- vectorised so that 10^6
    nodes take a second on the host.  Symmetric 0/1 scipy CSR, no self loops."""
    rs = np.random.RandomState(seed)
    src = np.repeat(np.arange(m, n, dtype=np.int64), m)
    dst = np.floor(src * rs.random_sample(src.shape[0]) ** 2).astype(np.int64)
    a = sp.coo_matrix((np.ones(src.shape[0], dtype=np.float32), (src, dst)), shape=(n, n)).tocsr()
    a = a + a.T
    a.setdiag(0)
    a.eliminate_zeros()
    a.data[:] = 1.0
    return a


def add_hubs(adj, degrees, seed=7, cluster=None):
    """Inject star rows into a symmetric 0/1 adjacency: for every entry of `degrees` one node
    (spread over the index range) is joined to that many random others.  `cluster` = (first_row,
    rows, degree) additionally gives `rows` CONSECUTIVE nodes `degree` random neighbours each (a
    run of heavy rows that are individually below the long-row threshold)."""
    n = adj.shape[0]
    rs = np.random.RandomState(seed)
    r, c = [], []
    for i, deg in enumerate(degrees):
        hub = (i + 1) * n // (len(degrees) + 1)
        nb = rs.choice(n - 1, size=min(deg, n - 1), replace=False)
        nb[nb >= hub] += 1
        r.append(np.full(nb.shape[0], hub)); c.append(nb)
    if cluster is not None:
        first, rows, deg = cluster
        for row in range(first, first + rows):
            nb = rs.choice(n - 1, size=deg, replace=False)
            nb[nb >= row] += 1
            r.append(np.full(deg, row)); c.append(nb)
    r, c = np.concatenate(r), np.concatenate(c)
    extra = sp.coo_matrix((np.ones(r.shape[0], dtype=np.float32), (r, c)), shape=(n, n)).tocsr()
    a = (adj + extra + extra.T).tocsr()
    a.data[:] = 1.0
    return a


def community_graph_set(n_graphs=1000, total_nodes=7850, total_edges=4684, n_min=2, n_max=14, seed=1432):
    """SynCommunity1000-shaped set (1000 graphs, 2-14 nodes, sum N = 7 850,
    sum E = 4 684 -- fewer edges than nodes, so the graphs are sparse and partly disconnected): two
    communities per graph, edges drawn among intra-community pairs plus one bridge, nudged to the
    exact edge total."""
    rs = np.random.RandomState(seed)
    sizes = rs.randint(n_min, n_max + 1, size=n_graphs).astype(np.int64)
    while sizes.sum() != total_nodes:
        i = rs.randint(n_graphs)
        step = 1 if sizes.sum() < total_nodes else -1
        if n_min <= sizes[i] + step <= n_max:
            sizes[i] += step
    cand = []                                              # per graph: candidate pairs, bridge first
    for n in sizes:
        h = max(1, n // 2)
        pairs = [(0, h)] if n > h else []
        pairs += [(i, j) for i in range(n) for j in range(i + 1, n) if (i < h) == (j < h)]
        cand.append(pairs)
    want = np.minimum(np.rint(sizes * (total_edges / total_nodes)).astype(np.int64), [len(c) for c in cand])
    while want.sum() != total_edges:
        i = rs.randint(n_graphs)
        step = 1 if want.sum() < total_edges else -1
        if 0 <= want[i] + step <= len(cand[i]):
            want[i] += step
    mats = []
    for n, pairs, k in zip(sizes, cand, want):
        pick = [0] + list(1 + rs.choice(len(pairs) - 1, size=k - 1, replace=False)) if k > 0 and len(pairs) > 1 \
            else ([0] if k > 0 else [])
        r = [pairs[q][0] for q in pick] + [pairs[q][1] for q in pick]
        c = [pairs[q][1] for q in pick] + [pairs[q][0] for q in pick]
        mats.append(sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n)))
    return mats, sizes


def citation_graph(n=2708, m=2, seed=1432):
    """Cora-shaped stand-in (config 4: the Planetoid download is unavailable offline): sparse
    power-law graph with n nodes and ~2 m n stored entries (Cora: 2708 nodes, 10 556)."""
    return power_law_graph(n, m, seed)
