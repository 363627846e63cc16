"""Adjacency / feature preprocessing on the device -- drop-in for `process`, `normalize` and
`sparse_mx_to_torch_sparse_tensor` of the reference's utils/data_utils.py:103-133.

The reference does this with scipy on the host and uploads the result; here adj_train goes up once
as its CSR index arrays and `normalize(adj + I)` -> COO (int64 indices, fp32 values, the entry order
of scipy's .tocoo()) is built by HIP kernels (csrc/csr.hip), so a 10^6-node graph never takes the
host round trip.  `mask_edges` keeps numpy's seeded shuffles on the host (they ARE the split); its
non-edge list is O(N^2) by construction, so `mask_edges_sampled` is the form a 10^6-node graph can use.
Dataset loading is not restated here."""
import numpy as np
import scipy.sparse as sp
import torch

from .. import _lib, ops


def _csr_arrays(adj, device):
    """scipy sparse matrix (any format) or (rowptr, col[, val]) tensors -> int64 device CSR arrays."""
    if isinstance(adj, (tuple, list)):
        rowptr, col = adj[0], adj[1]
        val = adj[2] if len(adj) > 2 else None
        n = rowptr.numel() - 1
        return (rowptr.to(device=device, dtype=torch.int64).contiguous(), col.to(device=device, dtype=torch.int64).contiguous(),
                None if val is None else val.to(device=device, dtype=torch.float64).contiguous(), n)
    csr = adj.tocsr()
    csr.sum_duplicates()
    csr.sort_indices()
    n = csr.shape[0]
    assert csr.shape[0] == csr.shape[1], "adjacency must be square"
    data = np.asarray(csr.data, dtype=np.float64)
    val = None if np.all(data == 1.0) else torch.from_numpy(data).to(device)
    return (torch.from_numpy(csr.indptr.astype(np.int64)).to(device), torch.from_numpy(csr.indices.astype(np.int64)).to(device),
            val, n)


def normalize_adj(adj, device="cuda"):
    """normalize(adj + sp.eye(n)) -> torch sparse COO on the device (:110-111,115-133)."""
    rowptr, col, val, n = _csr_arrays(adj, device)
    counts = torch.empty(n, dtype=torch.int64, device=device)
    flags = torch.zeros(1, dtype=torch.int32, device=device)
    _lib.call("hd_adj_count", ops._p(rowptr), ops._p(col), n, ops._p(counts), ops._p(flags), ops._stream())
    out_ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=out_ptr[1:])
    f = int(flags.item())                      # the one word read back (also orders the nnz read below)
    if f & 2:
        raise IndexError("adjacency CSR has an index out of range")
    if f & 1:
        raise ValueError("adjacency CSR rows must have strictly ascending columns (scipy's canonical form)")
    nnz = int(out_ptr[-1].item())
    idx = torch.empty(2, nnz, dtype=torch.int64, device=device)
    vals = torch.empty(nnz, dtype=torch.float32, device=device)
    _lib.call("hd_adj_fill", ops._p(rowptr), ops._p(col), ops._p(val), n, ops._p(out_ptr), 1, ops._p(idx[0]),
              ops._p(idx[1]), ops._p(vals), ops._stream())
    return torch.sparse_coo_tensor(idx, vals, (n, n), check_invariants=False)


def normalize_features(features, device="cuda"):
    """normalize(features) (:106-108,115-122): rows divided by their fp64 sums, fp32 out."""
    if hasattr(features, "todense"):
        features = np.asarray(features.todense())
    x = torch.as_tensor(features).to(device)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float64)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=device)
    _lib.call("hd_feat_normalize", ops._p(x), 1 if x.dtype == torch.float64 else 0, x.shape[0], x.shape[1], ops._p(out),
              ops._stream())
    return out


def sparse_mx_to_torch_sparse_tensor(sparse_mx, device="cuda"):
    """:125-133 (upload only: int64 indices, fp32 values, .tocoo() order)."""
    coo = sparse_mx.tocoo()
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64)).to(device)
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float32)).to(device), coo.shape,
                                   check_invariants=False)


def process(adj, features, normalize_adj_, normalize_feats, device="cuda"):
    """:103-112 -> (adjacency as torch sparse COO, features as fp32 tensor), both on the device."""
    if normalize_feats:
        features = normalize_features(features, device)
    else:
        if hasattr(features, "todense"):
            features = np.asarray(features.todense())
        features = torch.as_tensor(features).to(device=device, dtype=torch.float32)
    if normalize_adj_:
        adj_t = normalize_adj(adj, device)
    else:
        adj_t = sparse_mx_to_torch_sparse_tensor(adj, device)
    return adj_t, features


def _split_positives(adj, val_prop, test_prop, seed):
    """The positive half of mask_edges (:60-63,70-73,76-77): numpy's legacy generator seeded, the upper-
    triangle edges shuffled FIRST in its stream -- so this part is identical to the reference whatever is
    done about the negatives afterwards."""
    np.random.seed(seed)
    x, y = sp.triu(adj).nonzero()
    pos = np.stack([x, y], 1)
    np.random.shuffle(pos)
    n_val, n_test = int(len(pos) * val_prop), int(len(pos) * test_prop)
    val, test, train = pos[:n_val], pos[n_val:n_val + n_test], pos[n_val + n_test:]
    a = sp.csr_matrix((np.ones(train.shape[0]), (train[:, 0], train[:, 1])), shape=adj.shape)
    return a + a.T, train, val, test, n_val, n_test


def mask_edges(adj, val_prop, test_prop, seed):
    """utils/data_utils.py:59-79, output for output (adj_train, train_edges, train_edges_false, val_edges,
    val_edges_false, test_edges, test_edges_false as int64 tensors).  The reference lists every upper-
    triangle NON-edge (diagonal included) and shuffles that list: O(N^2) memory and random draws by
    definition, so this form is for graphs where the reference itself can run (a few 10^4 nodes).  The
    non-edge list is enumerated row by row from the CSR instead of from a dense 1 - A."""
    adj = sp.csr_matrix(adj)
    adj_train, train, val, test, n_val, n_test = _split_positives(adj, val_prop, test_prop, seed)
    n = adj.shape[0]
    if n > 46000:
        raise MemoryError("mask_edges: the reference's non-edge list has ~n^2/2 = %.1e rows at n = %d; use "
                          "mask_edges_sampled" % (n * n / 2.0, n))
    csr = adj.tocsr()
    csr.sort_indices()
    rows = []
    for i in range(n):                      # row i of triu(1 - A): columns i..n-1 that hold no edge
        keep = np.ones(n - i, dtype=bool)
        cols = csr.indices[csr.indptr[i]:csr.indptr[i + 1]]
        vals = csr.data[csr.indptr[i]:csr.indptr[i + 1]]
        cols = cols[(cols >= i) & (vals == 1.0)]        # 1 - a is zero only where a == 1
        keep[cols - i] = False
        c = np.flatnonzero(keep) + i
        rows.append(np.stack([np.full(c.shape[0], i, dtype=np.int64), c.astype(np.int64)], 1))
    neg = np.concatenate(rows, 0) if rows else np.zeros((0, 2), dtype=np.int64)
    np.random.shuffle(neg)
    val_f, test_f = neg[:n_val], neg[n_val:n_val + n_test]
    train_f = np.concatenate([neg, val, test], axis=0)
    L = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64).reshape(-1, 2))
    return adj_train, L(train), L(train_f), L(val), L(val_f), L(test), L(test_f)


def mask_edges_sampled(adj, val_prop, test_prop, seed, n_train_false=None):
    """mask_edges for graphs beyond the reference's reach (config 5: 10^6 nodes).  adj_train and the
    train / val / test POSITIVE edges are the reference's, bit for bit (same generator, same first
    shuffle).  The negatives cannot be: the reference shuffles the full O(N^2) non-edge list.  Here
    n_val + n_test + n_train_false (default: the number of training positives) distinct upper-triangle
    non-edges (diagonal allowed, as in the reference) are drawn uniformly by rejection against the sorted
    edge keys -- O(samples log E) time and memory -- which is all a training loop consumes
    (models/base_models.py:172 draws nb_edges of them per step).  train_edges_false = the sampled training
    negatives followed by the validation and test positives, like the reference's list."""
    adj = sp.csr_matrix(adj)
    adj_train, train, val, test, n_val, n_test = _split_positives(adj, val_prop, test_prop, seed)
    n = adj.shape[0]
    want = n_val + n_test + (len(train) if n_train_false is None else int(n_train_false))
    coo = sp.triu(adj).tocoo()
    edge_keys = np.unique(coo.row.astype(np.int64) * n + coo.col.astype(np.int64))
    total_nonedges = n * (n + 1) // 2 - edge_keys.shape[0]
    want = min(want, total_nonedges)
    got = np.zeros(0, dtype=np.int64)
    while got.shape[0] < want:
        m = int((want - got.shape[0]) * 1.2) + 16
        a = np.random.randint(0, n, size=m).astype(np.int64)
        b = np.random.randint(0, n, size=m).astype(np.int64)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        keys = lo * n + hi
        pos = np.searchsorted(edge_keys, keys)
        is_edge = (pos < edge_keys.shape[0]) & (edge_keys[np.minimum(pos, edge_keys.shape[0] - 1)] == keys)
        keys = keys[~is_edge]
        merged = np.concatenate([got, keys])
        _, first = np.unique(merged, return_index=True)          # distinct, in draw order
        got = merged[np.sort(first)]
    got = got[:want]
    neg = np.stack([got // n, got % n], 1)
    val_f, test_f, tr_f = neg[:n_val], neg[n_val:n_val + n_test], neg[n_val + n_test:]
    train_f = np.concatenate([tr_f, val, test], axis=0)
    L = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int64).reshape(-1, 2))
    return adj_train, L(train), L(train_f), L(val), L(val_f), L(test), L(test_f)
