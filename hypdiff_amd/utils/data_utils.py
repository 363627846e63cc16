"""Adjacency / feature preprocessing on the device -- drop-in for `process`, `normalize` and
`sparse_mx_to_torch_sparse_tensor` of the reference's utils/data_utils.py:103-133.

The reference does this with scipy on the host and uploads the result; here adj_train goes up once
as its CSR index arrays and `normalize(adj + I)` -> COO (int64 indices, fp32 values, the entry order
of scipy's .tocoo()) is built by HIP kernels (csrc/csr.hip), so a 10^6-node graph never takes the
host round trip.  Dataset loading and `mask_edges` (numpy's seeded shuffles) stay host code in the
reference and are not restated here."""
import numpy as np
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
