"""Oracle: adjacency preprocessing and CSR indexing (numpy / scipy, integer-exact).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import scipy.sparse as sp


def row_normalize(mx):
    """utils/data_utils.py:115-122 -- D^-1 * mx with 1/0 -> 0, fp64."""
    rowsum = np.asarray(mx.sum(1), dtype=np.float64).reshape(-1)
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    return sp.diags(r_inv).dot(mx)


def process_adj(adj, normalize_adj=True):
    """utils/data_utils.py:109-111,125-133 -> (indices int64 (2,nnz), values fp32, shape).

    `adj` is a scipy sparse symmetric 0/1 matrix (adj_train)."""
    if normalize_adj:
        adj = row_normalize(adj + sp.eye(adj.shape[0]))
    coo = sp.coo_matrix(adj)
    indices = np.vstack((coo.row, coo.col)).astype(np.int64)
    values = coo.data.astype(np.float32)
    return indices, values, tuple(coo.shape)


def coo_to_csr(indices, values, n_rows):
    """Stable row bucketing of a COO list: rowptr (int32, n_rows+1), col (int32),
    val in the original within-row entry order.  For a row-sorted COO this is the
    identity permutation on the entries (what scipy's tocsr() yields as well)."""
    row = np.asarray(indices[0], dtype=np.int64)
    col = np.asarray(indices[1], dtype=np.int64)
    order = np.argsort(row, kind="stable")
    counts = np.bincount(row, minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr.astype(np.int32), col[order].astype(np.int32), np.asarray(values)[order]


def block_diag_batch(adjs):
    """Block-diagonal batching of a graph set: one CSR/COO over sum(N_g) rows.
    Per-graph results are unchanged because no entry crosses a block."""
    big = sp.block_diag([sp.coo_matrix(a) for a in adjs], format="coo")
    offsets = np.zeros(len(adjs) + 1, dtype=np.int64)
    np.cumsum([a.shape[0] for a in adjs], out=offsets[1:])
    return big, offsets
