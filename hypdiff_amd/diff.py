"""Latent -> graph decoding -- drop-in for `generate`, `decode_adj` and `get_graph` of the reference's
diff.py:327-391.  The tensor part (logit, per-graph min-max normalisation, threshold, decode_adj) is one
HIP kernel over the whole graph set (hd_generate_adj_f32) instead of a Python loop with a host round trip
per graph; only the networkx objects are built on the host, as in the reference."""
import os
import pickle

import numpy as np
import torch

from . import ops


def generate_adj(h):
    """(G, 1, n, n) or (G, n, n) latents in (0, 1) -> (adj (G, n+1, n+1) fp32 0/1 on the device, any (G,)
    int32): diff.py:334-349 up to the networkx hand-off.  any[g] == 0: nothing above the threshold."""
    h = torch.as_tensor(h)
    if h.dim() == 4:
        h = h.squeeze(dim=1)
    if h.dim() != 3 or h.shape[1] != h.shape[2]:
        raise ValueError("generate: latents must be (G, 1, n, n) / (G, n, n), got %s" % (tuple(h.shape),))
    if not h.is_cuda:
        h = h.to(torch.device("cuda", torch.cuda.current_device()))
    ops._dev(h)
    h = h.to(torch.float32).contiguous()
    g, n = h.shape[0], h.shape[1]
    adj = torch.empty(g, n + 1, n + 1, dtype=torch.float32, device=h.device)
    some = torch.empty(g, dtype=torch.int32, device=h.device)
    ops._lib.call("hd_generate_adj_f32", ops._p(h), ops._p(adj), ops._p(some), g, n, ops._stream())
    return adj, some


def decode_adj(adj_output):
    """diff.py:355-378 for ONE (n, m) 0/1 encoding (host numpy, as in the reference: it is called per graph
    on `.cpu().numpy()` data).  Vectorised: adj[i, j] = adj_output[i, i - j] for max(0, i - m + 1) <= j <= i."""
    a = np.asarray(adj_output)
    n, m = a.shape
    i = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    k = i - j
    ok = (k >= 0) & (k < m)
    adj = np.where(ok, a[np.broadcast_to(i, (n, n)), np.clip(k, 0, m - 1)], 0.0)
    full = np.zeros((n + 1, n + 1))
    full[1:, :n] = np.tril(adj, 0)
    return full + full.T


def get_graph(adj):
    """diff.py:380-391: drop all-zero rows / columns, build the networkx graph"""
    import networkx as nx
    adj = np.asarray(adj)
    adj = adj[~np.all(adj == 0, axis=1)]
    adj = adj[:, ~np.all(adj == 0, axis=0)]
    build = getattr(nx, "from_numpy_array", None) or nx.from_numpy_matrix     # networkx >= 3 renamed it
    return build(np.asarray(adj))


def generate(args, h, max_num_nodes, graphs, save=True):
    """diff.py:327-352.  Returns the list of predicted graphs (the reference only pickles it)."""
    adj, some = generate_adj(h)
    adj_h, some_h = adj.cpu().numpy(), some.cpu().numpy()
    G_pred_list = [get_graph(adj_h[i]) if some_h[i] else graphs[i] for i in range(adj_h.shape[0])]
    if save:
        save_dir = os.path.join(os.environ['LOG_DIR'], args.dataset)
        with open(os.path.join(save_dir, 'diff_embeddings.npy'), "wb") as f:
            pickle.dump(G_pred_list, f)
    return G_pred_list
