"""Oracle: graph-VAE hyperbolic tail, edge masking, latent -> adjacency decoding, cal_u0.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatements (torch / numpy) of
models/hyp_layers.py:45-61,172-206,280-319, utils/data_utils.py:59-79, diff.py:327-380 and
models/direction_diffusion.py:130-139 of the reference, pinned by tests/golden/extras.npz
(captured from the reference itself by oracle/capture_golden.py --only extras).
"""
import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from .manifolds import pb_expmap, pb_expmap0, pb_logmap0, pb_ptransp0


# ---- graph-VAE tail: models/hyp_layers.py (PoincareBall, c = 1 hard-coded :180,:291,:309) ----
def vae_hyp_linear(x, weight, bias, c=1.0):
    """HypLinear.forward :195-206: the EUCLIDEAN product x W^T is taken as a tangent vector at the
    origin and mapped onto the ball (no proj); the bias is transported from the origin to that point.
    The last line of the reference is `manifold.expmap(x, bias)` with PoincareBall.expmap(u, p, c)
    (manifolds/poincare.py:55): the POINT x lands in the tangent slot u and the transported bias in
    the base-point slot p.  That argument order is part of the observable behaviour and is kept."""
    h = pb_expmap0(F.linear(x, weight), c)
    b = pb_ptransp0(h, bias.view(1, -1), c)
    return pb_expmap(h, b, c)


def vae_hyp_norm(h, ln_w, ln_b, c=1.0, eps=1e-5):
    """HypNorm.forward :313-319: LayerNorm in the tangent space at the origin."""
    t = pb_logmap0(h, c)
    t = F.layer_norm(t, (t.shape[-1],), ln_w, ln_b, eps)
    return pb_expmap0(t, c)


def vae_hyp_act(x, act=torch.relu, c=1.0):
    """HypAct.forward :292-296 (no proj after expmap0, unlike layers/hyp_layers.py:170-173)."""
    return pb_expmap0(act(pb_logmap0(x, c)), c)


def vae_hnn_layer(x, weight, bias, ln_w, ln_b, use_norm=True, act=torch.relu):
    """HNNLayer.forward :55-60."""
    h = vae_hyp_linear(x, weight, bias)
    if use_norm:
        h = vae_hyp_norm(h, ln_w, ln_b)
    return vae_hyp_act(h, act)


# ---- utils/data_utils.py:59-79 ----------------------------------------------------------------
def mask_edges(adj, val_prop, test_prop, seed):
    """The reference's split, stream for stream: seed numpy's legacy generator, shuffle the upper-
    triangle positives, then shuffle ALL upper-triangle non-edges (diagonal included: 1 - A has ones
    there), cut validation / test sets off both lists; train_edges_false = every non-edge followed by
    the validation and test positives.  Dense in N by construction (the non-edge list is O(N^2))."""
    np.random.seed(seed)
    x, y = sp.triu(adj).nonzero()
    pos = np.stack([x, y], 1)
    np.random.shuffle(pos)
    x, y = sp.triu(sp.csr_matrix(1.0 - adj.toarray())).nonzero()
    neg = np.stack([x, y], 1)
    np.random.shuffle(neg)
    n_val, n_test = int(len(pos) * val_prop), int(len(pos) * test_prop)
    val, test, train = pos[:n_val], pos[n_val:n_val + n_test], pos[n_val + n_test:]
    val_f, test_f = neg[:n_val], neg[n_val:n_val + n_test]
    train_f = np.concatenate([neg, val, test], axis=0)
    a = sp.csr_matrix((np.ones(len(train)), (train[:, 0], train[:, 1])), shape=adj.shape)
    return a + a.T, train, train_f, val, val_f, test, test_f


# ---- diff.py:327-380 ----------------------------------------------------------------------------
def decode_adj(adj_output):
    """decode_adj :360-378: row i of the (n, m) GraphRNN-style encoding lists, newest first, the links
    of node i+1 to its m predecessors; returns the symmetric (n+1, n+1) adjacency."""
    n, m = adj_output.shape
    adj = np.zeros((n, n))
    for i in range(n):
        lo = max(0, i - m + 1)
        k = i + 1 - lo                                   # predecessors that exist
        adj[i, lo:i + 1] = adj_output[i, :k][::-1]
    full = np.zeros((n + 1, n + 1))
    full[1:, :n] = np.tril(adj, 0)
    return full + full.T


def generate_adj(h):
    """generate :327-352 up to the networkx hand-off: logit of the latent, per-graph min-max
    normalisation, threshold at 0.5, decode_adj.  Returns a list with one (n+1, n+1) adjacency per
    graph, or None where nothing exceeds the threshold (the reference then keeps the input graph)."""
    e = torch.as_tensor(h).squeeze(dim=1)
    e = -torch.log((1 / (e + 1e-8)) - 1)
    out = []
    for g in e:
        y = (g - g.min()) / (g.max() - g.min())
        yr = torch.gt(y, torch.ones_like(y) * 0.5).float()
        out.append(decode_adj(yr.numpy().astype(np.float64)) if bool(torch.any(yr)) else None)
    return out


# ---- models/direction_diffusion.py:130-139 -------------------------------------------------------
def cal_u0(data):
    """Per cluster: the origin-tangent mean of its points, mapped back onto the ball (c = 1)."""
    return [pb_expmap0(torch.mean(pb_logmap0(x, 1.0), dim=0), 1.0) for x in data]
