"""Oracle: HypLinear / HypAgg / HypAct / HGCN encode (torch CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatement of
layers/hyp_layers.py and models/encoders.py:96-119 of the reference, eval mode
(dropout off), use_att=0 path.
"""
import torch

from .manifolds import MANIFOLDS


def spmm_coo(indices, values, shape, x):
    """torch.spmm(adj, x) of layers/hyp_layers.py:150 for an (uncoalesced,
    row-sorted) COO adjacency, evaluated row by row in entry order with fp32
    (or x.dtype) accumulation."""
    out = torch.zeros(shape[0], x.shape[1], dtype=x.dtype)
    out.index_add_(0, indices[0], values.to(x.dtype).unsqueeze(1) * x[indices[1]])
    return out


def hyp_linear(manifold, x, weight, c, bias=None):
    """layers/hyp_layers.py:98-108."""
    M = MANIFOLDS[manifold]
    res = M["proj"](M["mobius_matvec"](weight, x, c), c)
    if bias is not None:
        b = M["proj_tan0"](bias.view(1, -1), c)
        hb = M["proj"](M["expmap0"](b, c), c)
        res = M["proj"](M["mobius_add"](res, hb, c), c)
    return res


def hyp_agg(manifold, x, adj, c):
    """layers/hyp_layers.py:133-152, use_att=False.  adj = (indices, values, shape)."""
    M = MANIFOLDS[manifold]
    xt = M["logmap0"](x, c)
    support = spmm_coo(adj[0], adj[1], adj[2], xt)
    return M["proj"](M["expmap0"](support, c), c)


def hyp_act(manifold, x, c_in, c_out, act=torch.relu):
    """layers/hyp_layers.py:170-173."""
    M = MANIFOLDS[manifold]
    xt = act(M["logmap0"](x, c_in))
    xt = M["proj_tan0"](xt, c_out)
    return M["proj"](M["expmap0"](xt, c_out), c_out)


def hgconv(manifold, x, adj, weight, c_in, c_out, bias=None, act=torch.relu):
    """layers/hyp_layers.py:68-74."""
    h = hyp_linear(manifold, x, weight, c_in, bias)
    h = hyp_agg(manifold, h, adj, c_in)
    return hyp_act(manifold, h, c_in, c_out, act)


def hnn_layer(manifold, x, weight, c, bias=None, act=torch.relu):
    """layers/hyp_layers.py:51-54."""
    return hyp_act(manifold, hyp_linear(manifold, x, weight, c, bias), c, c, act)


def hgcn_encode(manifold, x, adj, weights, curvatures, biases=None, act=torch.relu,
                prepend_time=True):
    """models/base_models.py:78-82 + models/encoders.py:115-119.

    `curvatures` has len(weights)+1 entries (encoders.py:100-101).  For the
    hyperboloid BaseModel.encode first prepends a zero time column."""
    M = MANIFOLDS[manifold]
    if manifold == "Hyperboloid" and prepend_time:
        x = torch.cat([torch.zeros_like(x[:, 0:1]), x], dim=1)
    c0 = curvatures[0]
    h = M["proj"](M["expmap0"](M["proj_tan0"](x, c0), c0), c0)
    for i, w in enumerate(weights):
        b = None if biases is None else biases[i]
        h = hgconv(manifold, h, adj, w, curvatures[i], curvatures[i + 1], b, act)
    return h


def fermi_dirac(sqdist, r, t):
    """layers/layers.py:85-87."""
    return 1.0 / (torch.exp((sqdist - r) / t) + 1.0)


def lp_decode(manifold, h, idx, c, r, t):
    """models/base_models.py:152-159 (hyperbolic manifolds)."""
    M = MANIFOLDS[manifold]
    return fermi_dirac(M["sqdist"](h[idx[:, 0]], h[idx[:, 1]], c), r, t)
