"""Hyperbolic graph encoders -- drop-in for HNN / HGCN of the reference's
models/encoders.py:48-71,91-119 (the Euclidean baselines are out of scope)."""
import torch
import torch.nn as nn

from .. import manifolds, ops
from ..layers import hyp_layers


class Encoder(nn.Module):
    """models/encoders.py:13-28"""

    def __init__(self, c):
        super(Encoder, self).__init__()
        self.c = c

    def encode(self, x, adj):
        if self.encode_graph:
            output, _ = self.layers.forward((x, adj))
        else:
            output = self.layers.forward(x)
        return output


def _lift(manifold, x, c):
    """proj(expmap0(proj_tan0(x))) in one pass (encoders.py:70,116-118).  Inference on fp32
    features whose width is not a multiple of 4: the (persistent) feature matrix is zero-padded to
    16-byte rows once (cached); zero columns stay zero through the lift and the first layer's
    weight is padded to match (ops.match_padded), so nothing is padded per call."""
    if x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 and not torch.is_grad_enabled():
        x = ops.pad_k_cached(x, (x.shape[1] + 3) // 4 * 4)
    return ops.map1("hd_%s_expmap0_proj" % manifold.abi, x, c)


class HNN(Encoder):
    def __init__(self, c, args):
        super(HNN, self).__init__(c)
        self.manifold = getattr(manifolds, args.manifold)()
        assert args.num_layers > 1
        dims, acts, _ = hyp_layers.get_dim_act_curv(args)
        layers = []
        for i in range(len(dims) - 1):
            layers.append(hyp_layers.HNNLayer(self.manifold, dims[i], dims[i + 1], self.c, args.dropout, acts[i],
                                              args.bias))
        self.layers = nn.Sequential(*layers)
        self.encode_graph = False

    def encode(self, x, adj):
        return super(HNN, self).encode(_lift(self.manifold, x, self.c), adj)


class HGCN(Encoder):
    def __init__(self, c, args):
        super(HGCN, self).__init__(c)
        self.manifold = getattr(manifolds, args.manifold)()
        assert args.num_layers > 1
        dims, acts, self.curvatures = hyp_layers.get_dim_act_curv(args)
        self.curvatures.append(self.c)
        layers = []
        for i in range(len(dims) - 1):
            c_in, c_out = self.curvatures[i], self.curvatures[i + 1]
            layers.append(hyp_layers.HyperbolicGraphConvolution(
                self.manifold, dims[i], dims[i + 1], c_in, c_out, args.dropout, acts[i], args.bias, args.use_att,
                args.local_agg))
        self.layers = nn.Sequential(*layers)
        self.encode_graph = True

    def encode(self, x, adj):
        c0 = self.curvatures[0]
        if not torch.is_grad_enabled() and not self.layers[0].linear.use_bias:
            # identity / one-hot features (utils/data_utils.py:171): the first HypLinear gathers rows of W^T
            sr = ops.sparse_rows_cached(x, self.manifold, c0, lambda t: _lift(self.manifold, t, c0))
            if sr is not None:
                return super(HGCN, self).encode(sr, adj)
        return super(HGCN, self).encode(_lift(self.manifold, x, c0), adj)
