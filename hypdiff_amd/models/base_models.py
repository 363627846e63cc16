"""Encoder wrapper + link-prediction decoder -- drop-in for the hot-path part of the
reference's models/base_models.py (`BaseModel.encode` :78-88, `LPModel.decode` :152-159).
Metrics (sklearn ROC/AP on the host, :169-198) stay outside the accelerated path."""
import weakref

import torch
import torch.nn as nn

from .. import manifolds, ops
from . import encoders


class FermiDiracDecoder(nn.Module):
    """layers/layers.py:77-87"""

    def __init__(self, r, t):
        super(FermiDiracDecoder, self).__init__()
        self.r = r
        self.t = t

    def forward(self, dist):
        return 1. / (torch.exp((dist - self.r) / self.t) + 1.0)


class BaseModel(nn.Module):
    def __init__(self, args):
        super(BaseModel, self).__init__()
        self.manifold_name = args.manifold
        if args.c is not None:
            self.c = torch.tensor([args.c])
            if not args.cuda == -1:
                self.c = self.c.to(args.device)
        else:
            self.c = nn.Parameter(torch.Tensor([1.]))
        self.manifold = getattr(manifolds, self.manifold_name)()
        if self.manifold.name == 'Hyperboloid':
            args.feat_dim = args.feat_dim + 1
        self.nnodes = args.n_nodes
        self.encoder = getattr(encoders, args.model)(self.c, args)

    def encode(self, x, adj):
        """:78-88 -> (x, t, h0, adj) with x = h0 = the embeddings"""
        if self.manifold.name == 'Hyperboloid':
            # the zero time column (:79-81).  Inference on a persistent feature matrix: the widened copy is kept per
            # feature tensor, so that caches keyed on the operand (padded copies, sparse feature rows) keep hitting
            hit = getattr(self, "_time_col_cache", None)
            if (not torch.is_grad_enabled() and hit is not None and hit[0]() is x
                    and hit[1] == (x._version, x.data_ptr())):
                x = hit[2]
                ops._ready(hit[3])
            else:
                src, x = x, torch.cat([torch.zeros_like(x)[:, 0:1], x], dim=1)
                if not torch.is_grad_enabled() and x.shape[1] >= ops.SPARSE_ROWS_MIN_WIDTH:
                    self._time_col_cache = (weakref.ref(src), (src._version, src.data_ptr()), x, ops._tag())
        h = self.encoder.encode(x, adj)
        return h, 1, h, adj


class LPModel(BaseModel):
    def __init__(self, args):
        super(LPModel, self).__init__(args)
        self.dc = FermiDiracDecoder(r=args.r, t=args.t)
        self.nb_false_edges = args.nb_false_edges
        self.nb_edges = args.nb_edges

    def decode(self, h, idx):
        """:152-159: gather both endpoints, sqdist and the Fermi-Dirac map in ONE kernel"""
        ops._dev(h)
        idx = idx.to(device=h.device, dtype=torch.int64).contiguous()
        if torch.is_grad_enabled() and h.requires_grad:
            from .. import autograd as hag          # training (lp_train.py:66-82)
            out = hag.EdgeDecodeFn.apply(h, idx, self.manifold.abi, ops.curv(self.c), float(self.dc.r),
                                         float(self.dc.t))
            return out.unsqueeze(-1) if self.manifold.name == 'Hyperboloid' else out

        h = h.contiguous()
        out = torch.empty(idx.shape[0], dtype=h.dtype, device=h.device)
        ops._lib.call("hd_%s_edge_decode" % self.manifold.abi + ops._sfx(h), ops._p(h), ops._p(idx), h.shape[0],
                      ops._p(out), idx.shape[0], h.shape[1], ops.curv(self.c), 1, float(self.dc.r), float(self.dc.t),
                      ops._stream())
        return out.unsqueeze(-1) if self.manifold.name == 'Hyperboloid' else out

    def lp_loss(self, h, edges, edges_false):
        """The loss of compute_metrics (:169-184): BCE of the positive pairs against 1 plus BCE
        of the sampled negative pairs against 0 (the ROC / AP metrics stay on the host with
        sklearn, outside the accelerated path).  The two BCE means are scalar glue over the
        (E,) probabilities; decode and its backward are the kernels."""
        pos = self.decode(h, edges)
        neg = self.decode(h, edges_false)
        loss = torch.nn.functional.binary_cross_entropy(pos, torch.ones_like(pos))
        return loss + torch.nn.functional.binary_cross_entropy(neg, torch.zeros_like(neg))
