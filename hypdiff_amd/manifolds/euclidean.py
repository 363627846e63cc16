"""The flat manifold of the reference's Euclidean baselines (manifolds/euclidean.py:11-67): every map is the identity or
a plain vector-space operation, so there is nothing to accelerate -- the class exists so that
`getattr(manifolds, args.manifold)` resolves for manifold='Euclidean' and callers of the Manifold API (base.py:15-73)
get the reference's values.  Table-driven torch one-liners; the hyperbolic layers of hypdiff_amd.layers dispatch to HIP
kernels by `manifold.abi` and accept PoincareBall / Hyperboloid only (unsupported configurations)."""
import torch

from .base import Manifold

# name -> (positional names after self, expression over them); c / dim are accepted and ignored as in the reference
_FLAT = {
    "sqdist": ("p1 p2 c", lambda p1, p2, c: torch.sum(torch.square(p1 - p2), dim=-1)),
    "egrad2rgrad": ("p dp c", lambda p, dp, c: dp),
    "proj": ("p c", lambda p, c: p),
    "proj_tan": ("u p c", lambda u, p, c: u),
    "proj_tan0": ("u c", lambda u, c: u),
    "expmap": ("u p c", lambda u, p, c: p + u),
    "logmap": ("p1 p2 c", lambda p1, p2, c: p2 - p1),
    "expmap0": ("u c", lambda u, c: u),
    "logmap0": ("p c", lambda p, c: p),
    "mobius_matvec": ("m x c", lambda m, x, c: torch.matmul(x, m.transpose(-1, -2))),
    "ptransp": ("x y v c", lambda x, y, v, c: v),
    "ptransp0": ("x v c", lambda x, v, c: x + v),
}


class Euclidean(Manifold):
    name = "Euclidean"
    abi = None                      # no HIP kernel family: the maps are identities

    def mobius_add(self, x, y, c, dim=-1):
        return x + y

    def inner(self, p, c, u, v=None, keepdim=False):
        return torch.sum(u * (u if v is None else v), dim=-1, keepdim=keepdim)

    def init_weights(self, w, c, irange=1e-5):
        with torch.no_grad():
            w.uniform_(-irange, irange)
        return w

    def normalize(self, p):
        """rows of norm > 1 are scaled back to the unit sphere, in place (:15-18)"""
        flat = p.view(-1, p.size(-1))
        with torch.no_grad():
            n = flat.norm(dim=1, keepdim=True)
            flat.mul_(torch.where(n > 1.0, 1.0 / (n + 1e-7), torch.ones_like(n)))
        return p


def _bind(name, expr):
    def method(self, *args):
        return expr(*args)
    method.__name__ = name
    return method


for _name, (_sig, _expr) in _FLAT.items():
    setattr(Euclidean, _name, _bind(_name, _expr))
