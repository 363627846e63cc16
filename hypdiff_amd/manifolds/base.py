"""Abstract manifold API: the method set and argument orders of the reference's
manifolds/base.py:15-73, which are the drop-in boundary of the geometry layer.  A concrete
manifold overrides what it accelerates; anything else raises NotImplementedError by name."""

# method -> positional signature after self (per-class argument ORDER is part of the contract:
# expmap(u, p, c) but logmap(p1, p2, c); mobius_matvec(m, x, c); sqdist(p1, p2, c))
API = {
    "sqdist": "p1, p2, c", "egrad2rgrad": "p, dp, c", "proj": "p, c", "proj_tan": "u, p, c",
    "proj_tan0": "u, c", "expmap": "u, p, c", "logmap": "p1, p2, c", "expmap0": "u, c", "logmap0": "p, c",
    "mobius_add": "x, y, c, dim=-1", "mobius_matvec": "m, x, c", "init_weights": "w, c, irange=1e-5",
    "inner": "p, c, u, v=None, keepdim=False", "ptransp": "x, y, u, c", "ptransp0": "x, u, c",
}


def _abstract(name, sig):
    def method(self, *args, **kwargs):
        raise NotImplementedError("%s.%s(%s) is not implemented" % (type(self).__name__, name, sig))
    method.__name__ = name
    method.__doc__ = "%s(%s) -- manifolds/base.py" % (name, sig)
    return method


class Manifold(object):
    def __init__(self):
        super().__init__()
        self.eps = 10e-8


for _name, _sig in API.items():
    setattr(Manifold, _name, _abstract(_name, _sig))


import torch  # noqa: E402


class ManifoldParameter(torch.nn.Parameter):
    """nn.Parameter that remembers the manifold and curvature it lives on (manifolds/base.py:76-88: what
    optimizers/radam.py keys its Riemannian update on).  Construction and repr follow the reference; a parameter of
    this type is an ordinary leaf tensor for everything else."""

    def __new__(cls, data, requires_grad, manifold, c):
        return super().__new__(cls, data, requires_grad)

    def __init__(self, data, requires_grad, manifold, c):
        self.manifold, self.c = manifold, c

    def __repr__(self):
        return f"{self.manifold.name} Parameter containing:\n{torch.Tensor.__repr__(self)}"
