"""Graph-VAE hyperbolic tail on HIP kernels -- drop-in for the classes of the reference's
models/hyp_layers.py that actually run (HNNLayer :45-61 -> HypLinear :172-206, HypNorm :298-319,
HypAct :280-296; the file's other classes reference undefined names).  models/hyp_model.py:19,38 builds
ONE HNNLayer(GraphLatntDim, GraphLatntDim, PoincareBall(), PoincareBall()) on the pooled (B, D) latent.

Same constructors, attribute names and state_dict keys (`linear.linear.weight`, `linear.bias`,
`norm.norm.weight`, `norm.norm.bias`).  HNNLayer.forward is two launches: the Euclidean product on the
matrix cores and ONE fused row kernel (hd_pb_vae_hnn_tail) for everything after it; the sub-modules stay
individually callable through the same kernel.  Inference path (the VAE's training loop is out of scope):
under autograd the layers raise."""
import torch
import torch.nn as nn
import torch.nn.init as init
from torch.nn.modules.module import Module

from .. import ops

_ACTS = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Sigmoid: "sigmoid", nn.ELU: "elu", nn.Identity: "identity"}


def _act_code(act):
    if act is None:
        return ops.ACT_CODES["identity"]
    if isinstance(act, nn.LeakyReLU):
        if abs(act.negative_slope - 0.01) > 1e-12:
            raise NotImplementedError("fused HypAct supports LeakyReLU(0.01) only")
        return ops.ACT_CODES["leaky_relu"]
    for cls, name in _ACTS.items():
        if isinstance(act, cls):
            if cls is nn.ELU and act.alpha != 1.0:
                raise NotImplementedError("fused HypAct supports ELU(alpha=1) only")
            return ops.ACT_CODES[name]
    raise NotImplementedError("HypAct: activation %r has no fused kernel code" % (act,))


def _ball(manifold):
    if getattr(manifold, "name", None) != "PoincareBall":
        raise NotImplementedError("models/hyp_layers.py runs on the PoincareBall only (its Lorentz class is undefined "
                                  "in the reference as well)")
    return manifold


def _tail(mx, bias, ln, flags, act_code, c=1.0, eps=1e-5):
    """hd_pb_vae_hnn_tail on (..., d) rows; parameters cast to the rows' dtype"""
    ops._dev(mx)
    ops._no_grad(mx, bias, *(ln or ()))
    x2, shape = ops._rows(mx)
    out = torch.empty_like(x2)
    cast = lambda t: None if t is None else t.detach().to(x2.dtype).reshape(-1).contiguous()
    b, w, lb = cast(bias), cast(ln[0]) if ln else None, cast(ln[1]) if ln else None
    ops._lib.call("hd_pb_vae_hnn_tail" + ops._sfx(x2), ops._p(x2), ops._p(b), ops._p(w), ops._p(lb), ops._p(out),
                  x2.shape[0], x2.shape[1], ops.curv(c), float(eps), flags, act_code, ops._stream())
    return out.view(shape)


class HypLinear(nn.Module):
    """models/hyp_layers.py:172-206"""

    def __init__(self, in_dim, out_dim, manifold_in, dropout):
        super(HypLinear, self).__init__()
        self.manifold = _ball(manifold_in)
        self.bias = nn.Parameter(torch.Tensor(1, out_dim))
        self.linear = nn.Linear(in_dim, out_dim, bias=False)
        self.dp = nn.Dropout(dropout)
        self.c = 1.0
        self.scale = 1.
        self.reset_parameters()

    def reset_parameters(self):
        init.constant_(self.bias, 0)

    def _product(self, x):
        if self.training and self.dp.p > 0:
            raise NotImplementedError("models.hyp_layers.HypLinear: dropout is a training feature; call .eval()")
        x2, shape = ops._rows(x)
        w = self.linear.weight.detach()
        mx = ops.gemm(x2, w.to(x2.dtype))
        return mx.view(shape[:-1] + (w.shape[0],))

    def forward(self, x):
        ops._dev(x); ops._no_grad(x, self.linear.weight, self.bias)
        return _tail(self._product(x), self.bias, None, 1, 0, self.c)


class HypAct(Module):
    """models/hyp_layers.py:280-296"""

    def __init__(self, manifold_in, manifold_out, act):
        super(HypAct, self).__init__()
        self.manifold_in = _ball(manifold_in)
        self.manifold_out = _ball(manifold_out)
        self.act = act
        self.c = 1.0

    def forward(self, x):
        return _tail(x, None, None, 4, _act_code(self.act), self.c)


class HypNorm(nn.Module):
    """models/hyp_layers.py:298-319"""

    def __init__(self, in_features, manifold, method='ln'):
        super(HypNorm, self).__init__()
        self.manifold = _ball(manifold)
        if method == 'ln':
            self.norm = nn.LayerNorm(in_features)
        self.c = 1.0

    def forward(self, h):
        return _tail(h, None, (self.norm.weight, self.norm.bias), 2, 0, self.c, self.norm.eps)


class HNNLayer(nn.Module):
    """models/hyp_layers.py:41-61"""

    def __init__(self, in_dim, out_dim, manifold_in, manifold_out, dropout=0., act=nn.ReLU(), use_norm=True):
        super(HNNLayer, self).__init__()
        self.linear = HypLinear(in_dim, out_dim, manifold_in, dropout)
        self.hyp_act = HypAct(manifold_in, manifold_out, act)
        self.norm = HypNorm(out_dim, manifold_in)
        self.use_norm = use_norm

    def forward(self, x):
        ops._dev(x); ops._no_grad(x, *self.parameters())
        flags = 1 | (2 if self.use_norm else 0) | 4
        return _tail(self.linear._product(x), self.linear.bias, (self.norm.norm.weight, self.norm.norm.bias), flags,
                     _act_code(self.hyp_act.act), 1.0, self.norm.norm.eps)
