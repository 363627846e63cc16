"""Hyperbolic layers on HIP kernels -- drop-in for the reference's layers/hyp_layers.py.

Same constructors, attribute names and state_dict keys (`weight`, `bias`).  The
three sub-layers stay individually callable; HyperbolicGraphConvolution.forward runs
them as a fused pipeline of 3-4 launches per layer:

    [rownorm] -> MFMA GEMM -> linear tail (+proj, +logmap0: emits the TANGENT row)
              -> CSR SpMM with expmap0/proj/HypAct fused into its write-out
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init
from torch.nn.modules.module import Module

from .. import ops


def act_code(act):
    """Map an activation callable to a fused-kernel code, or None if it has to run as
    a separate elementwise op between two kernels."""
    if act is None:
        return ops.ACT_CODES["identity"]
    name = getattr(act, "__name__", "")
    if name == "<lambda>":
        return None
    table = {F.relu: "relu", torch.relu: "relu", torch.tanh: "tanh", F.tanh: "tanh",
             torch.sigmoid: "sigmoid", F.sigmoid: "sigmoid", F.elu: "elu", F.leaky_relu: "leaky_relu"}
    key = table.get(act)
    return ops.ACT_CODES[key] if key else None


def get_dim_act_curv(args):
    """layers/hyp_layers.py:13-38"""
    if not args.act:
        act = None
    else:
        act = getattr(F, args.act)
    acts = [act] * (args.num_layers - 1)
    dims = [args.feat_dim] + ([args.dim] * (args.num_layers - 1))
    dims += [args.dim]
    acts += [act]
    n_curvatures = args.num_layers
    if args.c is None:
        curvatures = [nn.Parameter(torch.Tensor([1.])) for _ in range(n_curvatures)]
    else:
        curvatures = [torch.tensor([args.c]) for _ in range(n_curvatures)]
        if not args.cuda == -1:
            curvatures = [curv.to(args.device) for curv in curvatures]
    return dims, acts, curvatures


class HypLinear(nn.Module):
    """layers/hyp_layers.py:77-113"""

    def __init__(self, manifold, in_features, out_features, c, dropout, use_bias):
        super(HypLinear, self).__init__()
        self.manifold = manifold
        self.in_features = in_features
        self.out_features = out_features
        self.c = c
        self.dropout = dropout
        self.use_bias = use_bias
        self.bias = nn.Parameter(torch.Tensor(out_features))
        self.weight = nn.Parameter(torch.Tensor(out_features, in_features))
        self.reset_parameters()

    def reset_parameters(self):
        init.xavier_uniform_(self.weight, gain=math.sqrt(2))
        init.constant_(self.bias, 0)

    def _weight(self):
        # weight dropout stays a torch op ahead of the kernel (:99)
        w = self.weight         # the parameter object itself: ops caches its padded copy by identity
        if self.training and self.dropout > 0:
            return F.dropout(w, self.dropout, training=True)
        return w

    def forward(self, x, _tangent=False):
        """proj(mobius_matvec(W, x)) [+ bias branch]; `_tangent` (internal) returns
        logmap0 of the result, which is what HypAgg consumes."""
        man, c = self.manifold, self.c
        w = self._weight()
        if not self.use_bias:
            return man.mobius_matvec(w, x, c, _flags=3 if _tangent else 1)
        # every step below carries its VJP: the row maps and mobius_matvec through their kernels (autograd.py), the
        # ball's mobius_add through hd_pb_map2_bwd; the hyperboloid's mobius_add (logmap0 -> ptransp0 -> expmap,
        # manifolds/hyperboloid.py:115-146) is tensor algebra over those, differentiated by torch
        res = man.mobius_matvec(w, x, c, _flags=1)
        bias = man.proj_tan0(self.bias.view(1, -1), c)
        hyp_bias = man.proj(man.expmap0(bias, c), c)
        res = man.proj(man.mobius_add(res, hyp_bias, c=c), c)
        return man.logmap0(res, c) if _tangent else res

    def extra_repr(self):
        return 'in_features={}, out_features={}, c={}'.format(self.in_features, self.out_features, self.c)


class HypAgg(Module):
    """layers/hyp_layers.py:116-155 (use_att=0 path; dense attention is out of scope)"""

    def __init__(self, manifold, c, in_features, dropout, use_att, local_agg):
        super(HypAgg, self).__init__()
        if use_att:
            raise NotImplementedError("HypAgg(use_att=1) (O(N^2) DenseAtt) is outside the accelerated path; "
                                      "config.py:41 defaults to use_att=0")
        self.manifold = manifold
        self.c = c
        self.in_features = in_features
        self.dropout = dropout
        self.local_agg = local_agg
        self.use_att = use_att

    def forward(self, x, adj, _tangent_in=False, _act=None):
        """`_tangent_in`: x is already logmap0(x).  `_act` = (c_out, code) fuses HypAct."""
        csr = ops.csr_of(adj)
        xt = x if _tangent_in else self.manifold.logmap0(x, c=self.c)
        if torch.is_grad_enabled() and xt.requires_grad:
            from .. import autograd as hag
            cin = ops.curv(self.c)
            if _act is None:
                return hag.HypAggFn.apply(xt, csr, self.manifold.abi, cin, cin, 0, 0)
            return hag.HypAggFn.apply(xt, csr, self.manifold.abi, cin, ops.curv(_act[0]), _act[1], 1)
        if _act is None:
            return ops.agg(self.manifold.abi, csr, xt, self.c)
        return ops.agg(self.manifold.abi, csr, xt, self.c, _act[0], _act[1])

    def extra_repr(self):
        return 'c={}'.format(self.c)


class HypAct(Module):
    """layers/hyp_layers.py:158-178"""

    def __init__(self, manifold, c_in, c_out, act):
        super(HypAct, self).__init__()
        self.manifold = manifold
        self.c_in = c_in
        self.c_out = c_out
        self.act = act

    def forward(self, x):
        code = act_code(self.act)
        if code is not None and not (torch.is_grad_enabled() and x.requires_grad):
            return ops.hyp_act(self.manifold.abi, x, self.c_in, self.c_out, code)
        # unknown callable, or training: the activation runs between the logmap0 kernel and the
        # fused proj_tan0 -> expmap0 -> proj kernel (each with its VJP kernel under autograd)
        if self.act is None:
            self_act = lambda t: t
        else:
            self_act = self.act
        man = self.manifold
        xt = self_act(man.logmap0(x, c=self.c_in))
        xt = man.proj_tan0(xt, c=self.c_out)
        name = "hd_%s_expmap0_proj" % man.abi
        return ops.map1(name, xt, self.c_out)

    def extra_repr(self):
        return 'c_in={}, c_out={}'.format(self.c_in, self.c_out)


class HNNLayer(nn.Module):
    """layers/hyp_layers.py:41-54"""

    def __init__(self, manifold, in_features, out_features, c, dropout, act, use_bias):
        super(HNNLayer, self).__init__()
        self.linear = HypLinear(manifold, in_features, out_features, c, dropout, use_bias)
        self.hyp_act = HypAct(manifold, c, c, act)

    def forward(self, x):
        return self.hyp_act.forward(self.linear.forward(x))


class HyperbolicGraphConvolution(nn.Module):
    """layers/hyp_layers.py:57-74"""

    def __init__(self, manifold, in_features, out_features, c_in, c_out, dropout, act, use_bias, use_att,
                 local_agg):
        super(HyperbolicGraphConvolution, self).__init__()
        self.linear = HypLinear(manifold, in_features, out_features, c_in, dropout, use_bias)
        self.agg = HypAgg(manifold, c_in, out_features, dropout, use_att, local_agg)
        self.hyp_act = HypAct(manifold, c_in, c_out, act)

    def forward(self, input):
        x, adj = input
        code = act_code(self.hyp_act.act)
        xt = self.linear.forward(x, _tangent=True)
        if code is not None:
            h = self.agg.forward(xt, adj, _tangent_in=True, _act=(self.hyp_act.c_out, code))
        else:
            h = self.hyp_act.forward(self.agg.forward(xt, adj, _tangent_in=True))
        return h, adj
