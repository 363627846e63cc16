"""PoincareBall on HIP kernels -- drop-in for the reference's manifolds/poincare.py.

Same method names, argument order and return shapes; each hot method is ONE fused
kernel launch (hd_pb_* of include/hypdiff.h) instead of a chain of aten ops."""
import torch

from .. import ops
from .base import Manifold


class PoincareBall(Manifold):
    abi = "pb"

    def __init__(self):
        super(PoincareBall, self).__init__()
        self.name = 'PoincareBall'
        self.min_norm = 1e-15
        self.eps = {torch.float32: 4e-3, torch.float64: 1e-5}

    # ---- hot path --------------------------------------------------------------
    def proj(self, x, c):                       # poincare.py:42-47
        return ops.map1("hd_pb_proj", x, c)

    def expmap0(self, u, c):                    # :73-77
        return ops.map1("hd_pb_expmap0", u, c)

    def logmap0(self, p, c):                    # :79-83
        return ops.map1("hd_pb_logmap0", p, c)

    def proj_tan(self, u, p, c):                # :49-50
        return u

    def proj_tan0(self, u, c):                  # :52-53
        return u

    def mobius_add(self, x, y, c, dim=-1):      # :85-91
        if dim not in (-1, x.dim() - 1):
            raise NotImplementedError("mobius_add kernels reduce over the last dimension")
        return ops.map2("hd_pb_mobius_add", x, y, c)

    def expmap(self, u, p, c):                  # :55-64
        return ops.map2("hd_pb_expmap", p, u, c)

    def logmap(self, p1, p2, c):                # :66-71
        return ops.map2("hd_pb_logmap", p1, p2, c)

    def sqdist(self, p1, p2, c):                # :25-31 -> (...,)
        return ops.sqdist("hd_pb_sqdist", p1, p2, c)

    def mobius_matvec(self, m, x, c, _flags=0):  # :93-102
        """x @ m^T on the matrix cores, then the row-norm / tanh / artanh tail in one
        fused row kernel.  `_flags` (internal): 1 = also proj, 2 = then logmap0."""
        if isinstance(x, ops.SparseRows):          # sparse feature rows (first layer, identity features): gather of W^T rows
            mx = ops.sparse_matvec(x, m)
            out = torch.empty_like(mx)
            ops._lib.call("hd_pb_matvec_epilogue" + ops._sfx(mx), ops._p(mx), ops._p(x.xnorm), ops._p(out), mx.shape[0],
                          mx.shape[1], ops.curv(c), _flags, ops._stream())
            return out
        ops._dev(m, x)
        x2, shape = ops._rows(x)
        if torch.is_grad_enabled() and (m.requires_grad or x.requires_grad):
            from .. import autograd as hag         # training: forward + VJP kernels (lp_train.py:66-82)
            return hag.HypLinearFn.apply(x2, m, "pb", ops.curv(c), _flags).view(shape[:-1] + (m.shape[0],))
        if m.dtype != x2.dtype:
            m = m.to(x2.dtype)
        m = ops.match_padded(x2, m)
        fmt = ops.limb_rows_fmt(x2, m)
        if fmt:
            img, xn = ops.row_pack(x2, 2, fmt=fmt)   # row norms and the operand's limb image in one pass over x
            mx, _ = ops.gemm_limbs(img, ops.weight_limbs(m, fmt)[0])
        else:
            xn = ops.rownorm(x2)
            mx = ops.gemm(x2, m)
        out = torch.empty_like(mx)
        ops._lib.call("hd_pb_matvec_epilogue" + ops._sfx(mx), ops._p(mx), ops._p(xn), ops._p(out), mx.shape[0],
                      mx.shape[1], ops.curv(c), _flags, ops._stream())
        return out.view(shape[:-1] + (m.shape[0],))

    # ---- off the hot path (optimizer / analysis helpers): plain tensor algebra ----
    def _lambda_x(self, x, c):                  # :33-35
        x_sqnorm = torch.sum(x.data.pow(2), dim=-1, keepdim=True)
        return 2 / (1. - c * x_sqnorm).clamp_min(self.min_norm)

    def egrad2rgrad(self, p, dp, c):            # :37-40
        dp /= self._lambda_x(p, c).pow(2)
        return dp

    def init_weights(self, w, c, irange=1e-5):  # :104-106
        with torch.no_grad():
            w.uniform_(-irange, irange)     # on w itself (not .data): the version counter must move
        return w

    def inner(self, x, c, u, v=None, keepdim=False):   # :120-124
        if v is None:
            v = u
        return self._lambda_x(x, c) ** 2 * (u * v).sum(dim=-1, keepdim=keepdim)

    def ptransp0(self, x, u, c):                # :136-138
        return 2 * u / self._lambda_x(x, c).clamp_min(self.min_norm)
