"""Hyperboloid on HIP kernels -- drop-in for the reference's manifolds/hyperboloid.py.
Column 0 is the time coordinate; like the reference, proj/expmap0/logmap0 take 2-D
inputs (hyperboloid.py:47,95,108 index dim=1)."""
import os

import torch

from .. import ops
from .base import Manifold


_NO_SMALLK = bool(os.environ.get("HD_NO_SMALLK"))      # debugging knob: force the wide HypLinear path


class Hyperboloid(Manifold):
    abi = "hb"

    def __init__(self):
        super(Hyperboloid, self).__init__()
        self.name = 'Hyperboloid'
        self.eps = {torch.float32: 1e-7, torch.float64: 1e-15}
        self.min_norm = 1e-15
        self.max_norm = 1e6

    @staticmethod
    def _2d(x):
        if x.dim() != 2:
            raise ValueError("Hyperboloid maps take 2-D (rows, d+1) inputs, as in the reference")
        return x

    def proj(self, x, c):                       # hyperboloid.py:43-52
        return ops.map1("hd_hb_proj", self._2d(x), c)

    def proj_tan0(self, u, c):                  # :64-68
        return ops.map1("hd_hb_proj_tan0", self._2d(u), c)

    def expmap0(self, u, c):                    # :90-101
        return ops.map1("hd_hb_expmap0", self._2d(u), c)

    def logmap0(self, x, c):                    # :103-113
        return ops.map1("hd_hb_logmap0", self._2d(x), c)

    def sqdist(self, x, y, c):                  # :35-41 -> (rows, 1)
        return ops.sqdist("hd_hb_sqdist", x, y, c).unsqueeze(-1)

    def mobius_matvec(self, m, x, c, _flags=0):  # :120-123
        if isinstance(x, ops.SparseRows):          # sparse logmap0 rows (first layer, identity features): gather of W^T rows
            mu = ops.sparse_matvec(x, m)
            out = torch.empty_like(mu)
            ops._lib.call("hd_hb_linear_epilogue" + ops._sfx(mu), ops._p(mu), ops._p(out), mu.shape[0], mu.shape[1],
                          ops.curv(c), _flags, ops._stream())
            return out
        ops._dev(m, x)
        x2 = self._2d(x).contiguous()
        if torch.is_grad_enabled() and (m.requires_grad or x.requires_grad):
            from .. import autograd as hag         # training: forward + VJP kernels (lp_train.py:66-82)
            return hag.HypLinearFn.apply(x2, m, "hb", ops.curv(c), _flags)
        if m.dtype != x2.dtype:
            m = m.to(x2.dtype)
        K, N = x2.shape[1], m.shape[0]
        m = ops.match_padded(x2, m)
        # (the two 64-lane-per-row kernels below pay off from ~128 / 256 columns; narrower rows keep the
        # generic row kernels, which put 4 or 16 lanes on a row)
        if (x2.dtype == torch.float32 and m.dtype == torch.float32 and K <= 64 and 128 <= N <= 512 and N % 4 == 0
                and x2.shape[0] > 0 and not _NO_SMALLK):
            # narrow input (first layer of the graph-set configs): logmap0, the fp32 contraction and the
            # row tail in one launch -- nothing between x and the result touches memory
            out = torch.empty(x2.shape[0], N, dtype=torch.float32, device=x2.device)
            if ops.GEMM_MODE != "exact" and K <= 32 and N % 16 == 0:
                # the contraction on the matrix cores (3 bf16 limbs, fp32-faithful), one k-step per output tile
                ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", ops._p(x2), ops._p(ops.smallk_image_cached(m)),
                              ops._p(out), x2.shape[0], K, N, ops.curv(c), _flags, ops._stream())
            else:
                ops._lib.call("hd_hb_linear_smallk_f32", ops._p(x2), ops._p(ops.transposed_cached(m)), ops._p(out),
                              x2.shape[0], K, N, ops.curv(c), _flags, ops._stream())
            return out
        fmt = ops.limb_rows_fmt(x2, m)
        if fmt:
            # logmap0 writes the limb image of the GEMM operand directly (one pass over x)
            mu, _ = ops.gemm_limbs(ops.row_pack(x2, 1, c, fmt), ops.weight_limbs(m, fmt)[0])
        else:
            mu = ops.gemm(ops.map1("hd_hb_logmap0", x2, c), m)
        out = torch.empty_like(mu)
        if mu.dtype == torch.float32 and 256 <= mu.shape[1] <= 512 and mu.shape[1] % 4 == 0 and mu.is_contiguous():
            ops._lib.call("hd_hb_linear_tail_rows_f32", ops._p(mu), ops._p(out), mu.shape[0], mu.shape[1], ops.curv(c),
                          _flags, ops._stream())
        else:
            ops._lib.call("hd_hb_linear_epilogue" + ops._sfx(mu), ops._p(mu), ops._p(out), mu.shape[0], mu.shape[1],
                          ops.curv(c), _flags, ops._stream())
        return out

    # ---- bias path of HypLinear (use_bias=1, not the default): tensor algebra over the
    # kernels above; hyperboloid.py:31-33,54-62,70-78,115-118,133-146
    def minkowski_dot(self, x, y, keepdim=True):
        res = torch.sum(x * y, dim=-1) - 2 * x[..., 0] * y[..., 0]
        return res.view(res.shape + (1,)) if keepdim else res

    def minkowski_norm(self, u, keepdim=True):
        return torch.sqrt(torch.clamp(self.minkowski_dot(u, u, keepdim=keepdim), min=self.eps[u.dtype]))

    def proj_tan(self, u, x, c):
        d = x.size(1) - 1
        ux = torch.sum(x.narrow(-1, 1, d) * u.narrow(-1, 1, d), dim=1, keepdim=True)
        t0 = ux / torch.clamp(x[:, 0:1], min=self.eps[x.dtype])
        return torch.cat([t0, u[:, 1:]], dim=1)

    def expmap(self, u, x, c):
        sqrtK = ops.curv(c).sqrt_K
        normu = torch.clamp(self.minkowski_norm(u), max=self.max_norm)
        theta = torch.clamp(normu / sqrtK, min=self.min_norm)
        th = theta.clamp(-15, 15)
        return self.proj(th.cosh() * x + th.sinh() * u / theta, c)

    def ptransp0(self, x, u, c):
        sqrtK = ops.curv(c).sqrt_K
        x0 = x.narrow(-1, 0, 1)
        y = x[:, 1:]
        y_norm = torch.clamp(torch.norm(y, p=2, dim=1, keepdim=True), min=self.min_norm)
        y_n = y / y_norm
        v = torch.cat([-y_norm, (sqrtK - x0) * y_n], dim=1)
        alpha = torch.sum(y_n * u[:, 1:], dim=1, keepdim=True) / sqrtK
        return self.proj_tan(u - alpha * v, x, c)

    def mobius_add(self, x, y, c):
        if y.shape[0] != x.shape[0]:
            y = y.expand(x.shape[0], -1)
        u = self.logmap0(y.contiguous(), c)
        return self.expmap(self.ptransp0(x, u, c), x, c)

    def init_weights(self, w, c, irange=1e-5):
        with torch.no_grad():
            w.uniform_(-irange, irange)     # on w itself (not .data): the version counter must move
        return w
