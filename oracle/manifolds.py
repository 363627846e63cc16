"""Oracle: Poincare-ball and hyperboloid row maps (torch CPU, vectorised).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Conventions follow the reference: curvature c > 0, ball radius 1/sqrt(c),
hyperboloid -x0^2 + |y|^2 = -K with K = 1/c and column 0 the time coordinate.
`c` is a python float here; a 1-element tensor is reduced with float().
"""
import torch

MIN_NORM = 1e-15            # manifolds/poincare.py:22, manifolds/hyperboloid.py:22
PB_EPS = {torch.float32: 4e-3, torch.float64: 1e-5}      # manifolds/poincare.py:23
HB_EPS = {torch.float32: 1e-7, torch.float64: 1e-15}     # manifolds/hyperboloid.py:21
HB_MAX_NORM = 1e6                                        # manifolds/hyperboloid.py:23


def curv(c):
    """(c, sqrt(c), K=1/c, sqrt(K)) as python floats.  A python-float c is handled
    in double; a 1-element tensor c (layers/hyp_layers.py:35, models/base_models.py:24)
    takes `c ** 0.5`, `1. / c` and `K ** 0.5` in the tensor's own dtype, as the
    reference's expressions do."""
    if torch.is_tensor(c):
        c = c.detach().reshape(-1)[:1].cpu()
        K = 1.0 / c
        return float(c), float(c ** 0.5), float(K), float(K ** 0.5)
    c = float(c)
    return c, c ** 0.5, 1.0 / c, (1.0 / c) ** 0.5


def _c(c):
    return curv(c)[0]


def _sc(c):
    return curv(c)[1]


def _K(c):
    return curv(c)[2]


def _sK(c):
    return curv(c)[3]


# ---- clamped hyperbolic functions: utils/math_utils.py:6-15 -------------------
def tanh15(x):
    return torch.tanh(torch.clamp(x, -15.0, 15.0))


def cosh15(x):
    return torch.cosh(torch.clamp(x, -15.0, 15.0))


def sinh15(x):
    return torch.sinh(torch.clamp(x, -15.0, 15.0))


def artanh(x):
    """utils/math_utils.py:30-36 -- clamp in x's dtype, evaluate in fp64."""
    x = torch.clamp(x, -1 + 1e-15, 1 - 1e-15)
    z = x.double()
    return (0.5 * (torch.log(1 + z) - torch.log(1 - z))).to(x.dtype)


def arcosh(x):
    """utils/math_utils.py:57-63."""
    x = torch.clamp(x, min=1.0 + 1e-15)
    z = x.double()
    return torch.log(torch.clamp(z + torch.sqrt(z * z - 1), min=1e-15)).to(x.dtype)


def _norm(x):
    return torch.clamp(torch.linalg.vector_norm(x, ord=2, dim=-1, keepdim=True), min=MIN_NORM)


# ---- Poincare ball ------------------------------------------------------------
def pb_proj(x, c):
    """manifolds/poincare.py:42-47."""
    n = _norm(x)
    maxnorm = (1 - PB_EPS[x.dtype]) / _sc(c)
    return torch.where(n > maxnorm, x / n * maxnorm, x)


def pb_expmap0(u, c):
    """manifolds/poincare.py:73-77."""
    sc = _sc(c)
    n = _norm(u)
    return tanh15(sc * n) * u / (sc * n)


def pb_logmap0(p, c):
    """manifolds/poincare.py:79-83."""
    sc = _sc(c)
    n = _norm(p)
    return (1.0 / sc * artanh(sc * n) / n) * p


def pb_mobius_add(x, y, c):
    """manifolds/poincare.py:85-91."""
    c = _c(c)
    x2 = (x * x).sum(-1, keepdim=True)
    y2 = (y * y).sum(-1, keepdim=True)
    xy = (x * y).sum(-1, keepdim=True)
    num = (1 + 2 * c * xy + c * y2) * x + (1 - c * x2) * y
    den = 1 + 2 * c * xy + c ** 2 * x2 * y2
    return num / torch.clamp(den, min=MIN_NORM)


def pb_lambda_x(x, c):
    """manifolds/poincare.py:33-35 (no gradient through it in the reference)."""
    c = _c(c)
    x = x.detach()                      # the reference computes it on x.data: a constant for autograd
    return 2 / torch.clamp(1.0 - c * (x * x).sum(-1, keepdim=True), min=MIN_NORM)


def pb_expmap(u, p, c):
    """manifolds/poincare.py:55-64."""
    sc = _sc(c)
    un = _norm(u)
    second = tanh15(sc / 2 * pb_lambda_x(p, c) * un) * u / (sc * un)
    return pb_mobius_add(p, second, c)


def pb_logmap(p1, p2, c):
    """manifolds/poincare.py:66-71."""
    sc = _sc(c)
    sub = pb_mobius_add(-p1, p2, c)
    sn = _norm(sub)
    lam = pb_lambda_x(p1, c)
    return 2 / sc / lam * artanh(sc * sn) * sub / sn


def pb_sqdist(p1, p2, c):
    """manifolds/poincare.py:25-31 -> shape (E,)."""
    sc = _sc(c)
    d = artanh(sc * torch.linalg.vector_norm(pb_mobius_add(-p1, p2, c), ord=2, dim=-1))
    return (d * 2 / sc) ** 2


def pb_mobius_matvec(m, x, c):
    """manifolds/poincare.py:93-102; m is (d_out, d_in)."""
    sc = _sc(c)
    xn = _norm(x)
    mx = x @ m.transpose(-1, -2)
    mxn = _norm(mx)
    res = tanh15(mxn / xn * artanh(sc * xn)) * mx / (mxn * sc)
    zero_row = (mx == 0).all(dim=-1, keepdim=True)
    return torch.where(zero_row, torch.zeros((), dtype=res.dtype), res)


def pb_ptransp0(x, u, c):
    """manifolds/poincare.py:136-138."""
    return 2 * u / torch.clamp(pb_lambda_x(x, c), min=MIN_NORM)


# ---- Hyperboloid (2-D inputs only, column 0 = time) -----------------------------
def hb_minkowski_dot(x, y):
    """manifolds/hyperboloid.py:25-29 (keepdim form)."""
    return ((x * y).sum(-1) - 2 * x[..., 0] * y[..., 0]).unsqueeze(-1)


def hb_proj(x, c):
    """manifolds/hyperboloid.py:43-52."""
    K = _K(c)
    y = x[:, 1:]
    ysq = torch.linalg.vector_norm(y, ord=2, dim=1, keepdim=True) ** 2
    x0 = torch.sqrt(torch.clamp(K + ysq, min=HB_EPS[x.dtype]))
    return torch.cat([x0, y], dim=1)


def hb_proj_tan0(u, c):
    """manifolds/hyperboloid.py:64-68 (written u - vals in the reference)."""
    out = u.clone()
    out[:, 0:1] = u[:, 0:1] - u[:, 0:1]
    return out


def hb_expmap0(u, c):
    """manifolds/hyperboloid.py:90-101 (ends with proj)."""
    K, sK = _K(c), _sK(c)
    x = u[:, 1:]
    n = torch.clamp(torch.linalg.vector_norm(x, ord=2, dim=1, keepdim=True), min=MIN_NORM)
    theta = n / sK
    res = torch.cat([sK * cosh15(theta), sK * sinh15(theta) * x / n], dim=1)
    return hb_proj(res, c)


def hb_logmap0(x, c):
    """manifolds/hyperboloid.py:103-113."""
    K, sK = _K(c), _sK(c)
    y = x[:, 1:]
    n = torch.clamp(torch.linalg.vector_norm(y, ord=2, dim=1, keepdim=True), min=MIN_NORM)
    theta = torch.clamp(x[:, 0:1] / sK, min=1.0 + HB_EPS[x.dtype])
    return torch.cat([torch.zeros_like(x[:, 0:1]), sK * arcosh(theta) * y / n], dim=1)


def hb_mobius_matvec(m, x, c):
    """manifolds/hyperboloid.py:120-123."""
    return hb_expmap0(hb_logmap0(x, c) @ m.transpose(-1, -2), c)


def hb_sqdist(x, y, c):
    """manifolds/hyperboloid.py:35-41 -> shape (E,1), clamped at 50."""
    K = _K(c)
    theta = torch.clamp(-hb_minkowski_dot(x, y) / K, min=1.0 + HB_EPS[x.dtype])
    return torch.clamp(K * arcosh(theta) ** 2, max=50.0)


def hb_minkowski_norm(u):
    """manifolds/hyperboloid.py:31-33."""
    return torch.sqrt(torch.clamp(hb_minkowski_dot(u, u), min=HB_EPS[u.dtype]))


def hb_proj_tan(u, x, c):
    """manifolds/hyperboloid.py:54-62."""
    ux = (x[:, 1:] * u[:, 1:]).sum(dim=1, keepdim=True)
    t0 = ux / torch.clamp(x[:, 0:1], min=HB_EPS[x.dtype])
    return torch.cat([t0, u[:, 1:]], dim=1)


def hb_expmap(u, x, c):
    """manifolds/hyperboloid.py:70-78."""
    sK = _sK(c)
    normu = torch.clamp(hb_minkowski_norm(u), max=HB_MAX_NORM)
    theta = torch.clamp(normu / sK, min=MIN_NORM)
    return hb_proj(cosh15(theta) * x + sinh15(theta) * u / theta, c)


def hb_ptransp0(x, u, c):
    """manifolds/hyperboloid.py:133-146."""
    sK = _sK(c)
    x0 = x[:, 0:1]
    y = x[:, 1:]
    yn = torch.clamp(torch.linalg.vector_norm(y, ord=2, dim=1, keepdim=True), min=MIN_NORM)
    yhat = y / yn
    v = torch.cat([-yn, (sK - x0) * yhat], dim=1)
    alpha = (yhat * u[:, 1:]).sum(dim=1, keepdim=True) / sK
    return hb_proj_tan(u - alpha * v, x, c)


def hb_mobius_add(x, y, c):
    """manifolds/hyperboloid.py:115-118 (bias path of HypLinear)."""
    return hb_expmap(hb_ptransp0(x, hb_logmap0(y, c), c), x, c)


MANIFOLDS = {
    "PoincareBall": dict(proj=pb_proj, expmap0=pb_expmap0, logmap0=pb_logmap0,
                         mobius_matvec=pb_mobius_matvec, mobius_add=pb_mobius_add,
                         sqdist=pb_sqdist, proj_tan0=lambda u, c: u),
    "Hyperboloid": dict(proj=hb_proj, expmap0=hb_expmap0, logmap0=hb_logmap0,
                        mobius_matvec=hb_mobius_matvec, mobius_add=hb_mobius_add,
                        sqdist=hb_sqdist, proj_tan0=hb_proj_tan0),
}
