"""Oracle: hyperbolic k-means (numpy fp64, vectorised over samples) -- TEST INFRASTRUCTURE ONLY.
Restates hyperbolic_learning/hyperbolic_kmeans/hkmeans.py:20-70,97-217 and util_hk.py:14-16,28-61,
112-133 of the reference, including the `**-1/2` precedence of minkowski_distance_gradient
(hkmeans.py:27), which makes the weight d/(p^2-1)."""
import numpy as np


def to_hyperboloid(Y, eps=1e-6):
    r2 = np.linalg.norm(Y, axis=1) ** 2
    s = 2 / (1 - r2 + eps)
    return np.concatenate([s[:, None] * Y, (s * (1 + r2) / 2)[:, None]], axis=1)


def hdot(U, v):
    return U[..., :-1] @ v[:-1] - U[..., -1] * v[-1]


def compute_mean(theta, X, num_rounds=10, alpha=0.1):
    c = theta.copy()
    for _ in range(num_rounds):
        p = hdot(X, c)
        with np.errstate(invalid="ignore"):
            d = np.arccosh(-p)
        d = np.where(np.isnan(d), 1e-6, d)
        grad = np.mean((d / (p * p - 1))[:, None] * X, axis=0)
        g = -grad
        tangent = g + hdot(c, g) * c
        v = -alpha * tangent
        nv = np.linalg.norm(v)
        c = np.cosh(nv) * c + np.sinh(nv) * v / (nv + 1e-6)
    return c


def to_poincare(x, eps=1e-6):
    p = x[:-1] / ((x[-1] + 1) + eps)
    n = np.linalg.norm(p)
    return p / n - 0.1 if n >= 1 else p


def poincare_dist(X, c, eps=1e-5):
    a = 1 + 2 * np.linalg.norm(X - c, axis=1) ** 2 / ((1 - np.linalg.norm(X, axis=1) ** 2) * (1 - np.linalg.norm(c) ** 2) + eps)
    return np.arccosh(a)


def fit(X, centroids, labels, K, max_epochs):
    X = np.asarray(X, dtype=np.float64)
    nr = np.linalg.norm(X, axis=1)
    if (nr > 1).any():
        X = X / nr.max()
    cen, lab = np.array(centroids, dtype=np.float64), np.array(labels).copy()
    H = to_hyperboloid(X)
    for _ in range(max_epochs):
        new = cen.copy()
        for k in range(K):
            m = lab == k
            if m.any():
                new[k] = to_poincare(compute_mean(to_hyperboloid(cen[k:k + 1])[0], H[m]))
        cen = new
        D = np.stack([poincare_dist(X, cen[k]) for k in range(K)], axis=1)
        lab = np.argmin(D, axis=1)
    return cen, lab
