"""Deterministic inputs for the golden fixtures -- TEST INFRASTRUCTURE ONLY.

Inputs that would be too large to commit (denoiser weights) are regenerated
from numpy's frozen legacy stream (`RandomState`), by oracle/capture_golden.py
when it drives the reference and by the tests when they drive the oracle / the
HIP path.  Small inputs are stored in the .npz next to the reference outputs
as well, so a stream change would be detected rather than silently followed.
"""
import numpy as np
import scipy.sparse as sp

MANIFOLD_DIMS = (2, 14, 16, 65, 256)
MANIFOLD_SCALES = (1e-8, 1e-3, 0.3, 3.0, 30.0)
MANIFOLD_CS = (0.6, 1.0)
MANIFOLD_ROWS = 5          # row 0 is all-zero on purpose (min_norm path)
MLP_WIDTHS = (800, 900, 800, 700)      # models/direction_diffusion.py:50-54


def manifold_case(d, scale, dtype, seed=1432):
    """Two (rows, d) tangent-like inputs and one (d_out, d) matrix."""
    rs = np.random.RandomState(seed + 7 * d + int(1000 * np.log10(scale) + 20000))
    u = (rs.standard_normal((MANIFOLD_ROWS, d)) * scale).astype(dtype)
    v = (rs.standard_normal((MANIFOLD_ROWS, d)) * scale).astype(dtype)
    u[0] = 0
    d_out = max(2, min(d, 24))
    w = (rs.standard_normal((d_out, d)) / np.sqrt(d)).astype(dtype)
    return u, v, w


def mlp_params(dim, seed=1432):
    """fc1..fc5 of the mlp denoiser as [(W, b)] fp32, fan-in scaled normal."""
    rs = np.random.RandomState(seed + dim)
    widths = (dim + 1,) + MLP_WIDTHS + (dim,)
    out = []
    for fi, fo in zip(widths[:-1], widths[1:]):
        w = (rs.standard_normal((fo, fi)) / np.sqrt(fi)).astype(np.float32)
        b = (rs.standard_normal((fo,)) * 0.05).astype(np.float32)
        out.append((w, b))
    return out


def synthetic_graph(kind, n, seed=1432):
    """Symmetric 0/1 scipy CSR adjacency without self loops.

    kind: 'ba' (preferential attachment, m=4: BA.pkl-like), 'ego' (hub + ring of
    neighbours + chords: SynEgo-like), 'community' (two dense blocks joined by one
    edge: SynCommunity-like), 'citation' (sparse power-law, m=2), 'ring' (node i linked to i +- 1, i +- 2 and, one
    node in twenty, to a random node: a graph whose node order has locality, for the halo form of the row sharding)."""
    rs = np.random.RandomState(seed + n)
    rows, cols = [], []

    def add(i, j):
        if i != j:
            rows.extend((i, j)); cols.extend((j, i))

    if kind in ("ba", "citation"):
        m = 4 if kind == "ba" else 2
        targets = list(range(m))
        repeated = []
        for src in range(m, n):
            for t in set(targets):
                add(src, t)
            repeated.extend(set(targets)); repeated.extend([src] * m)
            targets = [repeated[rs.randint(len(repeated))] for _ in range(m)]
    elif kind == "ego":
        for i in range(1, n):
            add(0, i)
            if rs.rand() < 0.4:
                add(i, 1 + rs.randint(n - 1))
    elif kind == "community":
        h = n // 2
        for i in range(n):
            for j in range(i + 1, n):
                same = (i < h) == (j < h)
                if same and rs.rand() < 0.7:
                    add(i, j)
        add(0, n - 1)
    elif kind == "ring":
        for i in range(n):
            add(i, (i + 1) % n)
            add(i, (i + 2) % n)
            if rs.rand() < 0.05:
                add(i, rs.randint(n))
    else:
        raise ValueError(kind)
    a = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    a.data[:] = 1.0
    return a


def diffusion_case(n, dim, k=3, seed=1432, dtype=np.float32):
    """Embeddings inside the unit ball, labels, centres u0, noise, timesteps."""
    rs = np.random.RandomState(seed + 13 * n + dim)
    x = rs.standard_normal((n, dim)) * 0.2
    x = (x / np.maximum(1.0, np.linalg.norm(x, axis=1, keepdims=True) / 0.9)).astype(dtype)
    u0 = (rs.standard_normal((k, dim)) * 0.1).astype(dtype)
    labels = rs.randint(0, k, size=n).astype(np.int64)
    noise = rs.standard_normal((n, dim))
    img = rs.standard_normal((n, dim)).astype(np.float32)
    return dict(x=x, u0=u0, labels=labels, noise=noise, img=img, rs=rs)


def ddpm_state(dim_in, hidden, num_layer, n_steps, seed=1432):
    """Seeded state_dict of ddpm.Model's backbone (ddpm.py:30-41) at any size: nn.Linear-style uniform(+-1/sqrt(fan_in))
    weights and biases, N(0, 1) embedding rows.  The same function feeds the reference at capture time and the
    oracle / HIP path at test time, so a fixture at the real size (hidden 512, 10 layers: 6.3 M weights) only holds
    inputs and outputs."""
    import torch
    rs = np.random.RandomState(seed)
    sd = {}

    def linear(prefix, n_out, n_in):
        b = 1.0 / np.sqrt(n_in)
        sd[prefix + ".weight"] = torch.from_numpy(rs.uniform(-b, b, size=(n_out, n_in)).astype(np.float32))
        sd[prefix + ".bias"] = torch.from_numpy(rs.uniform(-b, b, size=n_out).astype(np.float32))

    for name, d_in, d_out in (("backbone.linear_model1", dim_in, hidden), ("backbone.linear_model2", hidden, dim_in)):
        linear(f"{name}.layers.0", hidden, d_in)
        for i in range(num_layer):
            linear(f"{name}.layers.{i + 1}", hidden, hidden)
        linear(f"{name}.layers.{num_layer + 1}", d_out, hidden)
    sd["backbone.embedding_layer.weight"] = torch.from_numpy(rs.standard_normal((n_steps, hidden)).astype(np.float32))
    return sd
