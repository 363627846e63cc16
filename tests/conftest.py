import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


def assert_close(a, b, rtol=1e-5, atol_scale=1e-6, what="", scale=None, exact=None, cond_slack=8.0, atol_abs=0.0,
                 exact_group="row"):
    """Parity bar for floating point (north_star: 1e-5 relative), element by element.

    |a-b| <= rtol*|b| + atol_scale*scale, scale = max|b| unless the caller passes the natural
    magnitude of the computation (e.g. the pre-clamp denoiser output for a clamped result);
    NaN/Inf must coincide.

    `exact` (optional) is the same formula evaluated in fp64 on the same inputs.  Where the
    reference's own fp32 result b is off from it by delta (ill-conditioned spots such as
    mobius_add next to the ball boundary, where 1 - c|x|^2 cancels), any other correct fp32
    evaluation order is off by a comparable amount, so the bound of an element is widened by
    cond_slack * (the largest |b - exact| of ITS OWN ROW, i.e. along the last axis): an
    ill-conditioned row loosens only itself, well-conditioned rows keep the plain bar.
    exact_group="all" takes the largest |b - exact| of the whole array instead -- only for small
    homogeneous cases (the 5-row golden manifold cases, whose rows are drawn at one scale and share
    one conditioning; a 2-element row is too small a sample of the reference's own error)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    fin = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)]), f"{what}: Inf pattern differs"
    if fin.any():
        scale = np.abs(b[fin]).max() if scale is None else float(scale)
        err = np.where(fin, np.abs(a - np.where(fin, b, 0.0)), 0.0)
        bound = rtol * np.where(fin, np.abs(b), 0.0) + atol_scale * scale + atol_abs
        if exact is not None:
            ex = np.asarray(exact, dtype=np.float64).reshape(b.shape)
            okx = fin & np.isfinite(ex)
            delta = np.where(okx, np.abs(np.where(okx, b, 0.0) - np.where(okx, ex, 0.0)), 0.0)
            if exact_group == "all" or delta.ndim < 2 or delta.shape[-1] == 1:
                rowmax = delta.max() if delta.size else 0.0
            else:
                rowmax = delta.max(axis=-1, keepdims=True)
            bound = bound + cond_slack * rowmax
        worst = (err - bound)[fin].max()
        assert worst <= 0, f"{what}: max err {err[fin].max():.3e} (scale {scale:.3e}), over bound by {worst:.3e}"


def err_stats(a, b, rtol=1e-5, atol_scale=1e-6):
    """(max |a-b| / max|b|, max over elements of |a-b| / (rtol |b| + atol_scale max|b|)): the max-norm
    relative error, and the largest fraction of the PLAIN element-wise bar (no conditioning slack)
    that any element uses -- <= 1 means the result meets north_star's 1e-5 as it stands."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.abs(b).max()
    e = np.abs(a - b)
    return float(e.max() / m), float((e / (rtol * np.abs(b) + atol_scale * m)).max())
