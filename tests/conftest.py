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


def assert_close(a, b, rtol=1e-5, atol_scale=1e-6, what="", scale=None, exact=None, cond_slack=8.0, atol_abs=0.0):
    """Parity bar for floating point (north_star: 1e-5 relative).

    |a-b| <= rtol*|b| + atol_scale*scale elementwise, scale = max|b| unless the caller
    passes the natural magnitude of the computation (e.g. the pre-clamp denoiser
    output for a clamped result); NaN/Inf must coincide.

    `exact` (optional) is the same formula evaluated in fp64 on the same inputs.  Where
    the reference's own fp32 result b is off from it by delta (ill-conditioned spots such
    as mobius_add next to the ball boundary, where 1 - c|x|^2 cancels), any other correct
    fp32 evaluation order is off by a comparable amount, so the bound is widened to
    cond_slack * max|b - exact| there.  Well-conditioned cases keep the plain bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    fin = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)]), f"{what}: Inf pattern differs"
    if fin.any():
        scale = np.abs(b[fin]).max() if scale is None else float(scale)
        err = np.abs(a[fin] - b[fin])
        bound = rtol * np.abs(b[fin]) + atol_scale * scale + atol_abs
        if exact is not None:
            ex = np.asarray(exact, dtype=np.float64).reshape(b.shape)
            okx = fin & np.isfinite(ex)
            if okx.any():
                bound = bound + cond_slack * np.abs(b[okx] - ex[okx]).max()
        worst = (err - bound).max()
        assert worst <= 0, f"{what}: max err {err.max():.3e} (scale {scale:.3e}), over bound by {worst:.3e}"
