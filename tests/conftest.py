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


def assert_close(a, b, rtol=1e-5, atol_scale=1e-6, what="", scale=None):
    """Parity bar for floating point (north_star: 1e-5 relative).

    |a-b| <= rtol*|b| + atol_scale*scale elementwise, scale = max|b| unless the caller
    passes the natural magnitude of the computation (e.g. the pre-clamp denoiser
    output for a clamped result); NaN/Inf must coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    fin = np.isfinite(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)]), f"{what}: Inf pattern differs"
    if fin.any():
        scale = np.abs(b[fin]).max() if scale is None else float(scale)
        err = np.abs(a[fin] - b[fin])
        bound = rtol * np.abs(b[fin]) + atol_scale * scale
        worst = (err - bound).max()
        assert worst <= 0, f"{what}: max err {err.max():.3e} (scale {scale:.3e}), over bound by {worst:.3e}"
