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


# Every assert_close() records how much of its bound the comparison used.  With HD_PARITY_MARGIN_OUT=<file> the
# session writes the table as JSON (tools/parity_margin.sh -> profiles/r03_parity_margin.json): the margin of the
# parity claim is on file, not only "passed".
MARGINS = {}
_CURRENT_TEST = [""]


@pytest.fixture(autouse=True)
def _margin_test_name(request):
    _CURRENT_TEST[0] = request.node.nodeid
    yield
    _CURRENT_TEST[0] = ""


def record_margin(what, **fields):
    key = f"{_CURRENT_TEST[0]} :: {what}"
    old = MARGINS.get(key)
    if old is None or fields.get("bound_used", 0.0) >= old.get("bound_used", 0.0):
        MARGINS[key] = fields


def pytest_sessionfinish(session, exitstatus):
    out = os.environ.get("HD_PARITY_MARGIN_OUT")
    if out and MARGINS:
        import json
        rows = [dict(check=k, **v) for k, v in sorted(MARGINS.items())]
        worst = max(rows, key=lambda r: r["bound_used"])
        with open(out, "w") as f:
            json.dump({"note": "per assert_close() of the GPU parity tests: max_norm_rel = max|got-ref| / max|ref|; "
                               "plain_bar_used = largest |got-ref| / (1e-5 |ref| + 1e-6 max|ref|) over the elements (<= 1: meets "
                               "north_star's bar as it stands); bound_used = the same against the bound the test asserts "
                               "(its rtol / atol_scale / scale and, where given, the per-row conditioning slack); "
                               "slack_rows = fraction of rows whose bound needed the conditioning slack at all",
                       "exit_status": int(exitstatus), "checks": len(rows), "worst_bound_used": worst, "rows": rows}, f, indent=1)


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
        plain = 1e-5 * np.where(fin, np.abs(b), 0.0) + 1e-6 * (np.abs(b[fin]).max() or 1.0)
        rec = dict(max_norm_rel=float(err[fin].max() / (np.abs(b[fin]).max() or 1.0)),
                   plain_bar_used=float((err / plain)[fin].max()),
                   bound_used=float((err / np.maximum(bound, 1e-300))[fin].max()),
                   rtol=rtol, atol_scale=atol_scale, elements=int(fin.sum()))
        if exact is not None:
            plain_t = rtol * np.where(fin, np.abs(b), 0.0) + atol_scale * scale + atol_abs
            over = (err > plain_t) & fin
            rec["slack_rows"] = float(over.any(axis=-1).mean()) if over.ndim >= 2 else float(over.any())
            rec["cond_slack"] = cond_slack
        record_margin(what or "unnamed", **rec)
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
