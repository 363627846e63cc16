"""GPU: hd_moments_noalloc (the one entry point that allocates nothing: its partial sums live in a
device array inside the library) against hd_moments through the C ABI -- bit-identical out3, on
two calls in a row, the second with a smaller input; beyond-maximum sizes are refused."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(x):
    from hypdiff_amd import ops
    lib = ops._lib
    ws = torch.empty(int(lib.load().hd_moments_workspace_doubles()), dtype=torch.float64, device=x.device)
    want = torch.empty(3, dtype=torch.float64, device=x.device)
    got = torch.full((3,), float("nan"), dtype=torch.float64, device=x.device)
    dt = 1 if x.dtype == torch.float64 else 0
    lib.call("hd_moments", ops._p(x), dt, x.numel(), ops._p(ws), ops._p(want), ops._stream())
    lib.call("hd_moments_noalloc", ops._p(x), dt, x.numel(), ops._p(got), ops._stream())
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy()


def test_noalloc_moments_equal_workspace_moments():
    gen = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(15777, 512, device="cuda", generator=gen) * 3 + 0.5          # 1024 partial blocks in use
    small = torch.randn(37, 14, device="cuda", generator=gen, dtype=torch.float64)   # 3 partial blocks: stale slots beyond
    for x in (big, small, big[:1000].contiguous(), small.float()):
        got, want = _both(x)
        assert np.array_equal(got, want), (tuple(x.shape), got, want)
        ref = np.array([x.double().sum().item(), (x.double() ** 2).sum().item(), x.numel()])
        assert np.allclose(got, ref, rtol=1e-12)


def test_noalloc_moments_refuses_oversize_and_null():
    from hypdiff_amd import ops
    out = torch.empty(3, dtype=torch.float64, device="cuda")
    x = torch.zeros(8, device="cuda")
    with pytest.raises(RuntimeError, match="exceeds the maximum"):
        ops._lib.call("hd_moments_noalloc", ops._p(x), 0, (1 << 32) + 1, ops._p(out), ops._stream())
    with pytest.raises(RuntimeError):
        ops._lib.call("hd_moments_noalloc", None, 0, 8, ops._p(out), ops._stream())
    with pytest.raises(RuntimeError):
        ops._lib.call("hd_moments_noalloc", ops._p(x), 0, 0, ops._p(out), ops._stream())
