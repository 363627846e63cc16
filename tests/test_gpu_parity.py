"""GPU: the HIP path (through the C ABI) against the reference-captured golden vectors
and against the oracle on seeded inputs.  Bar: 1e-5 relative (north_star) for fp32,
tighter for fp64; CSR indexing bit-exact."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import adjacency, diffusion as odiff, golden_inputs as gi, layers as olayers
from oracle import manifolds as om

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "f64": torch.float64}


def G(a, dtype=None):
    t = torch.from_numpy(np.array(a, copy=True)).cuda()
    return t if dtype is None else t.to(dtype)


def H(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from hypdiff_amd import _lib
    assert _lib.load().hd_version() >= 100
    assert torch.cuda.is_available()


@pytest.fixture
def gemm_mode(request):
    """Run a test under one of the fp32 GEMM kernel families (ops.GEMM_MODE), then restore."""
    from hypdiff_amd import ops
    old, ops.GEMM_MODE = ops.GEMM_MODE, request.param
    yield request.param
    ops.GEMM_MODE = old


BOTH_GEMMS = pytest.mark.parametrize("gemm_mode", ["bf16x3", "f16x2", "exact"], indirect=True)


# ---------------------------------------------------------------------------------
@pytest.mark.parametrize("mname", ["PoincareBall", "Hyperboloid"])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_manifold_golden(golden, mname, dt):
    import hypdiff_amd.manifolds as hm
    man = getattr(hm, mname)()
    g = golden("manifold_" + mname)
    tol = dict(rtol=1e-5, atol_scale=1e-6) if dt == "f32" else dict(rtol=1e-9, atol_scale=1e-10)
    # two-point maps in fp64: the fp64 ball margin is 1e-5, so points clipped by proj amplify
    # rounding by ~1e5 (1 - c|x|^2 cancels); bar = the north-star 1e-5, plus an absolute floor
    # for squared distances of (nearly) identical points
    tol2 = tol if dt == "f32" else dict(rtol=1e-5, atol_scale=1e-6, atol_abs=1e-12)
    with torch.no_grad():
        for c in gi.MANIFOLD_CS:
            for d in gi.MANIFOLD_DIMS:
                for s in gi.MANIFOLD_SCALES:
                    k = f"{dt}/c{c}/d{d}/s{s:g}/"
                    u, v, w, p, q = (G(g[k + x]) for x in "uvwpq")
                    # 5 rows per case, all at one scale: one conditioning for the whole case (exact_group)
                    chk = lambda out, name, **kw: assert_close(H(out), g[k + name], what=k + name, exact_group="all",
                                                               **{**tol, **kw})
                    # the same formulas in fp64 on the same inputs: conditioning yardstick
                    pre = "pb_" if mname == "PoincareBall" else "hb_"
                    p64, q64, v64, w64 = (torch.from_numpy(g[k + x]).double() for x in "pqvw")
                    ex = lambda name, *a: getattr(om, pre + name)(*a, c).numpy()
                    chk(man.proj(u, c), "proj")
                    chk(man.expmap0(u, c), "expmap0")
                    chk(man.logmap0(p, c), "logmap0", exact=ex("logmap0", p64))
                    chk(man.logmap0(p, torch.tensor([c]).cuda()), "logmap0_ctensor", exact=ex("logmap0", p64))
                    chk(man.mobius_matvec(w, p, c), "mobius_matvec", exact=ex("mobius_matvec", w64, p64))
                    chk(man.sqdist(p, q, c), "sqdist", exact=ex("sqdist", p64, q64), **tol2)
                    chk(man.proj_tan0(u, c), "proj_tan0")
                    # Hyperboloid.mobius_add (bias path only) on points saturated at cosh(15): the
                    # Minkowski norm of a ~1e7-sized vector cancels to rounding noise that decides
                    # which clamp fires -- chaotic in the reference itself, not comparable
                    if not (mname == "Hyperboloid" and s >= 3):
                        chk(man.mobius_add(p, q, c), "mobius_add", exact=ex("mobius_add", p64, q64), **tol2)
                    if mname == "PoincareBall":
                        chk(man.expmap(v, p, c), "expmap", exact=ex("expmap", v64, p64), **tol2)
                        chk(man.logmap(p, q, c), "logmap", exact=ex("logmap", p64, q64), **tol2)


@pytest.mark.parametrize("mname", ["PoincareBall", "Hyperboloid"])
@pytest.mark.parametrize("d", [3, 20, 64, 128, 257, 512, 1024, 2708])
def test_manifold_vs_oracle_sizes(mname, d):
    """All register-tile configurations (G=4/16/64, scalar and 16-byte loads) and the
    streaming path for rows beyond the tile (d=2708: identity features of config 4)."""
    import hypdiff_amd.manifolds as hm
    man = getattr(hm, mname)()
    pre = "pb_" if mname == "PoincareBall" else "hb_"
    rs = np.random.RandomState(d)
    for n in (1, 37, 300):
        for scale in (0.05, 2.0):
            u = (rs.standard_normal((n, d)) * scale / np.sqrt(d)).astype(np.float32)
            u[0] = 0
            for c in (0.6, 1.0):
                ut = torch.from_numpy(u)
                with torch.no_grad():
                    p = getattr(om, pre + "proj")(getattr(om, pre + "expmap0")(ut, c), c)
                    assert_close(H(man.expmap0(G(u), c)), getattr(om, pre + "expmap0")(ut, c), what="expmap0")
                    assert_close(H(man.proj(G(u), c)), getattr(om, pre + "proj")(ut, c), what="proj")
                    assert_close(H(man.logmap0(p.cuda(), c)), getattr(om, pre + "logmap0")(p, c), what="logmap0")
                    if d <= 1024:
                        from hypdiff_amd import ops
                        got = ops.hyp_act(man.abi, p.cuda(), c, 1.0, 1)
                        assert_close(H(got), olayers.hyp_act(mname, p, c, 1.0), what="hyp_act")
                        q = torch.roll(p, 1, 0)
                        assert_close(H(man.sqdist(p.cuda(), q.cuda(), c)), getattr(om, pre + "sqdist")(p, q, c),
                                     what="sqdist", exact=getattr(om, pre + "sqdist")(p.double(), q.double(), c))


@pytest.mark.parametrize("mname", ["PoincareBall", "Hyperboloid"])
@pytest.mark.parametrize("d", [16, 48, 64])
def test_hyp_act_narrow_rows_32_per_wavefront(mname, d):
    """hyp_act_rows32 (csrc/rowops.hip: rows of <= 64 columns, >= 4096 of them, 32 rows per wavefront and pass, the
    scalar chain once per row): against the oracle (layers/hyp_layers.py:170-173) on a row count off every boundary, and
    for the other activations against the generic row kernel (which short inputs still take)."""
    import hypdiff_amd.manifolds as hm
    from hypdiff_amd import ops
    man = getattr(hm, mname)()
    pre = "pb_" if mname == "PoincareBall" else "hb_"
    rs = np.random.RandomState(d)
    n = 4099
    u = (rs.standard_normal((n, d)) * rs.choice([0.05, 2.0], size=(n, 1)) / np.sqrt(d)).astype(np.float32)
    u[0] = 0
    u[n - 1] = 0
    for c in (0.6, 1.0):
        with torch.no_grad():
            p = getattr(om, pre + "proj")(getattr(om, pre + "expmap0")(torch.from_numpy(u), c), c)
        pg = p.cuda()
        got = ops.hyp_act(man.abi, pg, c, 1.0, 1)
        assert_close(H(got), olayers.hyp_act(mname, p, c, 1.0), what="hyp_act rows32")
        for act in (0, 1, 5, 2):                       # identity, relu, leaky_relu on the new kernel; tanh on the generic one
            full = ops.hyp_act(man.abi, pg, c, 0.8, act)
            parts = torch.cat([ops.hyp_act(man.abi, pg[i:i + 1000].contiguous(), c, 0.8, act) for i in range(0, n, 1000)])
            assert_close(H(full), H(parts), what="hyp_act rows32 vs row kernel, act %d" % act)


def test_empty_inputs():
    import hypdiff_amd.manifolds as hm
    for man in (hm.PoincareBall(), hm.Hyperboloid()):
        x = torch.zeros(0, 16, device="cuda")
        assert man.expmap0(x, 1.0).shape == (0, 16)
        assert man.logmap0(x, 1.0).shape == (0, 16)
        assert man.sqdist(x, x, 1.0).shape[0] == 0


def test_cpu_tensor_is_refused():
    import hypdiff_amd.manifolds as hm
    with pytest.raises(RuntimeError):
        hm.PoincareBall().expmap0(torch.zeros(2, 4), 1.0)


# ---------------------------------------------------------------------------------
GRAPHS = [("ba", 100), ("ego", 17), ("community", 12), ("citation", 300)]


def _adj(g, k0, n):
    idx, val = G(g[k0 + "coo_indices"]), G(g[k0 + "coo_values"])
    return torch.sparse_coo_tensor(idx, val, (n, n), check_invariants=False)


@pytest.mark.parametrize("kind,n", GRAPHS)
def test_csr_bit_exact(golden, kind, n):
    from hypdiff_amd import ops
    g = golden("layers")
    k0 = f"{kind}{n}/"
    csr = ops.csr_of(_adj(g, k0, n))
    rowptr, col, val = adjacency.coo_to_csr(g[k0 + "coo_indices"], g[k0 + "coo_values"], n)
    assert csr.rowptr.dtype == torch.int32 and csr.col.dtype == torch.int32
    assert np.array_equal(H(csr.rowptr), rowptr)
    assert np.array_equal(H(csr.col), col)
    assert np.array_equal(H(csr.val), val)


@pytest.mark.parametrize("kind,n", GRAPHS)
def test_device_process_bit_exact(golden, kind, n):
    """utils/data_utils.process on the device == the COO the reference's own process() produced from the
    same adj_train (indices, order and fp32 values bit for bit); identity features stay the identity."""
    import scipy.sparse as sp
    from hypdiff_amd.utils import data_utils
    g = golden("layers")
    k0 = f"{kind}{n}/"
    indptr, indices = g[k0 + "adj_train_indptr"], g[k0 + "adj_train_indices"]
    adj_train = sp.csr_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n))
    adj_t, feats = data_utils.process(adj_train, sp.eye(n).tocsr(), 1, 1)
    assert adj_t.is_sparse and adj_t.device.type == "cuda"
    assert np.array_equal(H(adj_t._indices()), g[k0 + "coo_indices"])
    assert np.array_equal(H(adj_t._values()), g[k0 + "coo_values"])
    assert np.array_equal(H(feats), np.eye(n, dtype=np.float32))
    # the same from device index arrays (no scipy on the way), isolated nodes and explicit self loops
    t2 = data_utils.normalize_adj((torch.from_numpy(indptr.astype(np.int64)), torch.from_numpy(indices.astype(np.int64))))
    assert np.array_equal(H(t2._values()), g[k0 + "coo_values"])
    rs = np.random.RandomState(n)
    a = sp.random(60, 60, density=0.05, random_state=rs, format="csr")
    a.data[:] = 1.0
    a = ((a + a.T) > 0).astype(np.float64).tolil()
    a[3, 3] = 1.0
    a[7, :] = 0
    a[:, 7] = 0
    a = a.tocsr()
    idx, val, shape = adjacency.process_adj(a)
    got = data_utils.normalize_adj(a)
    assert np.array_equal(H(got._indices()), idx) and np.array_equal(H(got._values()), val)
    w = sp.random(40, 40, density=0.1, random_state=rs, format="csr")       # weighted entries, fp64 row sums
    idx, val, shape = adjacency.process_adj(w)
    got = data_utils.normalize_adj(w)
    assert np.array_equal(H(got._indices()), idx)
    assert_close(H(got._values()), val, rtol=2e-7, what="weighted normalize")
    f = rs.rand(33, 70)
    f[5] = 0.0
    want = adjacency.row_normalize(sp.csr_matrix(f)).toarray().astype(np.float32)
    assert_close(H(data_utils.normalize_features(f)), want, rtol=2e-7, what="normalize_features")
    with pytest.raises(ValueError):
        data_utils.normalize_adj((torch.tensor([0, 2, 2]), torch.tensor([1, 0])))


def test_f16x2_edge_cases():
    """Empty operands, rows past the one-pass producers' limit, non-finite inputs and rows whose bound
    overflows fp32 (hd_gemm_f16x2_f32 / hd_row_pack_f16x2_f32 / hd_mlp_pack_input_f16x2)."""
    from hypdiff_amd import ops
    from hypdiff_amd.utils import data_utils
    rs = np.random.RandomState(11)
    w = G(rs.standard_normal((40, 36)).astype(np.float32))
    wl = ops.limb_pack(w, "f16x2")
    e = ops.limb_pack(torch.zeros(0, 36, device="cuda"), "f16x2")
    c, cl = ops.gemm_limbs(e, wl, want_f32=True, want_limbs=True, w_norm=1.0, bias_max=0.0)
    assert c.shape == (0, 40) and cl.rows == 0
    assert ops.row_pack(torch.zeros(0, 36, device="cuda"), 1, 1.0).rows == 0
    with pytest.raises(RuntimeError):
        ops._lib.call("hd_row_pack_f16x2_f32", None, 1, ops.curv(1.0), None, None, None, None, 4, 2000, ops._stream())
    with pytest.raises(RuntimeError):
        ops._lib.call("hd_gemm_f16x2_f32", None, None, None, None, None, None, None, None, None, None, 4, 4, 4, 4, 2,
                      0.0, 0.0, ops._stream())
    # a zero row, a row of 1e37 (its bound overflows fp32: exponent falls back to 2^-114), an inf and a NaN
    a = rs.standard_normal((6, 36)).astype(np.float32)
    a[1] = 0.0
    a[2] *= 1e37 / np.abs(a[2]).max()
    a[3, 5] = np.inf
    a[4, 7] = np.nan
    wn = float(w.double().norm(dim=1).max())
    c, cl = ops.gemm_limbs(ops.limb_pack(G(a), "f16x2"), wl, want_f32=True, want_limbs=True, w_norm=wn, bias_max=0.0)
    hc = H(c)
    ref = a.astype(np.float64) @ H(w).astype(np.float64).T
    mag = np.abs(a).astype(np.float64) @ np.abs(H(w)).astype(np.float64).T
    for r in (0, 1, 2, 5):
        assert np.all(np.abs(hc[r] - ref[r]) <= 2e-6 * mag[r] + 1e-30), r
    assert not np.isfinite(hc[3]).any() and np.isnan(hc[4]).all()
    back = H(ops.limb_unpack(cl))
    for r in (0, 1, 5):
        assert np.all(np.abs(back[r] - hc[r]) <= 2.0 ** -20 * np.abs(hc[r]).max() + 1e-30), r
    assert np.all(np.isfinite(back[2])) and np.all(np.abs(back[2] - hc[2]) <= 2.0 ** -9 * np.abs(hc[2]).max())
    # empty graph through the device preprocessing
    import scipy.sparse as sp
    t = data_utils.normalize_adj(sp.csr_matrix((3, 3)))
    assert np.array_equal(H(t._indices()), np.array([[0, 1, 2], [0, 1, 2]])) and np.array_equal(H(t._values()), np.ones(3, np.float32))


def test_hb_linear_narrow_and_tail_kernels():
    """hd_hb_linear_smallk_f32 (one-launch HypLinear for d_in <= 64; W^T in LDS or, when it does not fit, from L2)
    and hd_hb_linear_tail_rows_f32 against the composition they replace -- hd_hb_logmap0 -> exact fp32 GEMM ->
    hd_hb_linear_epilogue -- and against the oracle's mobius_matvec chain."""
    from hypdiff_amd import ops
    from oracle import manifolds as om
    rs = np.random.RandomState(21)
    for (n, K, N) in [(777, 21, 512), (300, 64, 512), (130, 5, 8), (1, 24, 64), (515, 32, 256)]:
        u = rs.standard_normal((n, K)).astype(np.float32) * 0.5
        u[:, 0] = 0.0
        if n > 3:
            u[3, 1:] = 0.0                                                         # a row at the origin
        x = ops.map1("hd_hb_expmap0_proj", G(u), 1.0)
        w = G((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        for flags in (0, 1, 3):
            got = torch.empty(n, N, device="cuda")
            ops._lib.call("hd_hb_linear_smallk_f32", ops._p(x), ops._p(w.t().contiguous()), ops._p(got), n, K, N,
                          ops.curv(1.0), flags, ops._stream())
            old, ops.GEMM_MODE = ops.GEMM_MODE, "exact"
            try:
                kp = (K + 3) // 4 * 4
                mu = ops.gemm(ops._pad_k(ops.map1("hd_hb_logmap0", x, 1.0), kp), ops._pad_k(w, kp))
            finally:
                ops.GEMM_MODE = old
            want = torch.empty_like(mu)
            ops._lib.call("hd_hb_linear_epilogue_f32", ops._p(mu), ops._p(want), n, N, ops.curv(1.0), flags, ops._stream())
            assert_close(H(got), H(want), what=f"smallk {(n, K, N, flags)}", atol_scale=2e-6)
            fast = torch.empty_like(mu)
            ops._lib.call("hd_hb_linear_tail_rows_f32", ops._p(mu), ops._p(fast), n, N, ops.curv(1.0), flags, ops._stream())
            assert_close(H(fast), H(want), what=f"tail_rows {(n, N, flags)}", atol_scale=2e-6)
        ref = om.hb_proj(om.hb_mobius_matvec(torch.from_numpy(H(w)), torch.from_numpy(H(x)), 1.0), 1.0) \
            if True else None
        if ref is not None:
            got = torch.empty(n, N, device="cuda")
            ops._lib.call("hd_hb_linear_smallk_f32", ops._p(x), ops._p(w.t().contiguous()), ops._p(got), n, K, N,
                          ops.curv(1.0), 1, ops._stream())
            assert_close(H(got), ref.numpy(), what=f"smallk vs oracle {(n, K, N)}", atol_scale=2e-6)


def test_hb_linear_narrow_on_matrix_cores():
    """hd_hb_linear_smallk_bf16x3_f32 (d_in <= 32: one MFMA k-step per output tile, 3 bf16 limbs) against the exact
    fp32 fma-chain kernel hd_hb_linear_smallk_f32 and against the oracle; c = 0.7 so the curvature terms show."""
    from hypdiff_amd import ops
    from oracle import manifolds as om
    rs = np.random.RandomState(22)
    for (n, K, N, c) in [(777, 21, 512, 1.0), (64, 24, 512, 0.7), (65, 32, 256, 0.7), (1, 1, 16, 1.0), (130, 5, 48, 2.0),
                         (1000, 17, 128, 1.0), (15777, 24, 512, 1.0)]:
        u = rs.standard_normal((n, K)).astype(np.float32) * 0.5 * np.exp(rs.standard_normal((n, 1))).astype(np.float32)
        u[:, 0] = 0.0
        if n > 3:
            u[3, 1:] = 0.0                                                         # a row at the origin
        x = ops.map1("hd_hb_expmap0_proj", G(u), c)
        w = G((rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        img = ops.smallk_image_cached(w)
        assert img.numel() * 4 == ops._lib.load().hd_smallk_image_bytes(N)
        for flags in (0, 1, 3):
            got = torch.full((n, N), float("nan"), device="cuda")
            ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", ops._p(x), ops._p(img), ops._p(got), n, K, N, ops.curv(c),
                          flags, ops._stream())
            want = torch.empty(n, N, device="cuda")
            ops._lib.call("hd_hb_linear_smallk_f32", ops._p(x), ops._p(w.t().contiguous()), ops._p(want), n, K, N,
                          ops.curv(c), flags, ops._stream())
            assert_close(H(got), H(want), what=f"smallk mfma {(n, K, N, c, flags)}", atol_scale=2e-6)
        if n <= 1000:
            ref = om.hb_proj(om.hb_mobius_matvec(torch.from_numpy(H(w)), torch.from_numpy(H(x)), c), c)
            got = torch.empty(n, N, device="cuda")
            ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", ops._p(x), ops._p(img), ops._p(got), n, K, N, ops.curv(c), 1,
                          ops._stream())
            assert_close(H(got), ref.numpy(), what=f"smallk mfma vs oracle {(n, K, N, c)}", atol_scale=2e-6)
    with pytest.raises(RuntimeError, match="outside the fused path"):
        ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", None, None, None, 4, 33, 512, ops.curv(1.0), 0, ops._stream())
    ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", None, None, None, 0, 21, 512, ops.curv(1.0), 0, ops._stream())


def test_csr_unsorted_empty_rows_and_bounds():
    from hypdiff_amd import ops
    rs = np.random.RandomState(3)
    n = 50
    row = rs.randint(5, 40, size=400)
    col = rs.randint(0, n, size=400)
    val = rs.standard_normal(400).astype(np.float32)
    csr = ops.csr_from_coo(G(np.vstack([row, col]).astype(np.int64)), G(val), (n, n))
    rp, cc, vv = adjacency.coo_to_csr(np.vstack([row, col]), val, n)
    assert np.array_equal(H(csr.rowptr), rp) and np.array_equal(H(csr.col), cc) and np.array_equal(H(csr.val), vv)
    empty = ops.csr_from_coo(torch.zeros(2, 0, dtype=torch.int64, device="cuda"), torch.zeros(0, device="cuda"), (7, 7))
    assert np.array_equal(H(empty.rowptr), np.zeros(8, dtype=np.int32))
    with pytest.raises(IndexError):
        ops.csr_from_coo(G(np.array([[0, 1], [0, 99]], dtype=np.int64)), G(np.ones(2, dtype=np.float32)), (5, 5))


def test_spmm_power_law_vs_oracle():
    from hypdiff_amd import ops
    n, d = 3000, 256
    adj = gi.synthetic_graph("ba", n)
    idx, val, shape = adjacency.process_adj(adj)
    rs = np.random.RandomState(0)
    x = rs.standard_normal((n, d)).astype(np.float32)
    csr = ops.csr_from_coo(G(idx), G(val), shape)
    got = ops.spmm(csr, G(x))
    want = olayers.spmm_coo(torch.from_numpy(idx), torch.from_numpy(val), shape, torch.from_numpy(x))
    assert_close(H(got), want, what="spmm")
    # linearity: A(ax + by) == a Ax + b Ay
    y = rs.standard_normal((n, d)).astype(np.float32)
    lhs = ops.spmm(csr, G(2 * x - 3 * y))
    rhs = 2 * got - 3 * ops.spmm(csr, G(y))
    assert_close(H(lhs), H(rhs), rtol=1e-4, atol_scale=1e-5, what="spmm linearity")


@pytest.mark.parametrize("kind,n", GRAPHS)
@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_layers_golden(golden, kind, n, mname, c):
    import hypdiff_amd.manifolds as hm
    from hypdiff_amd.layers import hyp_layers as L
    import torch.nn.functional as F
    g = golden("layers")
    k0 = f"{kind}{n}/"
    adj = _adj(g, k0, n)
    man = getattr(hm, mname)()
    for bias in (0, 1):
        k = k0 + f"{mname}/bias{bias}/"
        if k + "encode" not in g.files:
            continue
        dim = g[k + "W0"].shape[0]
        feat = g[k + "W0"].shape[1]
        ct = torch.tensor([c]).cuda()
        layers = []
        for i, (fi, fo) in enumerate([(feat, dim), (dim, dim)]):
            lyr = L.HyperbolicGraphConvolution(man, fi, fo, ct, ct, 0.0, F.relu, bias, 0, 0).cuda().eval()
            lyr.linear.load_state_dict({"weight": torch.from_numpy(g[k + f"W{i}"]), "bias": torch.from_numpy(g[k + f"b{i}"])})
            layers.append(lyr)
        with torch.no_grad():
            h = G(g[k + "lift"])
            for i, lyr in enumerate(layers):
                hin = G(g[k + f"L{i-1}/act"]) if i else h
                assert_close(H(lyr.linear(hin)), g[k + f"L{i}/linear"], what=k + f"L{i}/linear")
                assert_close(H(lyr.agg(G(g[k + f"L{i}/linear"]), adj)), g[k + f"L{i}/agg"], what=k + f"L{i}/agg")
                assert_close(H(lyr.hyp_act(G(g[k + f"L{i}/agg"]))), g[k + f"L{i}/act"], what=k + f"L{i}/act")
                fused, _ = lyr((hin, adj))
                assert_close(H(fused), g[k + f"L{i}/act"], what=k + f"L{i}/fused")      # plain bar: uses <= 0.18 of it (profiles/r03_parity_margin.json)


@BOTH_GEMMS
@pytest.mark.parametrize("kind,n", GRAPHS)
@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_hgcn_encode_golden(golden, kind, n, mname, c, gemm_mode):
    import types
    from hypdiff_amd.models.encoders import HGCN
    g = golden("layers")
    k0 = f"{kind}{n}/"
    k = k0 + f"{mname}/bias0/"
    dim = g[k + "W0"].shape[0]
    feat = g[k + "W0"].shape[1]
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=feat, dim=dim, num_layers=2,
                                 act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    enc = HGCN(torch.tensor([c]).cuda(), args).cuda().eval()
    for i, lyr in enumerate(enc.layers):
        lyr.linear.load_state_dict({"weight": torch.from_numpy(g[k + f"W{i}"]), "bias": torch.from_numpy(g[k + f"b{i}"])})
    x = G(g[k0 + "features"])
    if mname == "Hyperboloid":
        x = torch.cat([torch.zeros_like(x[:, :1]), x], 1)       # models/base_models.py:79-81
    with torch.no_grad():
        out = enc.encode(x, _adj(g, k0, n))
    assert_close(H(out), g[k + "encode"], what=k + "encode")                 # plain bar: uses <= 0.26 of it
    # LPModel: encode through the wrapper (zero time column prepended for the hyperboloid) + edge decoder
    from hypdiff_amd.models.base_models import LPModel
    largs = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=n, n_nodes=n, model="HGCN",
                                  r=2.0, t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, num_layers=2, act="relu",
                                  dropout=0.0, bias=0, use_att=0, local_agg=0)
    lp = LPModel(largs).cuda().eval()
    for i, lyr in enumerate(lp.encoder.layers):
        lyr.linear.load_state_dict({"weight": torch.from_numpy(g[k + f"W{i}"]), "bias": torch.from_numpy(g[k + f"b{i}"])})
    with torch.no_grad():
        emb, _, _, _ = lp.encode(G(g[k0 + "features"]), _adj(g, k0, n))
        assert_close(H(emb), g[k + "encode"], what=k + "lp encode")
        probs = lp.decode(G(g[k + "encode"]), G(g[k0 + "train_edges"]))
    ex = olayers.lp_decode(mname, torch.from_numpy(g[k + "encode"]).double(), torch.from_numpy(g[k0 + "train_edges"]), c, 2.0, 1.0)
    assert_close(H(probs), g[k + "decode_train"], what=k + "decode", exact=ex.numpy())


# ---------------------------------------------------------------------------------
@BOTH_GEMMS
def test_gemm_vs_fp64(gemm_mode):
    from hypdiff_amd import ops
    rs = np.random.RandomState(1)
    for (M, N, K) in [(1, 5, 8), (129, 700, 900), (300, 64, 516), (1000, 800, 65), (257, 130, 36)]:
        a = rs.standard_normal((M, K)).astype(np.float32)
        w = rs.standard_normal((N, K)).astype(np.float32)
        # asymmetric, exactly representable check first (catches transposed fragment maps)
        ai = rs.randint(-4, 5, size=(M, K)).astype(np.float32)
        wi = rs.randint(-4, 5, size=(N, K)).astype(np.float32)
        assert np.array_equal(H(ops.gemm(G(ai), G(wi))), ai @ wi.T)
        b = rs.standard_normal(N).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T + b
        mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
        got = H(ops.gemm(G(a), G(w), G(b)))
        assert np.all(np.abs(got - ref) <= 2e-6 * mag + 1e-6), (M, N, K, np.abs(got - ref).max())
        gl = H(ops.gemm(G(a), G(w), G(b), epi=1))
        want = torch.nn.functional.gelu(torch.from_numpy(ref)).numpy()
        assert np.all(np.abs(gl - want) <= 2e-6 * mag + 1e-6)
        a64, w64 = a.astype(np.float64), w.astype(np.float64)
        assert_close(H(ops.gemm(G(a64), G(w64))), a64 @ w64.T, rtol=1e-12, atol_scale=1e-13, what="gemm f64")


def _diff(dim=16, T=200):
    import types
    from hypdiff_amd.models.direction_diffusion import diffusion
    d = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T).eval()
    sd = {}
    for i, (w, b) in enumerate(gi.mlp_params(dim), 1):
        sd[f"fc{i}.weight"] = torch.from_numpy(w)
        sd[f"fc{i}.bias"] = torch.from_numpy(b)
    d.mlp.load_state_dict(sd)
    return d


def test_schedule_tables_golden(golden):
    g = golden("diffusion")
    for T in (200, 1000):
        d = _diff(16, T)
        for ours, theirs in (("alphas_cumprod", "alphas_cumprod"), ("sqrt_alphas_cumprod", "sqrt_alphas_cumprod"),
                             ("minus_alphas_cumprod", "minus_alphas_cumprod"),
                             ("sqrt_recip_alphas_cumprod", "sqrt_recip_alphas_cumprod"),
                             ("sqrt_recipm1_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")):
            # host-side torch ops: last-bit differences between host CPUs are allowed
            assert_close(H(getattr(d, ours)), g[f"T{T}/{theirs}"], rtol=1e-14, atol_scale=1e-15, what=ours)


def test_q_sample_golden(golden):
    g = golden("diffusion")
    d = _diff()
    u0 = G(g["c/u0"])
    labels = g["c/labels"]
    t = G(g["c/t"])
    with torch.no_grad():
        direction = d.directions(u0)
        assert np.array_equal(np.sign(H(torch.stack(direction))), np.sign(g["c/direction"]))
        for tag, xdt, ndt in (("f64f64", torch.float64, torch.float64), ("f32f64", torch.float32, torch.float64),
                              ("f32f32", torch.float32, torch.float32)):
            x = G(g["c/x"], xdt)
            for r in (0, 1):
                nz = G(g["c/noise"], ndt)
                out = d.q_sample(x, t, direction, nz, labels, u0, restrict=bool(r))
                assert out.dtype == torch.float64
                assert_close(H(out), g[f"q_sample/{tag}/restrict{r}/out"], rtol=1e-6, atol_scale=1e-7,
                             what=f"q_sample {tag} {r}")
                assert np.array_equal(H(nz), g[f"q_sample/{tag}/restrict{r}/noise_after"])


def test_limb_images_and_limb_gemm():
    """bf16 limb images (csrc/gemm_limbs.hip): exact round trip, fp32-accurate contraction from
    pre-split operands, limb write-out equal to the fp32 write-out, residual epilogue, chains."""
    from hypdiff_amd import ops
    rs = np.random.RandomState(5)
    x = rs.standard_normal((300, 70)).astype(np.float32) * np.exp(rs.standard_normal((300, 1)) * 8).astype(np.float32)
    x[7] = 0.0
    assert np.array_equal(H(ops.limb_unpack(ops.limb_pack(G(x)))), x)            # h + m + l is exact
    xs = G(np.ascontiguousarray(np.pad(x, ((0, 0), (0, 6)))))[:, :70]             # strided rows
    assert np.array_equal(H(ops.limb_unpack(ops.limb_pack(xs))), x)
    for (M, N, K) in [(1, 5, 8), (129, 700, 900), (300, 64, 516), (1000, 800, 65), (257, 130, 36), (128, 64, 32)]:
        ai = rs.randint(-4, 5, size=(M, K)).astype(np.float32)
        wi = rs.randint(-4, 5, size=(N, K)).astype(np.float32)
        got, _ = ops.gemm_limbs(ops.limb_pack(G(ai)), ops.limb_pack(G(wi)))
        assert np.array_equal(H(got), ai @ wi.T)
        # rows spanning 14 orders of magnitude: the error bar is relative to sum |a||w| per element
        a = rs.standard_normal((M, K)).astype(np.float32) * np.exp(rs.standard_normal((M, 1)) * 8).astype(np.float32)
        w = rs.standard_normal((N, K)).astype(np.float32)
        b = rs.standard_normal(N).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T + b
        mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
        al, wl = ops.limb_pack(G(a)), ops.limb_pack(G(w))
        c, cl = ops.gemm_limbs(al, wl, G(b), want_f32=True, want_limbs=True)
        assert np.all(np.abs(H(c) - ref) <= 2e-6 * mag), (M, N, K, (np.abs(H(c) - ref) / mag).max())
        assert np.array_equal(H(ops.limb_unpack(cl)), H(c))                       # same values, two formats
        gl, _ = ops.gemm_limbs(al, wl, G(b), epi=1)
        want = torch.nn.functional.gelu(torch.from_numpy(ref)).numpy()
        assert np.all(np.abs(H(gl) - want) <= 2e-6 * mag + 1e-30)
        if True:
            r = rs.standard_normal((M, N)).astype(np.float32)
            gr, _ = ops.gemm_limbs(al, wl, G(b), epi=2, resid=G(r))
            assert np.all(np.abs(H(gr) - (r + want)) <= 2e-6 * mag + 1e-6 * np.abs(r) + 1e-30)
    # a chain (first layer gelu, residual layers, plain last layer; ddpm.py:17-27) in both modes
    ws = [torch.from_numpy((rs.standard_normal(s) / np.sqrt(s[1])).astype(np.float32)).cuda()
          for s in [(96, 68), (96, 96), (96, 96), (10, 96)]]
    bs = [torch.from_numpy(rs.standard_normal(w.shape[0]).astype(np.float32)).cuda() for w in ws]
    x0 = G(rs.standard_normal((333, 68)).astype(np.float32))
    layers = [(w, b, e) for w, b, e in zip(ws, bs, (1, 2, 2, 0))]
    h = x0.double()
    for w, b, e in layers:
        z = h @ w.double().T + b.double()
        h = z if e == 0 else (torch.nn.functional.gelu(z) if e == 1 else h + torch.nn.functional.gelu(z))
    old = ops.GEMM_MODE
    try:
        for mode in ("bf16x3", "f16x2", "exact"):
            ops.GEMM_MODE = mode
            assert_close(H(ops.linear_chain(x0, layers)), H(h), what="linear_chain " + mode, atol_scale=2e-6)
    finally:
        ops.GEMM_MODE = old


def test_f16x2_limb_images_and_gemm():
    """f16x2 limb images (csrc/gemm_limbs.hip): two row-scaled fp16 limbs, three MFMAs per product.
    Round trip to 2^-22 of the row maximum, integers exact, fp32-accurate contraction over rows
    spanning 14 orders of magnitude, exact row maxima out of the epilogue, row exponents from the
    Cauchy-Schwarz bound never overflow fp16, limb write-out equal to the fp32 write-out to 2^-22 of
    the row maximum, subnormal limbs kept by the matrix cores, and a gelu chain."""
    from hypdiff_amd import ops
    rs = np.random.RandomState(6)
    x = rs.standard_normal((300, 70)).astype(np.float32) * np.exp(rs.standard_normal((300, 1)) * 8).astype(np.float32)
    x[7] = 0.0
    x[9, 3] = 1e-30
    img = ops.limb_pack(G(x), "f16x2")
    rowmax = np.abs(x).max(axis=1, keepdims=True)
    assert np.all(np.abs(H(ops.limb_unpack(img)) - x) <= 2.0 ** -22 * rowmax)
    assert np.array_equal(H(img.amax.view(torch.float32)), rowmax[:, 0])
    xi = rs.randint(-2000, 2001, size=(130, 40)).astype(np.float32)               # 11-bit integers: one limb
    assert np.array_equal(H(ops.limb_unpack(ops.limb_pack(G(xi), "f16x2"))), xi)
    # fp16 subnormals take part in the product (a flushing matrix core would lose all of it)
    a = np.full((128, 64), 2.0 ** -30, np.float32) * (1 + rs.rand(128, 64).astype(np.float32))
    a[:, 0] = 1.5
    w = rs.standard_normal((64, 64)).astype(np.float32)
    w[:, 0] = 0.0
    got, _ = ops.gemm_limbs(ops.limb_pack(G(a), "f16x2"), ops.limb_pack(G(w), "f16x2"))
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    assert np.all(np.abs(H(got) - ref) <= 2.0 ** -8 * mag)
    for (M, N, K) in [(1, 5, 8), (129, 700, 900), (300, 64, 516), (1000, 800, 65), (257, 130, 36), (128, 64, 32),
                      (385, 896, 64)]:
        ai = rs.randint(-4, 5, size=(M, K)).astype(np.float32)
        wi = rs.randint(-4, 5, size=(N, K)).astype(np.float32)
        got, _ = ops.gemm_limbs(ops.limb_pack(G(ai), "f16x2"), ops.limb_pack(G(wi), "f16x2"))
        assert np.array_equal(H(got), ai @ wi.T)
        a = rs.standard_normal((M, K)).astype(np.float32) * np.exp(rs.standard_normal((M, 1)) * 8).astype(np.float32)
        w = rs.standard_normal((N, K)).astype(np.float32) * np.exp(rs.standard_normal((N, 1)) * 3).astype(np.float32)
        b = rs.standard_normal(N).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64).T + b
        mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
        al, wl = ops.limb_pack(G(a), "f16x2"), ops.limb_pack(G(w), "f16x2")
        c, cl = ops.gemm_limbs(al, wl, G(b), want_f32=True, want_limbs=True)
        hc = H(c)
        assert np.all(np.abs(hc - ref) <= 2e-6 * mag), (M, N, K, (np.abs(hc - ref) / mag).max())
        cmax = np.abs(hc).max(axis=1)
        assert np.array_equal(H(cl.amax.view(torch.float32)), cmax)               # exact, order-independent
        scaled = cmax.astype(np.float64) * 2.0 ** H(cl.exp).astype(np.float64)
        assert np.all(scaled < 2.0 ** 14), scaled.max()                           # the bound is a bound
        # two 11-bit limbs; below 2^-14 (scaled) fp16 is a fixed grid of 2^-24
        grid = 2.0 ** -24 * 2.0 ** -H(cl.exp).astype(np.float64)
        assert np.all(np.abs(H(ops.limb_unpack(cl)).astype(np.float64) - hc) <= 2.0 ** -22 * np.abs(hc) + grid[:, None])
        gl, gll = ops.gemm_limbs(al, wl, G(b), epi=1, want_f32=True, want_limbs=True)
        want = torch.nn.functional.gelu(torch.from_numpy(ref)).numpy()
        assert np.all(np.abs(H(gl) - want) <= 2e-6 * mag + 1e-30)
        assert np.array_equal(H(gll.amax.view(torch.float32)), np.abs(H(gl)).max(axis=1))
    # fused denoiser-input pack (cat(x, t) -> image) == pack_input followed by limb_pack, bit for bit
    for d, dt in [(512, np.float64), (511, np.float64), (14, np.float32), (700, np.float32), (1535, np.float64)]:
        n = 261
        xx = (rs.standard_normal((n, d)) * np.exp(rs.standard_normal((n, 1)) * 3)).astype(dt)
        for kind, tv in [(0, torch.from_numpy(rs.randint(0, 1000, size=n)).cuda()),
                         (1, torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()), (2, 37.0)]:
            tp, ts = (tv, 0.0) if kind != 2 else (None, tv)
            ld = (d + 1 + 3) // 4 * 4
            hbuf = torch.empty(n, ld, dtype=torch.float32, device="cuda")
            xg = G(xx)
            ops._lib.call("hd_mlp_pack_input", ops._p(xg), int(dt == np.float64), ops._p(tp), kind, ts, ops._p(hbuf),
                          n, d, ld, ops._stream())
            want = ops.limb_pack(hbuf[:, : d + 1], "f16x2")
            img = ops.limb_empty(n, d + 1, "cuda", "f16x2", zero_amax=False)
            ops._lib.call("hd_mlp_pack_input_f16x2", ops._p(xg), int(dt == np.float64), ops._p(tp), kind, ts,
                          ops._p(img.buf), ops._p(img.exp), ops._p(img.amax), n, d, ops._stream())
            assert torch.equal(img.exp, want.exp) and torch.equal(img.amax, want.amax), (d, dt, kind)
            assert np.array_equal(H(ops.limb_unpack(img)), H(ops.limb_unpack(want))), (d, dt, kind)
    ws = [torch.from_numpy((rs.standard_normal(s) / np.sqrt(s[1])).astype(np.float32)).cuda()
          for s in [(96, 68), (200, 96), (96, 200), (10, 96)]]
    bs = [torch.from_numpy(rs.standard_normal(w.shape[0]).astype(np.float32)).cuda() for w in ws]
    x0 = G(rs.standard_normal((333, 68)).astype(np.float32) * np.exp(rs.standard_normal((333, 1)) * 4).astype(np.float32))
    layers = [(w, b, e) for w, b, e in zip(ws, bs, (1, 1, 1, 0))]
    h = x0.double()
    for w, b, e in layers:
        z = h @ w.double().T + b.double()
        h = z if e == 0 else torch.nn.functional.gelu(z)
    old = ops.GEMM_MODE
    try:
        ops.GEMM_MODE = "f16x2"
        assert_close(H(ops.linear_chain(x0, layers)), H(h), what="linear_chain f16x2 (gelu)", atol_scale=2e-6)
    finally:
        ops.GEMM_MODE = old


def test_bf16x3_row_producers():
    """hd_mlp_pack_input_bf16x3 == hd_mlp_pack_input followed by hd_limb_pack_f32, and hd_row_pack_bf16x3_f32 ==
    the row kernel (logmap0 / identity + row norm) followed by hd_limb_pack_f32 -- both bit for bit (the whole
    real rows out to the K padding of the last 32-column slab, which the GEMM multiplies and so must be zero)."""
    from hypdiff_amd import ops
    rs = np.random.RandomState(77)

    def wide(im):
        return ops.LimbImage(im.buf, im.rows, (im.cols + 31) // 32 * 32)
    for d, dt in [(512, np.float64), (511, np.float64), (14, np.float32), (700, np.float32), (1023, np.float64)]:
        n = 261
        xx = (rs.standard_normal((n, d)) * np.exp(rs.standard_normal((n, 1)) * 3)).astype(dt)
        for kind, tv in [(0, torch.from_numpy(rs.randint(0, 1000, size=n)).cuda()),
                         (1, torch.from_numpy(rs.rand(n).astype(np.float32)).cuda()), (2, 37.0)]:
            tp, ts = (tv, 0.0) if kind != 2 else (None, tv)
            ld = (d + 1 + 3) // 4 * 4
            hbuf = torch.empty(n, ld, dtype=torch.float32, device="cuda")
            xg = G(xx)
            ops._lib.call("hd_mlp_pack_input", ops._p(xg), int(dt == np.float64), ops._p(tp), kind, ts, ops._p(hbuf),
                          n, d, ld, ops._stream())
            want = ops.limb_pack(hbuf[:, : d + 1], "bf16x3")
            img = ops.limb_empty(n, d + 1, "cuda", "bf16x3")
            img.buf.fill_(0x55)                  # every byte of the image must be written by the producer
            ops._lib.call("hd_mlp_pack_input_bf16x3", ops._p(xg), int(dt == np.float64), ops._p(tp), kind, ts,
                          ops._p(img.buf), n, d, ops._stream())
            assert np.array_equal(H(ops.limb_unpack(wide(img))), H(ops.limb_unpack(wide(want)))), (d, dt, kind)
    for n, d in [(261, 512), (16, 128), (1000, 130), (77, 1024), (129, 36)]:
        x = G((rs.standard_normal((n, d)) * 0.3).astype(np.float32))
        x[:, 0] = torch.sqrt(1.0 / 0.7 + (x[:, 1:] ** 2).sum(1))        # on the hyperboloid of K = 1 / 0.7
        img = ops.row_pack(x, 1, 0.7, "bf16x3")
        want = ops.limb_pack(ops.map1("hd_hb_logmap0", x, 0.7), "bf16x3")
        # (64 lanes per row here, 4 / 16 in the generic row kernel: the row reduction rounds differently)
        got_rows, want_rows = H(ops.limb_unpack(wide(img))), H(ops.limb_unpack(wide(want)))
        assert np.array_equal(got_rows[:, d:], want_rows[:, d:]) and not got_rows[:, d:].any()
        assert_close(got_rows, want_rows, what="logmap0 rows as bf16x3 limbs", rtol=4e-7, atol_scale=0)
        img, norms = ops.row_pack(x, 2, fmt="bf16x3")
        assert np.array_equal(H(ops.limb_unpack(wide(img))), H(ops.limb_unpack(wide(ops.limb_pack(x, "bf16x3"))))), (n, d)
        assert_close(H(norms), np.linalg.norm(H(x).astype(np.float64), axis=1), what="row norms", atol_scale=1e-6)
    with pytest.raises(RuntimeError, match="dim out of range"):
        ops._lib.call("hd_row_pack_bf16x3_f32", None, 1, ops.curv(1.0), None, None, 4, 2000, ops._stream())
    assert ops.row_pack(torch.zeros(0, 36, device="cuda"), 1, 1.0, "bf16x3").rows == 0


@BOTH_GEMMS
def test_mlp_and_p_sample_golden(golden, gemm_mode):
    g = golden("diffusion")
    d = _diff()
    u0 = G(g["c/u0"])
    labels = g["c/labels"]
    t = G(g["c/t"])
    params = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in gi.mlp_params(16)]
    with torch.no_grad():
        direction = d.directions(u0)
        x = G(g["c/x"], torch.float64)
        assert_close(H(d.mlp(x, t)), g["mlp/int_t"], what="mlp int t")
        assert_close(H(d.mlp(x, t / 200)), g["mlp/frac_t"], what="mlp frac t")
        for T, pre in ((200, ""), (1000, "T1000/")):
            dd = d if T == 200 else _diff(16, 1000)
            for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
                for step in ((150, 1, 0) if T == 200 else (999, 500, 0)):
                    key = f"{pre}p_sample/{tag}/t{step}/" if T == 200 else f"{pre}p_sample/t{step}/"
                    if key + "pred" not in g.files:
                        continue
                    img = torch.from_numpy(g["c/img"]).to(xdt)
                    raw = float(odiff.mlp_denoiser(img, torch.full((24,), step), params).abs().max())
                    pred, x0 = dd.p_sample(img.cuda(), None, 1, direction, labels, u0, step, None, True)
                    assert pred.dtype == torch.float64 and x0.dtype == torch.float32
                    # five chained fp32 GEMMs (K up to 900) on integer-t-sized activations: the
                    # summation order differs from the CPU sgemm, 3e-6 of the pre-clamp magnitude
                    assert_close(H(x0), g[key + "x_start"], what=key + "x0", scale=raw, atol_scale=3e-6)
                    assert_close(H(pred), g[key + "pred"], what=key + "pred", atol_scale=3e-6,
                                 scale=max(raw, float(np.abs(g[key + "pred"]).max())))


def ops_labels(labels):
    from hypdiff_amd import ops
    return ops.labels_dev(labels, "cuda")


def test_forward_loss_and_loop_golden(golden):
    g = golden("diffusion")
    d = _diff()
    u0 = G(g["c/u0"])
    labels = g["c/labels"]
    t = G(g["c/t"])
    with torch.no_grad():
        for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
            x = G(g["c/x"], xdt)
            loss = d(x, 1, labels, u0, restrict=True, graphset=True, t=t, noise=G(g["c/noise"], torch.float64))
            assert_close(H(loss), g[f"forward/{tag}/loss"], what="loss " + tag)
            assert_close(H(x), g[f"forward/{tag}/x_after"], what="x mutated " + tag)     # in-place side effect
        x = G(g["c/x"], torch.float32)
        ret = d.p_sample_loop(x, 1, labels, u0, graphset=True, img=G(g["c/img"]))
        assert_close(H(x), g["p_sample_loop/f32/x_after"], what="loop x mutated")
        # the end point against the reference's: 1e-5 + 3 D, D = the drift fp32 arithmetic itself causes over the 200
        # steps (oracle loop with the denoiser in fp32 vs in fp64, tests/loop_drift.py; D = 3e-6) -- see
        # test_loop_drift_stays_within_fp32s_own for the step-by-step picture
        from loop_drift import oracle_loop_states
        params = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in gi.mlp_params(16)]
        args = (torch.from_numpy(g["c/x"]).float(), labels, torch.from_numpy(g["c/u0"]), torch.from_numpy(g["c/img"]),
                odiff.schedule(200), params)
        _, e32 = oracle_loop_states(*args, exact=False)
        _, e64 = oracle_loop_states(*args, exact=True)
        D = float((e32 - e64).abs().max() / e64.abs().max())
        assert D < 1e-5, D
        assert_close(H(ret), g["p_sample_loop/f32/ret"], rtol=1e-5, atol_scale=1e-6 + 3 * D, what="loop end point (1e-5 + 3 D)")


@BOTH_GEMMS
def test_loop_drift_stays_within_fp32s_own(golden, gemm_mode):
    """p_sample_loop step by step (T = 200): after every step the distance of the HIP state from the exact trajectory
    (oracle loop with the denoiser in fp64) next to the distance of the oracle's fp32 loop from it.  fp32 rounding is
    carried from step to step but does not blow up (it peaks at ~3e-6 around step 10-50 and shrinks as the schedule
    contracts).  The HIP path is another fp32 evaluation with its own summation order: measured on MI355X it peaks at
    7.3e-6 with the bit-exact fp32-MFMA GEMMs (`exact`: a sequential fmaf chain over K <= 900, where the CPU's blocked
    sgemm sums pairwise-like), 8.9e-6 under f16x2 and 1.17e-5 under bf16x3 -- 2.3x / 2.9x / 3.8x the oracle's own worst
    -- and ends at 1.0e-6 / 3.3e-6 / 3.9e-6 (oracle 5e-7).  Asserted: at EVERY step within 5x the oracle's worst drift
    and within 2e-5 of the exact trajectory -- a measured model of the growth instead of a loose end-point tolerance."""
    from loop_drift import drift, oracle_loop_states
    g = golden("diffusion")
    d = _diff()
    u0 = G(g["c/u0"])
    labels = g["c/labels"]
    params = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in gi.mlp_params(16)]
    args = (torch.from_numpy(g["c/x"]).float(), labels, torch.from_numpy(g["c/u0"]), torch.from_numpy(g["c/img"]),
            odiff.schedule(200), params)
    s32, _ = oracle_loop_states(*args, exact=False)
    s64, _ = oracle_loop_states(*args, exact=True)
    with torch.no_grad():                               # the product's loop, :293-332, one step at a time
        x = G(g["c/x"], torch.float32)
        direction = d.directions(u0)
        lab = ops_labels(labels)
        img = d.tran_direction(direction, G(g["c/img"]).clone(), lab)
        h = d._logmap_centres(x, lab, u0)
        img = img + d._apply_moments(h, d.moments(h), 0)
        states = [img.cpu()]
        for t in reversed(range(200)):
            img, _ = d.p_sample(img, None, 1, direction, lab, u0, t, None, True)
            states.append(img.cpu())
    own = drift(s32, s64)
    hip = drift(states, s64)
    worst = own.max()
    print(f"\n[loop drift, GEMM_MODE={gemm_mode}] oracle fp32 vs exact: max {worst:.2e} at step {own.argmax()}, end {own[-1]:.2e}; "
          f"HIP vs exact: max {hip.max():.2e} at step {hip.argmax()}, end {hip[-1]:.2e}; largest HIP / oracle-worst {hip.max() / worst:.2f}")
    from conftest import record_margin
    record_margin("loop drift", max_norm_rel=float(hip.max()), plain_bar_used=float(hip.max() / 1e-5), bound_used=float(hip.max() / min(5 * worst, 2e-5)),
                  oracle_fp32_drift_max=float(worst), hip_drift_end=float(hip[-1]), oracle_fp32_drift_end=float(own[-1]), steps=200)
    assert worst < 1e-5
    assert (hip <= min(5 * worst, 2e-5)).all(), (hip.max(), worst)


def test_full_size_properties():
    """Config-2-sized batch (sum N = 15 777, D = 512): properties that need no oracle run."""
    import hypdiff_amd.manifolds as hm
    n, dim = 15777, 512
    gen = torch.Generator(device="cuda").manual_seed(0)
    u = torch.randn(n, dim, device="cuda", generator=gen) * 0.05
    for man in (hm.PoincareBall(), hm.Hyperboloid()):
        with torch.no_grad():
            uu = man.proj_tan0(u, 1.0) if man.name == "Hyperboloid" else u
            p = man.proj(man.expmap0(uu, 1.0), 1.0)
            back = man.logmap0(p, 1.0)
            assert_close(H(back), H(uu), rtol=1e-4, atol_scale=1e-5, what=man.name + " log0(exp0(u)) == u")
            assert_close(H(man.proj(p, 1.0)), H(p), rtol=1e-6, atol_scale=1e-7, what="proj idempotent")
            sd = man.sqdist(p, p, 1.0)
            assert float(sd.abs().max()) < 1e-5


@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_row_sharded_encode_equals_full(mname, c):
    """config 5 path: rows of one graph split over ranks, tangent all-gather per layer.  world = 1
    through dist.sharded_hgcn_encode; world = 2 and 3 simulated in-process (the all-gather is a
    concatenation of the padded blocks -- the collective itself is covered by the gloo test).
    Same per-row summation order as the 1-GPU kernel -> bit-identical."""
    import types
    from hypdiff_amd import dist as hdist
    from hypdiff_amd.models.encoders import HGCN, _lift
    n, feat, dim = 203, 12, 32
    idx, val, shape = adjacency.process_adj(gi.synthetic_graph("ba", n))
    rs = np.random.RandomState(5)
    x = rs.standard_normal((n, feat)).astype(np.float32) * 0.3
    if mname == "Hyperboloid":
        x = np.concatenate([np.zeros((n, 1), np.float32), x], 1)
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=x.shape[1], dim=dim,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    torch.manual_seed(3)
    enc = HGCN(torch.tensor([c]).cuda(), args).cuda().eval()
    adj = torch.sparse_coo_tensor(G(idx), G(val), shape, check_invariants=False)
    with torch.no_grad():
        full = enc.encode(G(x), adj)
        one = hdist.sharded_hgcn_encode(enc, G(x), hdist.RowShardedAdj(idx, val, shape, 0, 1))
        assert torch.equal(one, full)
        for world in (2, 3):
            shards = [hdist.RowShardedAdj(idx, val, shape, r, world) for r in range(world)]
            hs = [_lift(enc.manifold, G(x[s.lo:s.hi]), enc.curvatures[0]) for s in shards]
            for layer in enc.layers:
                xts = [layer.linear.forward(h, _tangent=True) for h in hs]
                gathered = torch.cat([torch.cat([xt, xt.new_zeros(s.rows_per_rank - xt.shape[0], xt.shape[1])])
                                      for s, xt in zip(shards, xts)])
                hs = [hdist.shard_aggregate(layer, hdist.shard_csr(s, "cuda"), gathered, s.hi - s.lo) for s in shards]
            assert torch.equal(torch.cat(hs), full), world


@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_row_sharded_encode_halo_form_equals_full(mname, c):
    """The halo form of the row-sharded encode (dist.RowShardedAdj._build_halo: the block re-indexed to [own rows |
    referenced remote rows], only those rows exchanged) on a graph with locality (a ring with short chords, < 50 % of the
    remote rows referenced): world = 2 and 3 simulated in-process -- the exchange is an index_select of the rows the
    lists name; the all_to_all_single itself is covered by the gloo test.  A row's entries keep their order, so the
    result is bit-identical to the 1-GPU encode."""
    import types
    from hypdiff_amd import dist as hdist
    from hypdiff_amd.models.encoders import HGCN, _lift
    n, feat, dim = 4500, 12, 32
    idx, val, shape = adjacency.process_adj(gi.synthetic_graph("ring", n))
    rs = np.random.RandomState(5)
    x = rs.standard_normal((n, feat)).astype(np.float32) * 0.3
    if mname == "Hyperboloid":
        x = np.concatenate([np.zeros((n, 1), np.float32), x], 1)
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=x.shape[1], dim=dim,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    torch.manual_seed(3)
    enc = HGCN(torch.tensor([c]).cuda(), args).cuda().eval()
    adj = torch.sparse_coo_tensor(G(idx), G(val), shape, check_invariants=False)
    with torch.no_grad():
        full = enc.encode(G(x), adj)
        for world in (2, 3):
            shards = [hdist.RowShardedAdj(idx, val, shape, r, world) for r in range(world)]
            assert all(s.use_halo and 0 < s.halo_fraction < 0.5 for s in shards)
            # what rank r sends to rank q is what rank q expects from rank r
            for q, sq in enumerate(shards):
                for r, sr in enumerate(shards):
                    assert np.array_equal(sr.send_rows[q] + sr.lo, sq.halo_ids[r])
            hs = [_lift(enc.manifold, G(x[s.lo:s.hi]), enc.curvatures[0]) for s in shards]
            for layer in enc.layers:
                xts = [layer.linear.forward(h, _tangent=True) for h in hs]
                gathered = torch.cat([torch.cat([xt, xt.new_zeros(s.rows_per_rank - xt.shape[0], xt.shape[1])])
                                      for s, xt in zip(shards, xts)])
                bufs = [torch.cat([gathered[s.rank * s.rows_per_rank:(s.rank + 1) * s.rows_per_rank],
                                   gathered[torch.from_numpy(np.concatenate(s.halo_ids)).cuda()]]) for s in shards]
                hs = [hdist.shard_aggregate(layer, hdist.shard_csr(s, "cuda", halo=True), b, s.hi - s.lo)
                      for s, b in zip(shards, bufs)]
            assert torch.equal(torch.cat(hs), full), world
