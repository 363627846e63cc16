"""CPU: the oracle (oracle/*.py) against vectors captured from the reference itself
(tests/golden/*.npz, made by oracle/capture_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import adjacency, diffusion as odiff, golden_inputs as gi, layers as olayers
from oracle import manifolds as om

DT = {"f32": torch.float32, "f64": torch.float64}


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("mname", ["PoincareBall", "Hyperboloid"])
@pytest.mark.parametrize("dt", ["f32", "f64"])
def test_manifold_ops(golden, mname, dt):
    g = golden("manifold_" + mname)
    pre = "pb_" if mname == "PoincareBall" else "hb_"
    n = 0
    for c in gi.MANIFOLD_CS:
        for d in gi.MANIFOLD_DIMS:
            for s in gi.MANIFOLD_SCALES:
                k = f"{dt}/c{c}/d{d}/s{s:g}/"
                u, v, w, p, q = (T(g[k + x]) for x in "uvwpq")
                u2, v2, w2 = gi.manifold_case(d, s, u.numpy().dtype)
                assert np.array_equal(u2, u.numpy()) and np.array_equal(w2, w.numpy())
                f = lambda name: getattr(om, pre + name)
                tight = dict(rtol=1e-6, atol_scale=1e-7) if dt == "f32" else dict(rtol=1e-12, atol_scale=1e-13)
                assert_close(f("proj")(u, c), g[k + "proj"], what=k + "proj", **tight)
                assert_close(f("expmap0")(u, c), g[k + "expmap0"], what=k + "expmap0", **tight)
                assert_close(f("logmap0")(p, c), g[k + "logmap0"], what=k + "logmap0", **tight)
                # a 1-element fp32 tensor c (hyp_layers.py:35) takes its sqrt in fp32: 6e-8 relative
                assert_close(f("logmap0")(p, torch.tensor([c])), g[k + "logmap0_ctensor"], what=k + "logmap0c",
                             rtol=1e-6, atol_scale=1e-7)
                assert_close(f("mobius_matvec")(w, p, c), g[k + "mobius_matvec"], what=k + "matvec", **tight)
                assert_close(f("sqdist")(p, q, c), g[k + "sqdist"], what=k + "sqdist", **tight)
                assert_close(f("mobius_add")(p, q, c), g[k + "mobius_add"], what=k + "mobius_add", **tight)
                if mname == "PoincareBall":
                    assert_close(om.pb_expmap(v, p, c), g[k + "expmap"], what=k + "expmap", **tight)
                    assert_close(om.pb_logmap(p, q, c), g[k + "logmap"], what=k + "logmap", **tight)
                else:
                    assert_close(om.hb_proj_tan0(u, c), g[k + "proj_tan0"], what=k + "proj_tan0", **tight)
                n += 1
    assert n == 50


GRAPHS = [("ba", 100), ("ego", 17), ("community", 12), ("citation", 300)]


@pytest.mark.parametrize("kind,n", GRAPHS)
def test_adjacency_bit_exact(golden, kind, n):
    """utils/data_utils.py:103-133 -- indices bit-exact, fp32 values bit-exact."""
    import scipy.sparse as sp
    g = golden("layers")
    k = f"{kind}{n}/"
    adj = gi.synthetic_graph(kind, n)
    assert np.array_equal(adj.indptr, g[k + "adj_in_indptr"])
    assert np.array_equal(adj.indices, g[k + "adj_in_indices"])
    adj_train = sp.csr_matrix((np.ones(len(g[k + "adj_train_indices"])), g[k + "adj_train_indices"],
                               g[k + "adj_train_indptr"]), shape=(n, n))
    idx, val, shape = adjacency.process_adj(adj_train)
    assert idx.dtype == np.int64 and val.dtype == np.float32
    assert np.array_equal(idx, g[k + "coo_indices"])
    assert np.array_equal(val, g[k + "coo_values"])
    rowptr, col, v = adjacency.coo_to_csr(idx, val, n)
    ref = sp.coo_matrix((val, (idx[0], idx[1])), shape=shape).tocsr()
    # scipy's tocsr() additionally sorts each row by column; ours keeps the COO entry
    # order (the order torch.spmm accumulates in), so compare row by row as sorted lists
    assert np.array_equal(rowptr, ref.indptr)
    for r in range(n):
        a, b = rowptr[r], rowptr[r + 1]
        o = np.argsort(col[a:b], kind="stable")
        assert np.array_equal(col[a:b][o], ref.indices[a:b])
        assert np.array_equal(v[a:b][o], ref.data[a:b])
    # and the within-row order is exactly the COO order
    assert np.array_equal(col, idx[1].astype(np.int32)) and np.array_equal(v, val)


@pytest.mark.parametrize("kind,n", GRAPHS)
@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_layers(golden, kind, n, mname, c):
    g = golden("layers")
    k0 = f"{kind}{n}/"
    adj = (T(g[k0 + "coo_indices"]), T(g[k0 + "coo_values"]), (n, n))
    feats = T(g[k0 + "features"])
    for bias in (0, 1):
        k = k0 + f"{mname}/bias{bias}/"
        if k + "encode" not in g.files:
            continue
        Ws = [T(g[k + f"W{i}"]) for i in range(2)]
        bs = [T(g[k + f"b{i}"]) for i in range(2)] if bias else None
        x = feats
        if mname == "Hyperboloid":
            x = torch.cat([torch.zeros_like(x[:, :1]), x], 1)
        M = om.MANIFOLDS[mname]
        h = M["proj"](M["expmap0"](M["proj_tan0"](x, c), c), c)
        assert_close(h, g[k + "lift"], what=k + "lift")
        for i in range(2):
            hin = T(g[k + f"L{i-1}/act"]) if i else h
            h1 = olayers.hyp_linear(mname, hin, Ws[i], c, bs[i] if bias else None)
            assert_close(h1, g[k + f"L{i}/linear"], what=k + f"L{i}/linear")
            h2 = olayers.hyp_agg(mname, T(g[k + f"L{i}/linear"]), adj, c)
            assert_close(h2, g[k + f"L{i}/agg"], what=k + f"L{i}/agg")
            h3 = olayers.hyp_act(mname, T(g[k + f"L{i}/agg"]), c, c)
            assert_close(h3, g[k + f"L{i}/act"], what=k + f"L{i}/act")
        enc = olayers.hgcn_encode(mname, feats, adj, Ws, [c, c, c], bs)
        assert_close(enc, g[k + "encode"], what=k + "encode")      # plain bar (uses <= 0.18 of it)
        dec = olayers.lp_decode(mname, T(g[k + "encode"]), T(g[k0 + "train_edges"]), c, 2.0, 1.0)
        assert_close(dec, g[k + "decode_train"], what=k + "decode")


def test_schedule_and_pri(golden):
    g = golden("diffusion")
    for Tn in (200, 1000):
        tb = odiff.schedule(Tn)
        for ours, theirs in (("alphas_cumprod", "alphas_cumprod"),
                             ("sqrt_alphas_cumprod", "sqrt_alphas_cumprod"),
                             ("sqrt_one_minus_alphas_cumprod", "minus_alphas_cumprod"),
                             ("sqrt_recip_alphas_cumprod", "sqrt_recip_alphas_cumprod"),
                             ("sqrt_recipm1_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")):
            assert_close(tb[ours], g[f"T{Tn}/{theirs}"], rtol=1e-14, atol_scale=1e-15, what=ours)
        pri = odiff.getpri(torch.arange(0, Tn, dtype=torch.long))
        assert pri.dtype == torch.float32
        assert_close(pri, g[f"T{Tn}/getpri"], rtol=1e-6, atol_scale=1e-7, what="getpri")
    assert_close(odiff.schedule(200)["sqrt_alphas_cumprod"], g["T200/get_alphas2_sqrt"], rtol=1e-14, what="alphas2")


def _case(g):
    params = [(T(w), T(b)) for w, b in gi.mlp_params(16)]
    u0 = T(g["c/u0"])
    signs = odiff.sign_table(u0)
    assert np.array_equal(signs.numpy(), np.sign(g["c/direction"]))
    return params, u0, signs, g["c/labels"], T(g["c/t"])


def test_q_sample_and_predict_noise(golden):
    g = golden("diffusion")
    params, u0, signs, labels, t = _case(g)
    tb = odiff.schedule(200)
    for tag, xdt, ndt in (("f64f64", torch.float64, torch.float64),
                          ("f32f64", torch.float32, torch.float64),
                          ("f32f32", torch.float32, torch.float32)):
        x = T(g["c/x"], xdt)
        for r in (0, 1):
            nz = T(g["c/noise"], ndt)
            out = odiff.q_sample(x, t, nz, signs, labels, tb, restrict=bool(r))
            ref = g[f"q_sample/{tag}/restrict{r}/out"]
            assert out.numpy().dtype == ref.dtype
            assert_close(out, ref, rtol=1e-7, atol_scale=1e-8, what=f"q_sample {tag} {r}")
            assert np.array_equal(nz.numpy(), g[f"q_sample/{tag}/restrict{r}/noise_after"])
        x0 = T(g["c/noise"], torch.float32).clamp(-1, 1)
        pn = odiff.predict_noise_from_start(x, t, x0, tb)
        assert pn.numpy().dtype == g[f"predict_noise/{tag}"].dtype
        assert_close(pn, g[f"predict_noise/{tag}"], rtol=1e-7, atol_scale=1e-8, what="predict_noise " + tag)


def test_mlp_and_p_sample(golden):
    g = golden("diffusion")
    params, u0, signs, labels, t = _case(g)
    tb = odiff.schedule(200)
    x = T(g["c/x"], torch.float64)
    assert_close(odiff.mlp_denoiser(x, t, params), g["mlp/int_t"], what="mlp int")
    assert_close(odiff.mlp_denoiser(x, t / 200, params), g["mlp/frac_t"], what="mlp frac")
    for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
        for step in (150, 1, 0):
            pred, x0 = odiff.p_sample(T(g["c/img"], xdt), step, signs, labels, tb, params)
            assert pred.dtype == torch.float64 and x0.dtype == torch.float32
            # x0 is the denoiser output clamped to [-1,1]; its rounding noise scales with the
            # pre-clamp magnitude (integer t up to 199 is an input feature), so that is the scale
            raw = odiff.mlp_denoiser(T(g["c/img"], xdt), torch.full((24,), step), params).abs().max()
            assert_close(x0, g[f"p_sample/{tag}/t{step}/x_start"], what=f"x0 {tag} {step}", scale=raw)
            assert_close(pred, g[f"p_sample/{tag}/t{step}/pred"], what=f"pred {tag} {step}",
                         scale=max(float(raw), float(np.abs(g[f"p_sample/{tag}/t{step}/pred"]).max())))
    img = T(g["c/img"], torch.float32)
    for step in (199, 198, 197):
        img, _ = odiff.p_sample(img, step, signs, labels, tb, params)
        assert_close(img, g[f"p_sample/chain/t{step}"], rtol=1e-4, atol_scale=1e-5, what=f"chain {step}")
    tb1k = odiff.schedule(1000)
    for step in (999, 500, 0):
        pred, x0 = odiff.p_sample(T(g["c/img"], torch.float32), step, signs, labels, tb1k, params)
        raw = odiff.mlp_denoiser(T(g["c/img"], torch.float32), torch.full((24,), step), params).abs().max()
        assert_close(pred, g[f"T1000/p_sample/t{step}/pred"], what=f"T1000 pred {step}",
                     scale=max(float(raw), float(np.abs(g[f"T1000/p_sample/t{step}/pred"]).max())))
        assert_close(x0, g[f"T1000/p_sample/t{step}/x_start"], what=f"T1000 x0 {step}", scale=raw)


def test_forward_loss_and_loop(golden):
    g = golden("diffusion")
    params, u0, signs, labels, t = _case(g)
    tb = odiff.schedule(200)
    for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
        loss, x_start, xt = odiff.forward_loss(T(g["c/x"], xdt), labels, u0, t, T(g["c/noise"], torch.float64),
                                                tb, params)
        assert_close(loss, g[f"forward/{tag}/loss"], what="loss " + tag)
    ret = odiff.p_sample_loop(T(g["c/x"], torch.float32), labels, u0, T(g["c/img"]), tb, params)
    # 200 chained fp32 denoiser evaluations: two correct fp32 evaluations (the reference's sgemm blocking, ours) differ
    # by rounding that is carried from step to step.  How far is measured, not guessed (tests/loop_drift.py): the same
    # loop with the denoiser in fp64 is the exact yardstick, the fp32 loop sits D = 3e-6 (max-norm, relative) from it
    # at the end point, and the reference's own end point may sit as far on the other side.
    from loop_drift import oracle_loop_states
    x32 = T(g["c/x"], torch.float32)
    _, e32 = oracle_loop_states(x32, labels, u0, T(g["c/img"]), tb, params, exact=False)
    _, e64 = oracle_loop_states(x32, labels, u0, T(g["c/img"]), tb, params, exact=True)
    assert torch.equal(e32, ret)
    D = float((e32 - e64).abs().max() / e64.abs().max())
    assert D < 1e-5, D
    assert_close(ret, g["p_sample_loop/f32/ret"], rtol=1e-5, atol_scale=1e-6 + 3 * D, what="loop end point (1e-5 + 3 D)")
