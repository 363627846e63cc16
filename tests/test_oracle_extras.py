"""CPU: oracle/extras.py and the HNN restatement against the vectors captured from the reference
(tests/golden/extras.npz): graph-VAE tail, mask_edges, decode_adj / generate, cal_u0."""
import numpy as np
import torch

from conftest import assert_close
from oracle import extras, golden_inputs as gi, layers as olayers


def T(a):
    return torch.from_numpy(np.array(a, copy=True))


def test_hnn_encode_golden(golden):
    g = golden("extras")
    x0 = T(g["hnn/x"])
    for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
        x = torch.cat([torch.zeros_like(x0[:, :1]), x0], 1) if mname == "Hyperboloid" else x0
        M = olayers.MANIFOLDS[mname]
        for bias in (0, 1):
            k = f"hnn/{mname}/bias{bias}/"
            h = M["proj"](M["expmap0"](M["proj_tan0"](x, c), c), c)
            for i in range(2):
                h = olayers.hnn_layer(mname, h, T(g[k + f"W{i}"]), c, T(g[k + f"b{i}"]) if bias else None)
            assert_close(h.numpy(), g[k + "encode"], rtol=1e-6, atol_scale=1e-7, what=k)


def test_vae_tail_golden(golden):
    g = golden("extras")
    for dt, tol in (("f32", dict(rtol=2e-6, atol_scale=1e-6)), ("f64", dict(rtol=1e-12, atol_scale=1e-13))):
        for shp in ("64x64", "20x36"):
            k = f"vae/{dt}/{shp}/"
            W, b, lw, lb = (T(g[k + n]) for n in ("W", "bias", "ln_w", "ln_b"))
            for s in ("0.05", "1", "6"):
                x = T(g[k + f"s{s}/x"])
                h1 = extras.vae_hyp_linear(x, W, b)
                assert_close(h1.numpy(), g[k + f"s{s}/linear"], what=k + s + " linear", **tol)
                h2 = extras.vae_hyp_norm(T(g[k + f"s{s}/linear"]), lw, lb)
                assert_close(h2.numpy(), g[k + f"s{s}/norm"], what=k + s + " norm", **tol)
                assert_close(extras.vae_hyp_act(T(g[k + f"s{s}/norm"])).numpy(), g[k + f"s{s}/act"], what=k + s + " act", **tol)
                assert_close(extras.vae_hnn_layer(x, W, b, lw, lb).numpy(), g[k + f"s{s}/layer"], what=k + s + " layer", **tol)
                assert_close(extras.vae_hnn_layer(x, W, b, lw, lb, use_norm=False).numpy(), g[k + f"s{s}/layer_nonorm"],
                             what=k + s + " layer (no norm)", **tol)


def test_mask_edges_golden(golden):
    g = golden("extras")
    for kind, n in (("citation", 300), ("ego", 17), ("ba", 100)):
        out = extras.mask_edges(gi.synthetic_graph(kind, n), 0.05, 0.1, 1432)
        k = f"mask/{kind}{n}/"
        a = out[0].tocsr()
        assert np.array_equal(a.indptr, g[k + "adj_train_indptr"]) and np.array_equal(a.indices, g[k + "adj_train_indices"])
        for name, v in zip(("train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges",
                            "test_edges_false"), out[1:]):
            assert np.array_equal(np.asarray(v).reshape(-1, 2), g[k + name].reshape(-1, 2)), k + name


def test_decode_adj_and_generate_golden(golden):
    g = golden("extras")
    for shp in ("9x9", "12x5", "1x1", "6x10"):
        assert np.array_equal(extras.decode_adj(g[f"decode_adj/{shp}/in"]), g[f"decode_adj/{shp}/out"]), shp
    got = extras.generate_adj(g["generate/h"])
    for i, a in enumerate(got):
        if a is None:
            assert f"generate/fallback{i}" in g.files
        else:
            assert np.array_equal(a, g[f"generate/adj{i}"]), i


def test_cal_u0_golden(golden):
    g = golden("extras")
    data = [T(g[f"cal_u0/x{i}"]) for i in range(3)]
    for i, u in enumerate(extras.cal_u0(data)):
        assert_close(u.numpy(), g[f"cal_u0/u{i}"], rtol=1e-6, atol_scale=1e-7, what=f"cal_u0 {i}")
