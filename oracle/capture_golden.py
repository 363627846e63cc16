#!/usr/bin/env python3 -B
"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (it imports
/root/reference, which never travels to the GPU box); the resulting .npz files
are committed.  Usage:

    python3 -B oracle/capture_golden.py [--out tests/golden]

Shims (none edits a reference file):
  * `torch_geometric` is not installed.  Placeholder modules are registered so
    that the class *bodies* of models/direction_diffusion.py, models/base_models.py
    and utils/data_utils.py import; the placeholder GCNConv is never called (all
    captures use the mlp denoiser / graphset=True path).  `gcndec` therefore stays
    PARITY UNPINNED.
  * The diffusion class hard-codes `.cuda()` (direction_diffusion.py:108-118,
    203-204); on this CPU-only container Tensor.cuda is made the identity.
  * torch.randint / torch.randn / torch.randn_like are swapped for injected
    tensors while `forward` / `p_sample_loop` run, so the draws are part of the
    fixture.
"""
import argparse
import contextlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

from oracle import golden_inputs as gi  # noqa: E402


def install_shims():
    class _GCNConvPlaceholder(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, *a, **k):
            raise RuntimeError("torch_geometric is absent: GCNConv is a placeholder")

    names = ["torch_geometric", "torch_geometric.nn", "torch_geometric.datasets",
             "torch_geometric.transforms", "torch_geometric.utils",
             "torch_geometric.utils.convert", "torch_geometric.data"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["torch_geometric.nn"].GCNConv = _GCNConvPlaceholder
    mods["torch_geometric.utils.convert"].to_networkx = None
    for a in ("Data", "InMemoryDataset", "download_url", "extract_tar"):
        setattr(mods["torch_geometric.data"], a, object)
    for n, m in mods.items():
        sys.modules[n] = m
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(mods[parent], child, m)
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    os.chdir(REF)


@contextlib.contextmanager
def injected(**repl):
    old = {k: getattr(torch, k) for k in repl}
    try:
        for k, v in repl.items():
            setattr(torch, k, v)
        yield
    finally:
        for k, v in old.items():
            setattr(torch, k, v)


def T(a, dtype=None):
    t = torch.from_numpy(np.array(a, copy=True))     # never alias the input arrays
    return t.to(dtype) if dtype is not None else t


def N(t):
    return t.detach().cpu().numpy().copy()


# ------------------------------------------------------------------------------
def capture_manifolds(out_dir):
    import manifolds
    for mname in ("PoincareBall", "Hyperboloid"):
        man = getattr(manifolds, mname)()
        store = {}
        for dt_name, npdt in (("f32", np.float32), ("f64", np.float64)):
            for c in gi.MANIFOLD_CS:
                for d in gi.MANIFOLD_DIMS:
                    for s in gi.MANIFOLD_SCALES:
                        u_, v_, w_ = gi.manifold_case(d, s, npdt)
                        u, v, w = T(u_), T(v_), T(w_)
                        key = f"{dt_name}/c{c}/d{d}/s{s:g}/"
                        r = {"u": u, "v": v, "w": w}
                        r["proj"] = man.proj(u, c)
                        r["expmap0"] = man.expmap0(u, c)
                        p = man.proj(man.expmap0(man.proj_tan0(u, c), c), c)
                        q = man.proj(man.expmap0(man.proj_tan0(v, c), c), c)
                        r["p"], r["q"] = p, q
                        r["logmap0"] = man.logmap0(p, c)
                        r["mobius_matvec"] = man.mobius_matvec(w, p, c)
                        r["sqdist"] = man.sqdist(p, q, c)
                        r["mobius_add"] = man.mobius_add(p, q, c)
                        r["proj_tan0"] = man.proj_tan0(u, c)
                        if mname == "PoincareBall":
                            r["expmap"] = man.expmap(v, p, c)
                            r["logmap"] = man.logmap(p, q, c)
                        # tensor-valued curvature (hyp_layers.py:35) must agree with float c
                        ct = torch.tensor([c])
                        r["logmap0_ctensor"] = man.logmap0(p, ct)
                        for k, val in r.items():
                            store[key + k] = N(val)
        np.savez_compressed(os.path.join(out_dir, f"manifold_{mname}.npz"), **store)
        print(mname, len(store), "arrays")


# ------------------------------------------------------------------------------
def lp_args(manifold, c, n, feat, dim, bias):
    return types.SimpleNamespace(
        manifold=manifold, c=c, cuda=-1, device="cpu", feat_dim=feat, n_nodes=n,
        model="HGCN", r=2.0, t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, hid1=8,
        hid2=4, num_layers=2, act="relu", dropout=0.0, bias=bias, use_att=0,
        local_agg=0, task="lp", n_heads=4, alpha=0.6, pretrained_embeddings=None,
        use_feats=0)


def capture_layers(out_dir):
    import scipy.sparse as sp
    from utils import data_utils
    from models.base_models import LPModel
    store = {}
    cases = [("ba", 100, 32), ("ego", 17, 16), ("community", 12, 14), ("citation", 300, 32)]
    for kind, n, dim in cases:
        adj = gi.synthetic_graph(kind, n)
        # reference split + preprocessing (utils/data_utils.py:59-79,103-133)
        adj_train, tr, trf, va, vaf, te, tef = data_utils.mask_edges(adj, 0.05, 0.1, 1432)
        adj_t, feats = data_utils.process(adj_train, sp.eye(n).tocsr(), 1, 1)
        g = f"{kind}{n}/"
        store[g + "adj_in_indptr"] = adj.indptr.astype(np.int64)
        store[g + "adj_in_indices"] = adj.indices.astype(np.int64)
        store[g + "adj_train_indptr"] = adj_train.tocsr().indptr.astype(np.int64)
        store[g + "adj_train_indices"] = adj_train.tocsr().indices.astype(np.int64)
        store[g + "coo_indices"] = N(adj_t._indices())
        store[g + "coo_values"] = N(adj_t._values())
        store[g + "train_edges"] = N(tr)
        store[g + "test_edges"] = N(te)
        store[g + "features"] = N(feats)
        for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
            for bias in (0, 1):
                if bias and kind not in ("ego",):
                    continue
                torch.manual_seed(1432)
                args = lp_args(mname, c, n, n, dim, bias)
                model = LPModel(args).eval()
                if bias:
                    with torch.no_grad():
                        for lyr in model.encoder.layers:
                            lyr.linear.bias.normal_(0, 0.1)
                k = g + f"{mname}/bias{bias}/"
                layers = list(model.encoder.layers)
                for i, lyr in enumerate(layers):
                    store[k + f"W{i}"] = N(lyr.linear.weight)
                    store[k + f"b{i}"] = N(lyr.linear.bias)
                with torch.no_grad():
                    x = feats
                    if mname == "Hyperboloid":
                        x = torch.cat([torch.zeros_like(x)[:, 0:1], x], dim=1)
                    man = model.manifold
                    cs = model.encoder.curvatures
                    h = man.proj(man.expmap0(man.proj_tan0(x, cs[0]), c=cs[0]), c=cs[0])
                    store[k + "lift"] = N(h)
                    for i, lyr in enumerate(layers):
                        h1 = lyr.linear.forward(h)
                        h2 = lyr.agg.forward(h1, adj_t)
                        h3 = lyr.hyp_act.forward(h2)
                        store[k + f"L{i}/linear"] = N(h1)
                        store[k + f"L{i}/agg"] = N(h2)
                        store[k + f"L{i}/act"] = N(h3)
                        h = h3
                    enc, _, _, _ = model.encode(feats, adj_t)
                    store[k + "encode"] = N(enc)
                    assert torch.equal(enc, h)
                    store[k + "decode_train"] = N(model.decode(enc, tr))
    np.savez_compressed(os.path.join(out_dir, "layers.npz"), **store)
    print("layers", len(store), "arrays")


# ------------------------------------------------------------------------------
def load_mlp(diff, dim):
    params = gi.mlp_params(dim)
    with torch.no_grad():
        for i, (w, b) in enumerate(params, 1):
            fc = getattr(diff.mlp, f"fc{i}")
            fc.weight.copy_(T(w)); fc.bias.copy_(T(b))


def retable(diff, mod, Tn):
    """Re-derive the instance tables for another horizon with the reference's own
    module-level get_alphas (direction_diffusion.py:71-81), same expressions as
    its __init__ (:96-107)."""
    diff.num_timesteps = Tn
    diff.alphas, betas = mod.get_alphas(Tn)
    diff.alphas_cumprod = torch.cumprod(diff.alphas, dim=0)
    diff.minus_alphas_cumprod = torch.sqrt(1. - diff.alphas_cumprod)
    diff.sqrt_alphas_cumprod = torch.sqrt(diff.alphas_cumprod)
    diff.sqrt_recip_alphas_cumprod = torch.sqrt(1. / diff.alphas_cumprod)
    diff.sqrt_recipm1_alphas_cumprod = torch.sqrt(1. / diff.alphas_cumprod - 1)


def capture_diffusion(out_dir):
    import models.direction_diffusion as dd
    store = {}
    n, dim, k = 24, 16, 3
    args = types.SimpleNamespace(r=2.0, t=1.0)
    torch.manual_seed(1432)
    diff = dd.diffusion(args, dim).eval()
    load_mlp(diff, dim)
    for name in ("alphas", "alphas_cumprod", "sqrt_alphas_cumprod", "minus_alphas_cumprod",
                 "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        store[f"T200/{name}"] = N(getattr(diff, name))
    a2, m2 = diff.get_alphas2(200)
    store["T200/get_alphas2_sqrt"] = N(a2); store["T200/get_alphas2_minus"] = N(m2)
    tt = torch.arange(0, 200, dtype=torch.long)
    store["T200/getpri"] = N(diff.getpri(tt))

    case = gi.diffusion_case(n, dim, k)
    rs = case["rs"]
    labels = case["labels"]
    u0 = T(case["u0"])
    direction = [diff.manifold.logmap0(u, c=1.0) for u in u0]
    store["c/x"], store["c/u0"], store["c/labels"] = case["x"], case["u0"], labels
    store["c/noise"], store["c/img"] = case["noise"], case["img"]
    store["c/direction"] = N(torch.stack(direction))
    t = T(rs.randint(0, 200, size=n).astype(np.int64))
    store["c/t"] = N(t)

    with torch.no_grad():
        for xdt, ndt, tag in ((torch.float64, torch.float64, "f64f64"),
                              (torch.float32, torch.float64, "f32f64"),
                              (torch.float32, torch.float32, "f32f32")):
            x = T(case["x"], xdt)
            for restrict in (False, True):
                nz = T(case["noise"], ndt)
                out = diff.q_sample(x, t, direction, nz, labels, u0, restrict=restrict)
                store[f"q_sample/{tag}/restrict{int(restrict)}/out"] = N(out)
                store[f"q_sample/{tag}/restrict{int(restrict)}/noise_after"] = N(nz)
            x0 = T(case["noise"], torch.float32).clamp(-1, 1)
            store[f"predict_noise/{tag}"] = N(diff.predict_noise_from_start(x, t, x0))

        # denoiser alone, both time encodings (:276-278 integer t, :355 t/T)
        x = T(case["x"], torch.float64)
        store["mlp/int_t"] = N(diff.mlp(x, t))
        store["mlp/frac_t"] = N(diff.mlp(x, t / 200))

        # single p_sample steps from fp32 and fp64 states
        for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
            for step in (150, 1, 0):
                xin = T(case["img"], xdt)
                pred, x0 = diff.p_sample(xin, None, 1, direction, labels, u0, step, None, True)
                store[f"p_sample/{tag}/t{step}/pred"] = N(pred)
                store[f"p_sample/{tag}/t{step}/x_start"] = N(x0)
        # three consecutive steps
        img = T(case["img"], torch.float32)
        for step in (199, 198, 197):
            img, x0 = diff.p_sample(img, None, 1, direction, labels, u0, step, None, True)
            store[f"p_sample/chain/t{step}"] = N(img)

        # training forward with injected draws (x is mutated in place by the reference)
        for tag, xdt in (("f32", torch.float32), ("f64", torch.float64)):
            xin = T(case["x"], xdt).clone()
            nz = T(case["noise"], torch.float64)
            with injected(randint=lambda *a, **k: t.clone(), randn_like=lambda *a, **k: nz.clone()):
                loss = diff(xin, 1, labels, u0, restrict=True, graphset=True)
            store[f"forward/{tag}/loss"] = N(loss)
            store[f"forward/{tag}/x_after"] = N(xin)

        # full sampling loop, injected initial Gaussian
        for tag, xdt in (("f32", torch.float32),):
            xin = T(case["x"], xdt).clone()
            img0 = T(case["img"], torch.float32)
            with injected(randn=lambda *a, **k: img0.clone()):
                ret = diff.p_sample_loop(xin, 1, labels, u0, graphset=True)
            store[f"p_sample_loop/{tag}/ret"] = N(ret)
            store[f"p_sample_loop/{tag}/x_after"] = N(xin)

        # T=1000 horizon (config 2): tables via the reference's own get_alphas
        retable(diff, dd, 1000)
        for name in ("alphas_cumprod", "sqrt_alphas_cumprod", "minus_alphas_cumprod",
                     "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
            store[f"T1000/{name}"] = N(getattr(diff, name))
        store["T1000/getpri"] = N(diff.getpri(torch.arange(0, 1000, dtype=torch.long)))
        for step in (999, 500, 0):
            xin = T(case["img"], torch.float32)
            pred, x0 = diff.p_sample(xin, None, 1, direction, labels, u0, step, None, True)
            store[f"T1000/p_sample/t{step}/pred"] = N(pred)
            store[f"T1000/p_sample/t{step}/x_start"] = N(x0)
    np.savez_compressed(os.path.join(out_dir, "diffusion.npz"), **store)
    print("diffusion", len(store), "arrays")


def capture_diffusionset(out_dir):
    """models/direction_diffusionset.py at (G, N, D) = (4, 14, 14), K = 3 centres per graph."""
    import models.direction_diffusionset as ds
    store = {}
    G_, Nn, D, K = 4, 14, 14, 3
    rs = np.random.RandomState(77)
    torch.manual_seed(1432)
    diff = ds.diffusionset(types.SimpleNamespace(r=2.0, t=1.0), D, D).eval()
    load_mlp(diff, D)
    x = rs.standard_normal((G_, Nn, D)) * 0.2
    x = (x / np.maximum(1.0, np.linalg.norm(x, axis=-1, keepdims=True) / 0.9)).astype(np.float32)
    u0 = (rs.standard_normal((G_, K, D)) * 0.1).astype(np.float32)
    labels = rs.randint(0, K, size=(G_, Nn)).astype(np.int64)
    noise = rs.standard_normal((G_, Nn, D))
    img = rs.standard_normal((G_, Nn, D)).astype(np.float32)
    t = rs.randint(0, 200, size=G_).astype(np.int64)
    for k, v in dict(x=x, u0=u0, labels=labels, noise=noise, img=img, t=t).items():
        store["c/" + k] = v
    u0t, tt = T(u0), T(t)
    direction = [diff.manifold.logmap0(u, c=1.0) for u in u0t]
    with torch.no_grad():
        for r in (0, 1):
            nz = T(noise)
            store[f"q_sample/restrict{r}/out"] = N(diff.q_sample(T(x, torch.float64), tt, direction, nz, labels, u0t,
                                                                 restrict=bool(r)))
            store[f"q_sample/restrict{r}/noise_after"] = N(nz)
        store["mlp/int_t"] = N(diff.mlp(T(x), tt))
        for step in (150, 0):
            pred, x0 = diff.p_sample(T(img), None, 1, direction, labels, u0t, step, None, True)
            store[f"p_sample/t{step}/pred"], store[f"p_sample/t{step}/x_start"] = N(pred), N(x0)
        for restrict in (False, True):
            xin = T(x).clone()
            nz = T(noise)
            with injected(randint=lambda *a, **k: tt.clone(), randn_like=lambda *a, **k: nz.clone()):
                loss = diff(xin, 1, labels, u0t, restrict=restrict, graphset=True)
            store[f"forward/restrict{int(restrict)}/loss"] = N(loss)
        xin = T(x).clone()
        img0 = T(img)
        with injected(randn=lambda *a, **k: img0.clone()):
            ret = diff.p_sample_loop(xin, 1, labels, u0t, graphset=True)
        store["p_sample_loop/ret"], store["p_sample_loop/x_after"] = N(ret), N(xin)
    np.savez_compressed(os.path.join(out_dir, "diffusionset.npz"), **store)
    print("diffusionset", len(store), "arrays")


def capture_ddpm(out_dir):
    """ddpm.Model / DDPMSampler (config 3): B latents of width 64, hidden 64, 3 layers, T=1000
    (diff_graphset.py:191 uses hidden 512 x 10 layers; the small net keeps the fixture small)."""
    import ddpm as rddpm
    import ddpm_utils as rutils
    store = {}
    rs = np.random.RandomState(1432)
    B, D, Hd, L, Tn = 10, 64, 64, 3, 1000
    torch.manual_seed(1432)
    model = rddpm.Model(D, Hd, L, Tn, 1e-4, 0.02).eval()
    for k, v in model.state_dict().items():
        store["sd/" + k] = N(v)
    store["alpha_bars"] = N(model.alpha_bars)
    x = T(rs.standard_normal((B, D)).astype(np.float32))
    eps = T(rs.standard_normal((B, D)).astype(np.float32))
    idx = T(rs.randint(0, Tn, size=B).astype(np.int64))
    store["x"], store["eps"], store["idx"] = N(x), N(eps), N(idx)
    with torch.no_grad():
        store["backbone"] = N(model.backbone(x, idx))
        store["infer_t500"] = N(model(x, 500))
        for restrict in (False, True):
            with injected(randint=lambda *a, **k: idx.clone(), randn_like=lambda *a, **k: eps.clone()):
                out, e, ab = model(x, None, get_target=True, restrict=restrict)
            store[f"train/restrict{int(restrict)}/out"] = N(out)
            store[f"train/restrict{int(restrict)}/alpha_bar"] = N(ab)
        with injected(randint=lambda *a, **k: idx.clone(), randn_like=lambda *a, **k: eps.clone()):
            store["loss"] = N(model.loss_fn(x))
        sampler = rutils.DDPMSampler(1e-4, 0.02, Tn, model, "cpu", (D,))
        noises = [T(rs.standard_normal((B, D)).astype(np.float32)) for _ in range(3)]
        it = iter(noises)
        with injected(randn_like=lambda *a, **k: next(it).clone()):
            gen = sampler._one_diffusion_step(x.clone(), torch.sign(x), False, 'pred_noise')
            for i in range(3):
                store[f"sampler/step{i}"] = N(next(gen))
        for i, nz in enumerate(noises):
            store[f"sampler/noise{i}"] = N(nz)
    np.savez_compressed(os.path.join(out_dir, "ddpm.npz"), **store)
    print("ddpm", len(store), "arrays")


def capture_ddpm_big(out_dir):
    """ddpm.Model / DDPMSampler at config 3's real size (diff_graphset.py:191: latents of width 64, hidden 512,
    10 layers, T = 1000) on B = 200 latents; the weights are oracle.golden_inputs.ddpm_state(...), so only
    inputs and outputs are stored."""
    import ddpm as rddpm
    import ddpm_utils as rutils
    from oracle import golden_inputs as gi
    store = {}
    rs = np.random.RandomState(2708)
    B, D, Hd, L, Tn = 200, 64, 512, 10, 1000
    model = rddpm.Model(D, Hd, L, Tn, 1e-4, 0.02).eval()
    model.load_state_dict(gi.ddpm_state(D, Hd, L, Tn))
    x = T(rs.standard_normal((B, D)).astype(np.float32))
    idx = T(rs.randint(0, Tn, size=B).astype(np.int64))
    store["x"], store["idx"] = N(x), N(idx)
    with torch.no_grad():
        store["backbone"] = N(model.backbone(x, idx))
        store["backbone_f64"] = N(model.double().backbone(x.double(), idx))      # conditioning yardstick
        model.float()
        store["infer_t500"] = N(model(x, 500))
        sampler = rutils.DDPMSampler(1e-4, 0.02, Tn, model, "cpu", (D,))
        noises = [T(rs.standard_normal((B, D)).astype(np.float32)) for _ in range(2)]
        it = iter(noises)
        with injected(randn_like=lambda *a, **k: next(it).clone()):
            gen = sampler._one_diffusion_step(x.clone(), torch.sign(x), False, 'pred_noise')
            for i in range(2):
                store[f"sampler/step{i}"] = N(next(gen))
        for i, nz in enumerate(noises):
            store[f"sampler/noise{i}"] = N(nz)
    np.savez_compressed(os.path.join(out_dir, "ddpm_big.npz"), **store)
    print("ddpm_big", len(store), "arrays")


def capture_train(out_dir):
    """Gradients of the two denoiser training losses, taken by the reference's own autograd:
    diffusion.forward (models/direction_diffusion.py:335-363, mlp denoiser) as diff.py:113-118
    runs it, and ddpm.Model.loss_fn (ddpm.py:71-85) as diff_graphset.py:64-75 runs it.  Inputs
    are the cases of diffusion.npz / ddpm.npz (same seeds), so only losses and gradients are
    stored here."""
    import models.direction_diffusion as dd
    import ddpm as rddpm
    store = {}
    n, dim, k = 24, 16, 3
    torch.manual_seed(1432)
    diff = dd.diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim).train()
    load_mlp(diff, dim)
    case = gi.diffusion_case(n, dim, k)
    rs = case["rs"]
    labels, u0 = case["labels"], T(case["u0"])
    t = T(rs.randint(0, 200, size=n).astype(np.int64))
    store["diffusion/t"] = N(t)
    xin = T(case["x"], torch.float32).clone()
    nz = T(case["noise"], torch.float64)
    with injected(randint=lambda *a, **k: t.clone(), randn_like=lambda *a, **k: nz.clone()):
        loss = diff(xin, 1, labels, u0, restrict=True, graphset=True)
    loss.backward()
    store["diffusion/loss"] = N(loss)
    for i in range(1, 6):
        fc = getattr(diff.mlp, f"fc{i}")
        g = N(fc.weight.grad)
        if g.size > 50000:      # big hidden layers: strided sample + row / column sums keep the fixture small
            store[f"diffusion/grad/fc{i}.weight/sample_7_5"] = g[::7, ::5].copy()
            store[f"diffusion/grad/fc{i}.weight/rowsum"] = g.astype(np.float64).sum(1)
            store[f"diffusion/grad/fc{i}.weight/colsum"] = g.astype(np.float64).sum(0)
        else:
            store[f"diffusion/grad/fc{i}.weight"] = g
        store[f"diffusion/grad/fc{i}.bias"] = N(fc.bias.grad)

    rs = np.random.RandomState(1432)
    B, D, Hd, L, Tn = 10, 64, 64, 3, 1000
    torch.manual_seed(1432)
    model = rddpm.Model(D, Hd, L, Tn, 1e-4, 0.02).train()
    x = T(rs.standard_normal((B, D)).astype(np.float32))
    eps = T(rs.standard_normal((B, D)).astype(np.float32))
    idx = T(rs.randint(0, Tn, size=B).astype(np.int64))
    idx[3] = idx[7]                      # a repeated step: two rows add into one embedding row
    store["ddpm/idx"] = N(idx)
    with injected(randint=lambda *a, **k: idx.clone(), randn_like=lambda *a, **k: eps.clone()):
        loss = model.loss_fn(x)
    loss.backward()
    store["ddpm/loss"] = N(loss)
    for name, p_ in model.named_parameters():
        store["ddpm/grad/" + name] = N(p_.grad)
    # graph-set node-level diffusion (diff.py:230-260 trains diffusionset.forward): inputs of diffusionset.npz
    import models.direction_diffusionset as dset
    gs = np.load(os.path.join(out_dir, "diffusionset.npz"))
    torch.manual_seed(1432)
    dsm = dset.diffusionset(types.SimpleNamespace(r=2.0, t=1.0), 14, 14).train()
    load_mlp(dsm, 14)
    tt, nz = T(gs["c/t"]), T(gs["c/noise"])
    with injected(randint=lambda *a, **k: tt.clone(), randn_like=lambda *a, **k: nz.clone()):
        loss = dsm(T(gs["c/x"]).clone(), 1, gs["c/labels"], T(gs["c/u0"]), restrict=True, graphset=True)
    loss.backward()
    store["diffusionset/loss"] = N(loss)
    for name in ("fc1.weight", "fc1.bias", "fc5.weight", "fc5.bias", "fc3.bias"):
        mod, attr = name.split(".")
        store["diffusionset/grad/" + name] = N(getattr(getattr(dsm.mlp, mod), attr).grad)

    # HGCN encoder (lp_train.py:66-82 trains it through loss.backward()): gradients of
    # sum(encode(x) * G) w.r.t. the two HypLinear weights, G a fixed Gaussian; models, graphs and
    # weights are those of layers.npz (same seeds), so only G-seed-derived outputs are stored
    import scipy.sparse as sp
    from utils import data_utils
    from models.base_models import LPModel
    for kind, n, dim in (("ego", 17, 16), ("ba", 100, 32)):
        adj = gi.synthetic_graph(kind, n)
        adj_train, tr, trf = data_utils.mask_edges(adj, 0.05, 0.1, 1432)[:3]
        adj_t, feats = data_utils.process(adj_train, sp.eye(n).tocsr(), 1, 1)
        neg = trf[:tr.shape[0]]                      # compute_metrics samples nb_edges false edges (:171-172)
        store[f"encoder/{kind}{n}/pos_edges"], store[f"encoder/{kind}{n}/neg_edges"] = N(tr), N(neg)
        for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
            torch.manual_seed(1432)
            model = LPModel(lp_args(mname, c, n, n, dim, 0)).train()
            h = model.encode(feats, adj_t)[0]
            G = T(np.random.RandomState(99).standard_normal(tuple(h.shape)).astype(np.float32))
            (h * G).sum().backward()
            k = f"encoder/{kind}{n}/{mname}/"
            store[k + "h"] = N(h)
            for i, lyr in enumerate(model.encoder.layers):
                store[k + f"W{i}"] = N(lyr.linear.weight)
                store[k + f"gradW{i}"] = N(lyr.linear.weight.grad)
            # the link-prediction training loss of lp_train.py:71-76 (base_models.py:169-184)
            model.zero_grad()
            h = model.encode(feats, adj_t)[0]
            pos, ng = model.decode(h, tr), model.decode(h, neg)
            loss = F.binary_cross_entropy(pos, torch.ones_like(pos)) + F.binary_cross_entropy(ng, torch.zeros_like(ng))
            loss.backward()
            store[k + "lp_loss"] = N(loss)
            for i, lyr in enumerate(model.encoder.layers):
                store[k + f"lp_gradW{i}"] = N(lyr.linear.weight.grad)
    np.savez_compressed(os.path.join(out_dir, "train.npz"), **store)
    print("train", len(store), "arrays")


def capture_train_bias(out_dir):
    """Gradients the reference's own autograd gives for (1) the HGCN encoder with use_bias = 1
    (layers/hyp_layers.py:103-107: proj_tan0 -> expmap0 -> proj -> mobius_add -> proj; Hyperboloid:
    logmap0 -> ptransp0 -> expmap, manifolds/hyperboloid.py:115-146) with NON-zero biases, and (2) the two-point maps of
    the Poincare ball called on their own (manifolds/poincare.py:25-31, 55-71, 85-91) and Hyperboloid.sqdist (:35-41),
    for a fixed Gaussian cotangent.  Inputs come from seeded generators; everything needed to replay is stored."""
    import scipy.sparse as sp
    import manifolds
    from utils import data_utils
    from models.base_models import LPModel
    store = {}
    kind, n, dim = "ego", 17, 16
    adj = gi.synthetic_graph(kind, n)
    adj_train = data_utils.mask_edges(adj, 0.05, 0.1, 1432)[0]
    adj_t, feats = data_utils.process(adj_train, sp.eye(n).tocsr(), 1, 1)
    for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
        torch.manual_seed(1432)
        model = LPModel(lp_args(mname, c, n, n, dim, 1)).train()
        rs = np.random.RandomState(31)
        with torch.no_grad():
            for lyr in model.encoder.layers:
                lyr.linear.bias.copy_(T((rs.standard_normal(lyr.linear.bias.shape) * 0.1).astype(np.float32)))
        h = model.encode(feats, adj_t)[0]
        G = T(np.random.RandomState(99).standard_normal(tuple(h.shape)).astype(np.float32))
        (h * G).sum().backward()
        k = f"encoder_bias/{kind}{n}/{mname}/"
        store[k + "h"], store[k + "G"] = N(h), N(G)
        for i, lyr in enumerate(model.encoder.layers):
            store[k + f"W{i}"], store[k + f"b{i}"] = N(lyr.linear.weight), N(lyr.linear.bias)
            store[k + f"gradW{i}"], store[k + f"gradb{i}"] = N(lyr.linear.weight.grad), N(lyr.linear.bias.grad)
    ball, hyp = manifolds.PoincareBall(), manifolds.Hyperboloid()
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for c in (0.6, 1.0):
            for d in (3, 16, 65):
                for scale in (0.05, 0.4):
                    rs = np.random.RandomState(1000 + d)
                    rows = 9
                    a = rs.standard_normal((rows, d)); a *= scale * rs.uniform(0.2, 1.0, (rows, 1)) / np.linalg.norm(a, axis=1, keepdims=True) / c ** 0.5
                    b = rs.standard_normal((rows, d)); b *= scale * rs.uniform(0.2, 1.0, (rows, 1)) / np.linalg.norm(b, axis=1, keepdims=True) / c ** 0.5
                    g = rs.standard_normal((rows, d))
                    gs = rs.standard_normal(rows)
                    k = f"map2/{tag}/c{c}/d{d}/s{scale}/"
                    store[k + "a"], store[k + "b"], store[k + "g"], store[k + "gs"] = a, b, g, gs
                    for name, fn in (("mobius_add", lambda x, y: ball.mobius_add(x, y, c)),
                                     ("expmap", lambda x, y: ball.expmap(y, x, c)),          # u = b, p = a
                                     ("logmap", lambda x, y: ball.logmap(x, y, c))):
                        x = T(a, dt).requires_grad_(True)
                        y = T(b, dt).requires_grad_(True)
                        out = fn(x, y)
                        (out * T(g, dt)).sum().backward()
                        store[k + name + "/out"], store[k + name + "/da"], store[k + name + "/db"] = N(out), N(x.grad), N(y.grad)
                    x = T(a, dt).requires_grad_(True)
                    y = T(b, dt).requires_grad_(True)
                    out = ball.sqdist(x, y, c)
                    (out * T(gs, dt)).sum().backward()
                    store[k + "sqdist/out"], store[k + "sqdist/da"], store[k + "sqdist/db"] = N(out), N(x.grad), N(y.grad)
                    # hyperboloid points: expmap0 of tangent vectors with a zero time coordinate
                    ha = hyp.expmap0(T(np.concatenate([np.zeros((rows, 1)), a * 3], 1), dt), c).detach()
                    hb = hyp.expmap0(T(np.concatenate([np.zeros((rows, 1)), b * 3], 1), dt), c).detach()
                    store[k + "hb/a"], store[k + "hb/b"] = N(ha), N(hb)
                    x, y = ha.clone().requires_grad_(True), hb.clone().requires_grad_(True)
                    out = hyp.sqdist(x, y, c)
                    (out.reshape(-1) * T(gs, dt)).sum().backward()
                    store[k + "hb/sqdist/out"], store[k + "hb/sqdist/da"], store[k + "hb/sqdist/db"] = N(out), N(x.grad), N(y.grad)
    np.savez_compressed(os.path.join(out_dir, "train_bias.npz"), **store)
    print("train_bias", len(store), "arrays")


def capture_kmeans(out_dir):
    """HyperbolicKMeans.fit of the reference (hyperkmeans.py:42-44: k=3, max_epochs=5) on embeddings
    inside the unit ball; numpy's global generator is seeded and its two initialisation draws
    are replayed into the fixture."""
    from hyperbolic_learning.hyperbolic_kmeans.hkmeans import HyperbolicKMeans
    store = {}
    for tag, n, d, seed in (("n60d16", 60, 16, 7), ("n150d2", 150, 2, 11), ("n90d65", 90, 65, 3)):
        rs = np.random.RandomState(100 + n)
        X = rs.standard_normal((n, d)) * (0.6 / np.sqrt(d))
        X[: n // 3] += 0.35 / np.sqrt(d)
        X[n // 3: 2 * n // 3] -= 0.3 / np.sqrt(d)
        if tag == "n150d2":
            X *= 1.6                                 # some rows outside the ball: exercises the rescale
        np.random.seed(seed)
        init_c = np.random.uniform(0, 0.3, size=(3, d))
        init_l = np.array([np.random.randint(0, 3) for _ in range(n)])
        np.random.seed(seed)
        km = HyperbolicKMeans(n_clusters=3)
        cen, lab = km.fit(X.copy(), max_epochs=5)
        store[tag + "/X"], store[tag + "/init_centroids"], store[tag + "/init_labels"] = X, init_c, init_l
        store[tag + "/centroids"], store[tag + "/labels"] = cen, lab
        store[tag + "/seed"] = np.array(seed)
    np.savez_compressed(os.path.join(out_dir, "kmeans.npz"), **store)
    print("kmeans", len(store), "arrays")


def capture_extras(out_dir):
    """Rows f3 / f4 of SURVEY.md section 8 and the leftovers of a14 / a21: the HNN encoder
    (models/encoders.py:48-71), the graph-VAE tail HNNLayer = HypLinear -> HypNorm -> HypAct of
    models/hyp_layers.py:45-61,172-206,280-319 (PoincareBall, as models/hyp_model.py:19 builds it),
    all seven outputs of utils.data_utils.mask_edges (:59-79), diff.generate's thresholding and
    diff.decode_adj (diff.py:327-380), and diffusion.cal_u0 (models/direction_diffusion.py:130-139)."""
    import scipy.sparse as sp
    import manifolds
    import models.hyp_layers as mhl
    from models.encoders import HNN
    from models.direction_diffusion import diffusion
    from utils import data_utils
    import diff as refdiff
    store = {}
    rs = np.random.RandomState(1432)
    # (1) HNN.encode
    n, feat, dim = 40, 12, 16
    x0 = (rs.standard_normal((n, feat)) * 0.4).astype(np.float32)
    store["hnn/x"] = x0
    for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
        x = T(x0)
        if mname == "Hyperboloid":
            x = torch.cat([torch.zeros_like(x)[:, 0:1], x], dim=1)
        for bias in (0, 1):
            args = lp_args(mname, c, n, x.shape[1], dim, bias)
            torch.manual_seed(1432)
            enc = HNN(torch.tensor([c]), args).eval()
            if bias:
                with torch.no_grad():
                    for lyr in enc.layers:
                        lyr.linear.bias.normal_(0, 0.1)
            k = f"hnn/{mname}/bias{bias}/"
            for i, lyr in enumerate(enc.layers):
                store[k + f"W{i}"] = N(lyr.linear.weight)
                store[k + f"b{i}"] = N(lyr.linear.bias)
            with torch.no_grad():
                store[k + "encode"] = N(enc.encode(x, None))
    # (2) graph-VAE tail
    pb = manifolds.PoincareBall()
    for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        for d_in, d_out in ((64, 64), (20, 36)):
            torch.manual_seed(7)
            lyr = mhl.HNNLayer(d_in, d_out, pb, pb).to(tdt).eval()
            with torch.no_grad():
                lyr.linear.bias.normal_(0, 0.2)
                lyr.norm.norm.weight.uniform_(0.5, 1.5)
                lyr.norm.norm.bias.normal_(0, 0.1)
            k = f"vae/{dt_name}/{d_in}x{d_out}/"
            store[k + "W"] = N(lyr.linear.linear.weight)
            store[k + "bias"] = N(lyr.linear.bias)
            store[k + "ln_w"] = N(lyr.norm.norm.weight)
            store[k + "ln_b"] = N(lyr.norm.norm.bias)
            for scale in (0.05, 1.0, 6.0):
                xin = T((rs.standard_normal((30, d_in)) * scale)).to(tdt)
                xin[0] = 0
                store[k + f"s{scale:g}/x"] = N(xin)
                with torch.no_grad():
                    h1 = lyr.linear(xin)
                    h2 = lyr.norm(h1.clone())
                    store[k + f"s{scale:g}/linear"] = N(h1)
                    store[k + f"s{scale:g}/norm"] = N(h2)
                    store[k + f"s{scale:g}/act"] = N(lyr.hyp_act(h2))
                    store[k + f"s{scale:g}/layer"] = N(lyr(xin))
                    lyr.use_norm = False
                    store[k + f"s{scale:g}/layer_nonorm"] = N(lyr(xin))
                    lyr.use_norm = True
    # (3) mask_edges, every output
    for kind, n in (("citation", 300), ("ego", 17), ("ba", 100)):
        adj = gi.synthetic_graph(kind, n)
        out = data_utils.mask_edges(adj, 0.05, 0.1, 1432)
        k = f"mask/{kind}{n}/"
        store[k + "adj_train_indptr"] = out[0].tocsr().indptr.astype(np.int64)
        store[k + "adj_train_indices"] = out[0].tocsr().indices.astype(np.int64)
        for name, v in zip(("train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges",
                            "test_edges_false"), out[1:]):
            store[k + name] = N(v)
    # (4) decode_adj / generate
    for (n, m) in ((9, 9), (12, 5), (1, 1), (6, 10)):
        a = (rs.rand(n, m) > 0.5).astype(np.float64)
        store[f"decode_adj/{n}x{m}/in"] = a
        store[f"decode_adj/{n}x{m}/out"] = refdiff.decode_adj(a.copy())
    g_n, mx = 6, 9
    h = rs.rand(g_n, 1, mx, mx).astype(np.float32) * 0.98 + 0.01
    h[2, 0] = 0.5                                          # constant graph: max == min -> nan -> no entry set
    got = []
    old_get, old_dump = refdiff.get_graph, refdiff.pickle.dump
    refdiff.get_graph = lambda adj: adj
    refdiff.pickle.dump = lambda obj, f: got.append(obj)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "ds"))
        os.environ["LOG_DIR"] = td
        try:
            refdiff.generate(types.SimpleNamespace(dataset="ds"), h, mx, ["fallback%d" % i for i in range(g_n)])
        finally:
            refdiff.get_graph, refdiff.pickle.dump = old_get, old_dump
    store["generate/h"] = h
    for i, gpred in enumerate(got[0]):
        if isinstance(gpred, str):
            store[f"generate/fallback{i}"] = np.array([1])
        else:
            store[f"generate/adj{i}"] = np.asarray(gpred)
    # (5) cal_u0
    dmod = diffusion(types.SimpleNamespace(r=2.0, t=1.0), 8)
    data = [T((rs.standard_normal((m, 8)) * 0.2).astype(np.float32)) for m in (5, 1, 9)]
    with torch.no_grad():
        u = dmod.cal_u0(data)
    for i, (d_, u_) in enumerate(zip(data, u)):
        store[f"cal_u0/x{i}"] = N(d_)
        store[f"cal_u0/u{i}"] = N(u_)
    np.savez_compressed(os.path.join(out_dir, "extras.npz"), **store)
    print("extras", len(store), "arrays")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out = os.path.abspath(a.out)
    os.makedirs(out, exist_ok=True)
    install_shims()
    torch.set_num_threads(1)          # fixed summation order for the captured vectors
    if a.only in ("", "manifolds"):
        capture_manifolds(out)
    if a.only in ("", "layers"):
        capture_layers(out)
    if a.only in ("", "diffusion"):
        capture_diffusion(out)
    if a.only in ("", "diffusionset"):
        capture_diffusionset(out)
    if a.only in ("", "ddpm"):
        capture_ddpm(out)
    if a.only in ("", "ddpm_big"):
        capture_ddpm_big(out)
    if a.only in ("", "train"):
        capture_train(out)
    if a.only in ("", "train_bias"):
        capture_train_bias(out)
    if a.only in ("", "extras"):
        capture_extras(a.out)
    if a.only in ("", "kmeans"):
        capture_kmeans(out)


if __name__ == "__main__":
    main()
