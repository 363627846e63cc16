"""Denoiser training step (loss + gradients): oracle autograd vs the reference-captured
gradients on CPU; the HIP forward/backward kernels vs both on the GPU.

Golden: tests/golden/train.npz (oracle/capture_golden.py::capture_train -- the reference's own
loss.backward() on diffusion.forward and ddpm.Model.loss_fn).  Gradient bar: |a-b| <=
1e-5*|b| + 2e-6*max|b| per tensor (fp32 sums over rows in a different order)."""
import types

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import adjacency, ddpm as oddpm, diffusion as odiff, extras as oextras, golden_inputs as gi, layers as olayers
from oracle import manifolds as om

N_, DIM, K_ = 24, 16, 3
B, D, Hd, L, Tn = 10, 64, 64, 3, 1000
GT = dict(rtol=1e-5, atol_scale=2e-6)


def _check_mlp_grads(grads, g, what):
    """grads: {"fc1.weight": tensor, ...}; big layers are pinned by sample + marginals"""
    for i in range(1, 6):
        w = grads[f"fc{i}.weight"].detach().cpu().numpy()
        key = f"diffusion/grad/fc{i}.weight"
        if key in g.files:
            assert_close(w, g[key], what=f"{what} {key}", **GT)
        else:
            scale = float(np.abs(g[key + "/sample_7_5"]).max())
            assert_close(w[::7, ::5], g[key + "/sample_7_5"], what=f"{what} {key} sample", **GT)
            assert_close(w.astype(np.float64).sum(1), g[key + "/rowsum"], what=f"{what} {key} rowsum",
                         rtol=1e-5, atol_scale=1e-5, scale=scale * np.sqrt(w.shape[1]))
            assert_close(w.astype(np.float64).sum(0), g[key + "/colsum"], what=f"{what} {key} colsum",
                         rtol=1e-5, atol_scale=1e-5, scale=scale * np.sqrt(w.shape[0]))
        assert_close(grads[f"fc{i}.bias"].detach().cpu().numpy(), g[f"diffusion/grad/fc{i}.bias"],
                     what=f"{what} fc{i}.bias", **GT)


def _oracle_diffusion_grads(g):
    case = gi.diffusion_case(N_, DIM, K_)
    params = [(torch.from_numpy(w).requires_grad_(), torch.from_numpy(b).requires_grad_())
              for w, b in gi.mlp_params(DIM)]
    t = torch.from_numpy(g["diffusion/t"])
    loss, _, _ = odiff.forward_loss(torch.from_numpy(case["x"].copy()), case["labels"], torch.from_numpy(case["u0"]), t,
                                    torch.from_numpy(case["noise"].copy()), odiff.schedule(200), params)
    loss.backward()
    grads = {}
    for i, (w, b) in enumerate(params, 1):
        grads[f"fc{i}.weight"], grads[f"fc{i}.bias"] = w.grad, b.grad
    return loss, grads


def test_oracle_diffusion_grads_golden(golden):
    g = golden("train")
    loss, grads = _oracle_diffusion_grads(g)
    assert_close(loss.detach(), g["diffusion/loss"], what="loss")
    _check_mlp_grads(grads, g, "oracle")


def _ddpm_inputs():
    rs = np.random.RandomState(1432)
    x = torch.from_numpy(rs.standard_normal((B, D)).astype(np.float32))
    eps = torch.from_numpy(rs.standard_normal((B, D)).astype(np.float32))
    return x, eps


def test_oracle_ddpm_grads_golden(golden):
    g, gd = golden("train"), golden("ddpm")
    sd = {k[3:]: torch.from_numpy(gd[k]).clone().requires_grad_() for k in gd.files if k.startswith("sd/")}
    m = lambda pre: [(sd[f"{pre}.layers.{i}.weight"], sd[f"{pre}.layers.{i}.bias"]) for i in range(L + 2)]
    x, eps = _ddpm_inputs()
    idx = torch.from_numpy(g["ddpm/idx"])
    _, ab = oddpm.schedule(Tn, 1e-4, 0.02)
    out = oddpm.backbone(oddpm.noise_forward(x, idx, eps, ab), idx, m("backbone.linear_model1"),
                         sd["backbone.embedding_layer.weight"], m("backbone.linear_model2"))
    loss = (out - eps).square().mean()
    loss.backward()
    assert_close(loss.detach(), g["ddpm/loss"], what="loss")
    for k, p in sd.items():
        assert_close(p.grad, g["ddpm/grad/" + k], what="oracle ddpm " + k, **GT)


# ------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_diffusion_training_grads(golden):
    from hypdiff_amd.models.direction_diffusion import diffusion
    g = golden("train")
    d = diffusion(types.SimpleNamespace(r=2.0, t=1.0), DIM).train()
    sd = {}
    for i, (w, b) in enumerate(gi.mlp_params(DIM), 1):
        sd[f"fc{i}.weight"], sd[f"fc{i}.bias"] = torch.from_numpy(w), torch.from_numpy(b)
    d.mlp.load_state_dict(sd)
    case = gi.diffusion_case(N_, DIM, K_)
    x = torch.from_numpy(case["x"].copy()).cuda()
    loss = d(x, 1, case["labels"], torch.from_numpy(case["u0"]).cuda(), restrict=True, graphset=True,
             t=torch.from_numpy(g["diffusion/t"]).cuda(), noise=torch.from_numpy(case["noise"].copy()).cuda())
    assert loss.requires_grad
    loss.backward()
    assert_close(loss.detach().cpu(), g["diffusion/loss"], what="loss")
    _check_mlp_grads({n: p.grad for n, p in d.mlp.named_parameters()}, g, "hip")
    # second backward accumulates like autograd does
    loss2 = d(torch.from_numpy(case["x"].copy()).cuda(), 1, case["labels"], torch.from_numpy(case["u0"]).cuda(),
              restrict=True, graphset=True, t=torch.from_numpy(g["diffusion/t"]).cuda(),
              noise=torch.from_numpy(case["noise"].copy()).cuda())
    (0.5 * loss2).backward()
    assert_close(d.mlp.fc5.bias.grad.cpu(), 1.5 * g["diffusion/grad/fc5.bias"], what="accumulated", **GT)


@pytest.mark.gpu
@pytest.mark.parametrize("noise_dtype", [torch.float64, torch.float32])
def test_hip_fused_training_prepass_same_bits(noise_dtype):
    """hd_q_sample_pack_train (normalization + q_sample + the mlp's input concat in one launch) against the three
    launches it replaces: loss, every gradient and the in-place sign flip of the caller's noise are bit-identical."""
    from hypdiff_amd.models.direction_diffusion import diffusion
    rs = np.random.RandomState(11)
    n, dim, k = 777, 64, 3
    d = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=200).train()
    emb = rs.standard_normal((n, dim)).astype(np.float32)
    emb *= 0.5 / np.linalg.norm(emb, axis=1, keepdims=True)
    labels = rs.randint(0, k, size=n)
    u0 = torch.from_numpy((rs.standard_normal((k, dim)) * 0.05).astype(np.float32)).cuda()
    t = torch.from_numpy(rs.randint(0, 200, size=n)).cuda()
    noise0 = torch.from_numpy(rs.standard_normal((n, dim))).to(noise_dtype)
    res = []
    for fuse in (True, False):
        d.fuse_prepass = fuse
        d.mlp.zero_grad(set_to_none=True)
        noise = noise0.clone().cuda()
        loss = d(torch.from_numpy(emb).cuda(), None, labels, u0, restrict=True, graphset=True, t=t, noise=noise)
        loss.backward()
        res.append((loss.detach().clone(), noise.clone(), [p.grad.clone() for p in d.mlp.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1]) and not torch.equal(res[0][1].cpu(), noise0)
    assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2]))


@pytest.mark.gpu
def test_hip_ddpm_training_grads(golden):
    from hypdiff_amd.ddpm import Model
    g, gd = golden("train"), golden("ddpm")
    sd = {k[3:]: torch.from_numpy(gd[k]) for k in gd.files if k.startswith("sd/")}
    model = Model(D, Hd, L, Tn, 1e-4, 0.02).cuda().train()
    model.load_state_dict(sd)
    x, eps = _ddpm_inputs()
    loss = model.loss_fn(x.cuda(), idx=torch.from_numpy(g["ddpm/idx"]).cuda(), epsilon=eps.cuda())
    loss.backward()
    assert_close(loss.detach().cpu(), g["ddpm/loss"], what="loss")
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        assert_close(p.grad.cpu(), g["ddpm/grad/" + k], what="hip ddpm " + k, **GT)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1000, 14), (4097, 64), (15777, 512)])
def test_hip_mlp_grads_vs_oracle_sizes(shape):
    """Odd widths (D = 14: config 1), a row count that is no multiple of any tile, and the
    full config-2 size (split reduction over 62 row slices) against oracle autograd."""
    from hypdiff_amd.models.direction_diffusion import mlp
    n, dim = shape
    torch.manual_seed(n + dim)
    net = mlp(dim).cuda().train()
    x = torch.randn(n, dim, dtype=torch.float64) * 0.5
    t = torch.rand(n)
    target = torch.randn(n, dim)
    from hypdiff_amd import autograd as hag
    out = net(x.cuda(), t.cuda())
    loss = hag.mse_loss(out, target.cuda())
    loss.backward()
    params = [(getattr(net, f"fc{i}").weight.detach().cpu().double().requires_grad_(),
               getattr(net, f"fc{i}").bias.detach().cpu().double().requires_grad_()) for i in range(1, 6)]
    h = torch.cat((x.float().double(), t.double().unsqueeze(1)), 1)
    for i, (w, b) in enumerate(params):
        h = torch.nn.functional.linear(h, w, b)
        if i < 4:
            h = torch.nn.functional.gelu(h)
    ref = torch.nn.functional.mse_loss(h, target.double())
    ref.backward()
    assert_close(loss.detach().cpu(), ref.detach(), what="loss")
    assert_close(out.detach().cpu(), h.detach(), what="out", atol_scale=3e-6)
    for i, (w, b) in enumerate(params, 1):
        fc = getattr(net, f"fc{i}")
        # fp32 sums over n rows vs fp64: bound scales with the gradient's magnitude
        assert_close(fc.weight.grad.cpu(), w.grad, what=f"fc{i}.weight", rtol=1e-5, atol_scale=5e-6)
        assert_close(fc.bias.grad.cpu(), b.grad, what=f"fc{i}.bias", rtol=1e-5, atol_scale=5e-6)


# ------------------------------------------------------------------ encoder (HGCN) gradients
ENC_CASES = [("ego", 17), ("ba", 100)]
ENC_MAN = [("PoincareBall", 0.6), ("Hyperboloid", 1.0)]
# weight gradients are sums over rows of products of O(1) terms: fp32 reassociation shows up
# relative to the largest entry of the gradient
ET = dict(rtol=2e-5, atol_scale=5e-6)


def _enc_G(shape):
    return np.random.RandomState(99).standard_normal(tuple(shape)).astype(np.float32)


@pytest.mark.parametrize("kind,n", ENC_CASES)
@pytest.mark.parametrize("mname,c", ENC_MAN)
def test_oracle_encoder_grads_golden(golden, kind, n, mname, c):
    from oracle import layers as olayers
    g, gl = golden("train"), golden("layers")
    k, k0 = f"encoder/{kind}{n}/{mname}/", f"{kind}{n}/"
    Ws = [torch.from_numpy(g[k + f"W{i}"]).clone().requires_grad_() for i in range(2)]
    adj = (torch.from_numpy(gl[k0 + "coo_indices"]), torch.from_numpy(gl[k0 + "coo_values"]), (n, n))
    h = olayers.hgcn_encode(mname, torch.from_numpy(gl[k0 + "features"]), adj, Ws, [c, c, c])
    assert_close(h.detach(), g[k + "h"], what=k + "h")
    (h * torch.from_numpy(_enc_G(h.shape))).sum().backward()
    for i in range(2):
        assert_close(Ws[i].grad, g[k + f"gradW{i}"], what=k + f"gradW{i}", **ET)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", ENC_CASES)
@pytest.mark.parametrize("mname,c", ENC_MAN)
def test_hip_encoder_grads_golden(golden, kind, n, mname, c):
    from hypdiff_amd.models.base_models import LPModel
    g, gl = golden("train"), golden("layers")
    k, k0 = f"encoder/{kind}{n}/{mname}/", f"{kind}{n}/"
    dim = g[k + "W0"].shape[0]
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=n, n_nodes=n, model="HGCN",
                                 r=2.0, t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, num_layers=2, act="relu",
                                 dropout=0.0, bias=0, use_att=0, local_agg=0)
    lp = LPModel(args).cuda().train()
    for i, lyr in enumerate(lp.encoder.layers):
        lyr.linear.weight.data.copy_(torch.from_numpy(g[k + f"W{i}"]))
    adj = torch.sparse_coo_tensor(torch.from_numpy(gl[k0 + "coo_indices"]).cuda(),
                                  torch.from_numpy(gl[k0 + "coo_values"]).cuda(), (n, n), check_invariants=False)
    h = lp.encode(torch.from_numpy(gl[k0 + "features"]).cuda(), adj)[0]
    assert h.requires_grad
    assert_close(h.detach().cpu(), g[k + "h"], rtol=3e-5, atol_scale=3e-6, what=k + "h")
    (h * torch.from_numpy(_enc_G(h.shape)).cuda()).sum().backward()
    for i, lyr in enumerate(lp.encoder.layers):
        assert_close(lyr.linear.weight.grad.cpu(), g[k + f"gradW{i}"], what=k + f"gradW{i}", **ET)
        assert lyr.linear.bias.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("mname,c", ENC_MAN)
def test_hip_manifold_map_grads_vs_oracle(mname, c):
    """VJP kernels of proj / expmap0 / logmap0 / HypAct (unfused) / HypAgg (unfused) against the
    oracle's autograd, on rows that exercise the proj clip and the plain regime, fp32."""
    import hypdiff_amd.manifolds as hm
    from hypdiff_amd.layers import hyp_layers as L
    from oracle import layers as olayers
    from oracle.manifolds import MANIFOLDS
    man, M = getattr(hm, mname)(), MANIFOLDS[mname]
    rs = np.random.RandomState(3)
    n, d = 257, 24
    scale = np.where(np.arange(n) % 3 == 0, 3.0, 0.2)[:, None]
    u = (rs.standard_normal((n, d)) * scale).astype(np.float32)
    Gm = rs.standard_normal((n, d)).astype(np.float32)

    def both(fn_hip, fn_ref, x_np, what, tol=dict(rtol=2e-5, atol_scale=3e-6)):
        xh = torch.from_numpy(x_np).cuda().requires_grad_()
        xr = torch.from_numpy(x_np).clone().requires_grad_()
        yh, yr = fn_hip(xh), fn_ref(xr)
        assert_close(yh.detach().cpu(), yr.detach(), what=what + " fwd", **tol)
        (yh * torch.from_numpy(Gm).cuda()).sum().backward()
        (yr * torch.from_numpy(Gm)).sum().backward()
        # rows clipped at the ball boundary: d artanh = 1/(1 - z^2) with 1 - z^2 = 8e-3 amplifies the
        # fp32 rounding of z 125-fold in the reference itself -> slack from its own fp64 evaluation
        x64 = torch.from_numpy(x_np).double().requires_grad_()
        (fn_ref(x64) * torch.from_numpy(Gm).double()).sum().backward()
        # (whole-array yardstick: every third row sits exactly ON the clip radius, where `norm > maxnorm`
        # is decided by the rounding of the norm -- the reference's own fp32 and fp64 runs take different
        # branches on some of those rows, and another summation order flips others)
        assert_close(xh.grad.cpu(), xr.grad, what=what + " grad", exact=x64.grad.numpy(), exact_group="all", **tol)

    both(lambda x: man.proj(man.expmap0(man.proj_tan0(x, c), c), c),
         lambda x: M["proj"](M["expmap0"](M["proj_tan0"](x, c), c), c), u, "expmap0+proj")
    p = M["proj"](M["expmap0"](M["proj_tan0"](torch.from_numpy(u), c), c), c).numpy()
    both(lambda x: man.logmap0(x, c), lambda x: M["logmap0"](x, c), p, "logmap0")
    both(lambda x: man.proj(x, c), lambda x: M["proj"](x, c), p * np.float32(1.0), "proj")
    act = L.HypAct(man, c, c, torch.nn.functional.relu)
    both(lambda x: act(x), lambda x: olayers.hyp_act(mname, x, c, c), p, "HypAct")
    adj_np = gi.synthetic_graph("ba", n)
    from oracle import adjacency
    idx, val, shape = adjacency.process_adj(adj_np)
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda(), shape,
                                  check_invariants=False)
    agg = L.HypAgg(man, c, d, 0.0, 0, 0)
    both(lambda x: agg(x, adj), lambda x: olayers.hyp_agg(mname, x, (torch.from_numpy(idx), torch.from_numpy(val), shape), c),
         p, "HypAgg")


@pytest.mark.parametrize("kind,n", ENC_CASES)
@pytest.mark.parametrize("mname,c", ENC_MAN)
def test_oracle_lp_loss_grads_golden(golden, kind, n, mname, c):
    """link-prediction loss (base_models.py:169-184) and its weight gradients, oracle autograd"""
    from oracle import layers as olayers
    import torch.nn.functional as F
    g, gl = golden("train"), golden("layers")
    k, k0 = f"encoder/{kind}{n}/{mname}/", f"{kind}{n}/"
    Ws = [torch.from_numpy(g[k + f"W{i}"]).clone().requires_grad_() for i in range(2)]
    adj = (torch.from_numpy(gl[k0 + "coo_indices"]), torch.from_numpy(gl[k0 + "coo_values"]), (n, n))
    h = olayers.hgcn_encode(mname, torch.from_numpy(gl[k0 + "features"]), adj, Ws, [c, c, c])
    pos = olayers.lp_decode(mname, h, torch.from_numpy(g[f"encoder/{kind}{n}/pos_edges"]), c, 2.0, 1.0)
    neg = olayers.lp_decode(mname, h, torch.from_numpy(g[f"encoder/{kind}{n}/neg_edges"]), c, 2.0, 1.0)
    loss = F.binary_cross_entropy(pos, torch.ones_like(pos)) + F.binary_cross_entropy(neg, torch.zeros_like(neg))
    loss.backward()
    assert_close(loss.detach(), g[k + "lp_loss"], what=k + "lp_loss")
    for i in range(2):
        assert_close(Ws[i].grad, g[k + f"lp_gradW{i}"], what=k + f"lp_gradW{i}", **ET)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", ENC_CASES)
@pytest.mark.parametrize("mname,c", ENC_MAN)
def test_hip_lp_train_step_golden(golden, kind, n, mname, c):
    """encode -> decode -> BCE -> backward of lp_train.py:66-82, all VJPs on HIP kernels"""
    from hypdiff_amd.models.base_models import LPModel
    g, gl = golden("train"), golden("layers")
    k, k0 = f"encoder/{kind}{n}/{mname}/", f"{kind}{n}/"
    dim = g[k + "W0"].shape[0]
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=n, n_nodes=n, model="HGCN",
                                 r=2.0, t=1.0, nb_false_edges=1, nb_edges=1, dim=dim, num_layers=2, act="relu",
                                 dropout=0.0, bias=0, use_att=0, local_agg=0)
    lp = LPModel(args).cuda().train()
    for i, lyr in enumerate(lp.encoder.layers):
        lyr.linear.weight.data.copy_(torch.from_numpy(g[k + f"W{i}"]))
    adj = torch.sparse_coo_tensor(torch.from_numpy(gl[k0 + "coo_indices"]).cuda(),
                                  torch.from_numpy(gl[k0 + "coo_values"]).cuda(), (n, n), check_invariants=False)
    h = lp.encode(torch.from_numpy(gl[k0 + "features"]).cuda(), adj)[0]
    loss = lp.lp_loss(h, torch.from_numpy(g[f"encoder/{kind}{n}/pos_edges"]).cuda(),
                      torch.from_numpy(g[f"encoder/{kind}{n}/neg_edges"]).cuda())
    loss.backward()
    assert_close(loss.detach().cpu(), g[k + "lp_loss"], what=k + "lp_loss")
    for i, lyr in enumerate(lp.encoder.layers):
        assert_close(lyr.linear.weight.grad.cpu(), g[k + f"lp_gradW{i}"], what=k + f"lp_gradW{i}", **ET)


# ------------------------------------------------------------------ gcndec (PARITY UNPINNED)
@pytest.mark.gpu
def test_hip_gcndec_vs_oracle():
    """gcndec forward, training loss gradients and one p_sample step against the oracle's
    restatement of GCNConv's published definition (torch_geometric is absent: parity unpinned,
    no reference vectors exist for this path)."""
    from oracle import gcn as ogcn, adjacency
    from hypdiff_amd.models.direction_diffusion import diffusion
    n, dim = 203, 16
    idx, val, shape = adjacency.process_adj(gi.synthetic_graph("ba", n))
    adj = torch.sparse_coo_tensor(torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda(), shape,
                                  check_invariants=False)
    torch.manual_seed(11)
    d = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim).train()
    with torch.no_grad():
        for p in d.gcn.parameters():
            if p.dim() == 1:
                p.normal_(0, 0.1)
    case = gi.diffusion_case(n, dim, 3)
    rs = np.random.RandomState(4)
    t = torch.from_numpy(rs.randint(0, 200, size=n).astype(np.int64))
    # forward of the denoiser alone
    x = torch.from_numpy(rs.standard_normal((n, dim)))
    convs = [(getattr(d.gcn, f"conv{i}").lin.weight.detach().cpu().double().requires_grad_(),
              getattr(d.gcn, f"conv{i}").bias.detach().cpu().double().requires_grad_()) for i in range(1, 5)]
    fcs = [(getattr(d.gcn, f"fc{i}").weight.detach().cpu().double().requires_grad_(),
            getattr(d.gcn, f"fc{i}").bias.detach().cpu().double().requires_grad_()) for i in range(1, 4)]
    ei = torch.from_numpy(idx)
    with torch.no_grad():
        got = d.gcn(x.cuda(), adj, (t / 200).cuda())
        want = ogcn.gcndec(x, ei, (t / 200).double(), convs, fcs)
    assert_close(got.cpu(), want, what="gcndec fwd", atol_scale=3e-6)
    # training loss + gradients (diff.py:113-118 with graphset=False)
    loss = d(torch.from_numpy(case["x"].copy()).cuda(), adj, case["labels"], torch.from_numpy(case["u0"]).cuda(),
             restrict=True, graphset=False, t=t.cuda(), noise=torch.from_numpy(case["noise"].copy()).cuda())
    loss.backward()
    tables = odiff.schedule(200)
    xs = odiff.pb_logmap(torch.from_numpy(case["u0"])[torch.from_numpy(case["labels"])].double(),
                         torch.from_numpy(case["x"]).double(), 1.0)
    xs = odiff.normalization(xs)
    xt = odiff.q_sample(xs, t, torch.from_numpy(case["noise"].copy()), odiff.sign_table(torch.from_numpy(case["u0"])),
                        case["labels"], tables, restrict=True)
    out = ogcn.gcndec(xt, ei, (t / 200).double(), convs, fcs)
    ref = torch.nn.functional.mse_loss(out, xs.float().double())
    ref.backward()
    assert_close(loss.detach().cpu(), ref.detach(), what="gcndec loss", rtol=2e-5)
    for i, (w, b) in enumerate(convs, 1):
        assert_close(getattr(d.gcn, f"conv{i}").lin.weight.grad.cpu(), w.grad, what=f"conv{i}.w", rtol=2e-5, atol_scale=1e-5)
        assert_close(getattr(d.gcn, f"conv{i}").bias.grad.cpu(), b.grad, what=f"conv{i}.b", rtol=2e-5, atol_scale=1e-5)
    for i, (w, b) in enumerate(fcs, 1):
        assert_close(getattr(d.gcn, f"fc{i}").weight.grad.cpu(), w.grad, what=f"fc{i}.w", rtol=2e-5, atol_scale=1e-5)
        assert_close(getattr(d.gcn, f"fc{i}").bias.grad.cpu(), b.grad, what=f"fc{i}.b", rtol=2e-5, atol_scale=1e-5)
    # one sampling step through the gcndec denoiser
    d.eval()
    with torch.no_grad():
        direction = d.directions(torch.from_numpy(case["u0"]).cuda())
        pred, x0 = d.p_sample(torch.from_numpy(case["img"]).cuda(), None, adj, direction, case["labels"],
                              torch.from_numpy(case["u0"]).cuda(), 150, None, False)
        x0_ref = ogcn.gcndec(torch.from_numpy(case["img"]), ei, torch.full((n,), 150.0, dtype=torch.float64), convs, fcs)
    assert_close(x0.cpu(), x0_ref.clamp(-1, 1), what="gcndec p_sample x0", atol_scale=1e-5,
                 scale=float(x0_ref.abs().max()))
    assert pred.shape == (n, dim) and bool(torch.isfinite(pred).all())


# ------------------------------------------------------------------ training GEMM forms, raw
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16x3", "exact"])
@pytest.mark.parametrize("R,P,Q", [(1, 4, 4), (31, 8, 132), (64, 128, 128), (257, 900, 516), (1000, 14, 17),
                                   (5000, 260, 68), (15777, 64, 512), (15777, 800, 900)])
def test_hip_gemm_tn_shapes(R, P, Q, mode, monkeypatch):
    """dW = A^T B over ragged slice / tile boundaries, exact on small-integer data (fp32 sums of
    integers below 2^24 are exact in any order) and within fp32 reassociation on Gaussian data; both forms
    (hd_gemm_tn_bf16x3_f32: 3 bf16 limbs on the matrix cores, hd_gemm_tn_f32: fp32 MFMA)."""
    from hypdiff_amd import autograd as hag, ops
    monkeypatch.setattr(ops, "GEMM_MODE", mode)
    rs = np.random.RandomState(R + P + Q)
    a = rs.randint(-3, 4, size=(R, P)).astype(np.float32)
    b = rs.randint(-3, 4, size=(R, Q)).astype(np.float32)
    got, cs = hag.gemm_tn(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), want_colsum=True)
    assert np.array_equal(got.cpu().numpy(), (a.astype(np.float64).T @ b.astype(np.float64)).astype(np.float32))
    assert np.array_equal(cs.cpu().numpy(), a.sum(0, dtype=np.float64).astype(np.float32))     # the bias gradient
    a, b = rs.standard_normal((R, P)).astype(np.float32), rs.standard_normal((R, Q)).astype(np.float32)
    cs = torch.full((P,), float("nan"), device="cuda")
    got = torch.full((P, Q), float("nan"), device="cuda")
    hag.gemm_tn(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), out=got, colsum_out=cs)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    bound = 2e-6 * (np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64))
    assert (np.abs(got.cpu().numpy() - want) <= bound + 1e-30).all()
    assert (np.abs(cs.cpu().numpy() - a.sum(0, dtype=np.float64)) <= 2e-6 * np.abs(a).sum(0, dtype=np.float64)).all()
    assert torch.equal(hag.gemm_tn(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()), got)     # reproducible


@pytest.mark.gpu
@pytest.mark.parametrize("R,P,Q", [(1, 4, 4), (31, 8, 132), (64, 128, 128), (257, 900, 516), (1000, 14, 17),
                                   (5000, 260, 68), (15777, 64, 512), (15777, 800, 513), (15777, 900, 800)])
def test_hip_gemm_tn_limbs_shapes(R, P, Q):
    """hd_gemm_tn_limbs_f32: dW = A^T B and db = colsum(A) from bf16x3 limb IMAGES (transposing LDS reads), over ragged
    row blocks / slices / tiles: exact on small-integer data, within fp32 reassociation on Gaussian data, reproducible,
    and pad rows of an image written by the limb GEMM's write-out are zero (the contraction runs over them)."""
    from hypdiff_amd import autograd as hag, ops
    rs = np.random.RandomState(R + P + Q)
    a = rs.randint(-3, 4, size=(R, P)).astype(np.float32)
    b = rs.randint(-3, 4, size=(R, Q)).astype(np.float32)
    pack = lambda x: ops.limb_pack(torch.from_numpy(x).cuda(), "bf16x3")
    got, cs = hag.gemm_tn_limbs(pack(a), pack(b))
    assert np.array_equal(got.cpu().numpy()[:P, :Q], (a.astype(np.float64).T @ b.astype(np.float64)).astype(np.float32))
    assert not got.cpu().numpy()[P:].any() and not got.cpu().numpy()[:, Q:].any()
    assert np.array_equal(cs.cpu().numpy()[:P], a.sum(0, dtype=np.float64).astype(np.float32))
    a, b = rs.standard_normal((R, P)).astype(np.float32), rs.standard_normal((R, Q)).astype(np.float32)
    ai, bi = pack(a), pack(b)
    got, cs = hag.gemm_tn_limbs(ai, bi)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    bound = 2e-6 * (np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64))
    assert (np.abs(got.cpu().numpy()[:P, :Q] - want) <= bound + 1e-30).all()
    assert (np.abs(cs.cpu().numpy()[:P] - a.sum(0, dtype=np.float64)) <= 2e-6 * np.abs(a).sum(0, dtype=np.float64)).all()
    got2, cs2 = hag.gemm_tn_limbs(ai, bi)
    assert torch.equal(got2, got) and torch.equal(cs2, cs)
    if R % 128 and P % 4 == 0:
        # the image of a GEMM product: poison the buffer first, then the write-out must have zeroed the pad rows
        K = 64
        x = torch.from_numpy(rs.standard_normal((R, K)).astype(np.float32)).cuda()
        w = torch.from_numpy(rs.standard_normal((P, K)).astype(np.float32)).cuda()
        torch.empty(1 << 24, dtype=torch.int32, device="cuda").fill_(0x7fc07fc0)       # NaN limbs into the allocator's cache
        _, c, cl = hag.gemm_limbs_train(ops.limb_pack(x), ops.limb_pack(w), None, None, 0, want_f32=True, want_limbs=True)
        got, _ = hag.gemm_tn_limbs(cl, bi)
        want = c.double().cpu().numpy().T @ b.astype(np.float64)
        bound = 2e-6 * (np.abs(c.double().cpu().numpy()).T @ np.abs(b).astype(np.float64))
        assert (np.abs(got.cpu().numpy()[:P, :Q] - want) <= bound + 1e-30).all()


@pytest.mark.gpu
def test_hip_limb_pack_multi_matches_single():
    """hd_limb_pack_multi_f32 (all weight images of a training step in one launch, W and W^T) writes the same bytes
    as hd_limb_pack_f32 of the matrix / of hd_transpose_f32's output; 17 jobs = two launches; ragged shapes."""
    from hypdiff_amd import autograd as hag, ops
    rs = np.random.RandomState(5)
    shapes = [(800, 513), (900, 800), (1, 1), (129, 33), (512, 700), (7, 260)] * 3
    mats = [torch.from_numpy((rs.standard_normal(sh) * 10.0 ** rs.randint(-3, 4)).astype(np.float32)).cuda()
            for sh in shapes[:17]]
    mats[4] = torch.from_numpy(rs.standard_normal((512, 704)).astype(np.float32)).cuda()[:, :700]     # row stride 704
    items = [(m, i % 2 == 1) for i, m in enumerate(mats)]
    got = ops.limb_pack_multi(items)
    for (m, tr), img in zip(items, got):
        want = ops.limb_pack(hag.transpose(m) if tr else m, "bf16x3")
        assert (img.rows, img.cols) == (want.rows, want.cols)
        assert torch.equal(img.buf, want.buf), (tuple(m.shape), tr)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(200, 512, 512), (37, 64, 512), (513, 96, 260)])
def test_hip_gemm_splitk_epilogues(M, N, K):
    """hd_gemm_f32_bf16x3_splitk (few tiles, long K: the graph-level DDPM's batches) under every epilogue against fp64:
    forward none / GELU / residual + GELU with the pre-activation copy, and the input-gradient form."""
    from hypdiff_amd import _lib, autograd as hag
    from scipy.special import erf
    assert _lib.load().hd_gemm_splitk_workspace_floats(M, N, K) > 0
    rs = np.random.RandomState(M + N + K)
    a, w = rs.standard_normal((M, K)).astype(np.float32), (rs.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b, r, z = (rs.standard_normal(sh).astype(np.float32) for sh in ((N,), (M, N), (M, N)))
    ad, wd, bd, rd, zd = (torch.from_numpy(v).cuda() for v in (a, w, b, r, z))
    lin = a.astype(np.float64) @ w.astype(np.float64).T
    bound = 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    gelu = lambda v: 0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))
    dgelu = lambda v: 0.5 * (1.0 + erf(v / np.sqrt(2.0))) + v * np.exp(-0.5 * v * v) / np.sqrt(2.0 * np.pi)
    for epi in (0, 1, 2):
        c, c2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        hag.gemm_ex(ad, wd, bd, rd if epi == 2 else None, None, c, c2 if epi else None, epi)
        pre = lin + b
        want = pre if epi == 0 else (gelu(pre) if epi == 1 else r + gelu(pre))
        assert (np.abs(c.cpu().numpy() - want) <= 2.0 * bound).all(), epi
        if epi:
            assert (np.abs(c2.cpu().numpy() - pre) <= bound).all(), epi
    g, dzb = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    hag.gemm_ex(ad, wd, None, rd, zd, g, dzb, 3)
    assert (np.abs(g.cpu().numpy() - (lin + r)) <= bound).all()
    assert (np.abs(dzb.cpu().numpy() - (lin + r) * dgelu(z.astype(np.float64))) <= 2.0 * bound).all()


@pytest.mark.gpu
def test_hip_dgrad_epilogue_transpose_colsum():
    from hypdiff_amd import autograd as hag
    rs = np.random.RandomState(2)
    M, N, K = 333, 70, 52
    dz = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).cuda()
    w = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)).cuda()
    z = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).cuda()
    resid = torch.from_numpy(rs.standard_normal((M, K)).astype(np.float32)).cuda()
    wt = hag.transpose(w)
    assert torch.equal(wt, w.t().contiguous())
    many = [torch.from_numpy(rs.standard_normal(sh).astype(np.float32)).cuda() for sh in [(70, 52), (1, 1), (513, 64), (65, 129)] * 17]
    many[3] = many[3][:, :100]                                      # a row stride that is not the width
    assert all(torch.equal(t, m.t().contiguous()) for t, m in zip(hag.transpose_multi(many), many))     # 68 jobs: two launches
    g, dzb = torch.empty(M, K, device="cuda"), torch.empty(M, K, device="cuda")
    hag.gemm_ex(dz, wt, None, resid, z, g, dzb, 3)
    g64 = dz.double().cpu() @ w.double().cpu() + resid.double().cpu()
    zz = z.double().cpu().requires_grad_()
    torch.nn.functional.gelu(zz).sum().backward()
    assert_close(g.cpu(), g64, what="dgrad g", atol_scale=3e-6)
    assert_close(dzb.cpu(), g64 * zz.grad, what="dgrad dz", atol_scale=3e-6)
    assert_close(hag.colsum(dz).cpu(), dz.double().cpu().sum(0), what="colsum", atol_scale=3e-6)
    big = torch.from_numpy(rs.standard_normal((20000, 33)).astype(np.float32)).cuda()
    assert_close(hag.colsum(big).cpu(), big.double().cpu().sum(0), what="colsum big", atol_scale=1e-5,
                 scale=float(big.abs().sum(0).max()))


# ---- use_bias = 1 and the two-point maps on their own: the reference's own autograd (tests/golden/train_bias.npz) ----
MAP2 = [(tag, c, d, s) for tag in ("f32", "f64") for c in (0.6, 1.0) for d in (3, 16, 65) for s in (0.05, 0.4)]


def _map2_tol(tag):
    # gradients of fp32 maps next to cancelling terms (1 - c |x|^2): 2e-5 of the largest gradient entry; fp64: 1e-10
    return dict(rtol=2e-5, atol_scale=2e-5) if tag == "f32" else dict(rtol=1e-9, atol_scale=1e-10)


def test_oracle_two_point_map_grads_vs_reference(golden):
    """the oracle's autograd (incl. the detached _lambda_x) against the reference's, manifolds/poincare.py:25-31,55-91"""
    g = golden("train_bias")
    for tag, c, d, s in MAP2:
        dt = torch.float32 if tag == "f32" else torch.float64
        k = f"map2/{tag}/c{c}/d{d}/s{s}/"
        for name, fn in (("mobius_add", lambda x, y: om.pb_mobius_add(x, y, c)), ("expmap", lambda x, y: om.pb_expmap(y, x, c)),
                         ("logmap", lambda x, y: om.pb_logmap(x, y, c)), ("sqdist", lambda x, y: om.pb_sqdist(x, y, c))):
            x = torch.from_numpy(g[k + "a"]).to(dt).requires_grad_(True)
            y = torch.from_numpy(g[k + "b"]).to(dt).requires_grad_(True)
            out = fn(x, y)
            cot = torch.from_numpy(g[k + ("gs" if name == "sqdist" else "g")]).to(dt)
            (out * cot).sum().backward()
            tol = dict(rtol=1e-5, atol_scale=2e-6) if tag == "f32" else dict(rtol=1e-10, atol_scale=1e-12)
            assert_close(out.detach().numpy(), g[k + name + "/out"], what=k + name + " out", **tol)
            assert_close(x.grad.numpy(), g[k + name + "/da"], what=k + name + " da", **tol)
            assert_close(y.grad.numpy(), g[k + name + "/db"], what=k + name + " db", **tol)
        x = torch.from_numpy(g[k + "hb/a"]).requires_grad_(True)
        y = torch.from_numpy(g[k + "hb/b"]).requires_grad_(True)
        out = om.hb_sqdist(x, y, c)
        (out.reshape(-1) * torch.from_numpy(g[k + "gs"]).to(dt)).sum().backward()
        tol = dict(rtol=1e-5, atol_scale=2e-6) if tag == "f32" else dict(rtol=1e-10, atol_scale=1e-12)
        assert_close(x.grad.numpy(), g[k + "hb/sqdist/da"], what=k + "hb sqdist da", **tol)


def test_oracle_bias_encoder_grads_vs_reference(golden):
    g = golden("train_bias")
    kind, n = "ego", 17
    adj = gi.synthetic_graph(kind, n)
    import scipy.sparse as sp
    idx, val, shape = adjacency.process_adj(oextras.mask_edges(adj, 0.05, 0.1, 1432)[0])
    x = torch.eye(n)
    for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
        k = f"encoder_bias/{kind}{n}/{mname}/"
        Ws = [torch.from_numpy(g[k + f"W{i}"]).requires_grad_(True) for i in range(2)]
        bs = [torch.from_numpy(g[k + f"b{i}"]).requires_grad_(True) for i in range(2)]
        h = olayers.hgcn_encode(mname, x, (torch.from_numpy(idx), torch.from_numpy(val), shape), Ws, [c, c, c], biases=bs)
        assert_close(h.detach().numpy(), g[k + "h"], what=k + "h", rtol=1e-5, atol_scale=2e-6)
        (h * torch.from_numpy(g[k + "G"])).sum().backward()
        for i in range(2):
            assert_close(Ws[i].grad.numpy(), g[k + f"gradW{i}"], what=k + f"gradW{i}", rtol=2e-5, atol_scale=2e-5)
            assert_close(bs[i].grad.numpy(), g[k + f"gradb{i}"], what=k + f"gradb{i}", rtol=2e-5, atol_scale=2e-5)


@pytest.mark.gpu
def test_hip_two_point_map_grads_vs_reference(golden):
    """hd_pb_map2_bwd / hd_{pb,hb}_sqdist_bwd through the drop-in manifold classes and torch autograd"""
    import hypdiff_amd.manifolds as hm
    g = golden("train_bias")
    ball, hyp = hm.PoincareBall(), hm.Hyperboloid()
    for tag, c, d, s in MAP2:
        dt = torch.float32 if tag == "f32" else torch.float64
        k = f"map2/{tag}/c{c}/d{d}/s{s}/"
        for name, fn in (("mobius_add", lambda x, y: ball.mobius_add(x, y, c)), ("expmap", lambda x, y: ball.expmap(y, x, c)),
                         ("logmap", lambda x, y: ball.logmap(x, y, c)), ("sqdist", lambda x, y: ball.sqdist(x, y, c))):
            x = torch.from_numpy(g[k + "a"]).to(dt).cuda().requires_grad_(True)
            y = torch.from_numpy(g[k + "b"]).to(dt).cuda().requires_grad_(True)
            out = fn(x, y)
            cot = torch.from_numpy(g[k + ("gs" if name == "sqdist" else "g")]).to(dt).cuda()
            (out * cot).sum().backward()
            assert_close(out.detach().cpu().numpy(), g[k + name + "/out"], what=k + name + " out",
                         **(dict(rtol=1e-5, atol_scale=2e-6) if tag == "f32" else dict(rtol=1e-10, atol_scale=1e-12)))
            assert_close(x.grad.cpu().numpy(), g[k + name + "/da"], what=k + name + " da", **_map2_tol(tag))
            assert_close(y.grad.cpu().numpy(), g[k + name + "/db"], what=k + name + " db", **_map2_tol(tag))
        x = torch.from_numpy(g[k + "hb/a"]).cuda().requires_grad_(True)
        y = torch.from_numpy(g[k + "hb/b"]).cuda().requires_grad_(True)
        out = hyp.sqdist(x, y, c)
        (out.reshape(-1) * torch.from_numpy(g[k + "gs"]).to(dt).cuda()).sum().backward()
        assert_close(x.grad.cpu().numpy(), g[k + "hb/sqdist/da"], what=k + "hb sqdist da", **_map2_tol(tag))
        assert_close(y.grad.cpu().numpy(), g[k + "hb/sqdist/db"], what=k + "hb sqdist db", **_map2_tol(tag))
    # a (1, d) second operand broadcast over the rows (HypLinear's bias row): its gradient is the sum over rows
    x = (torch.randn(40, 16, device="cuda") * 0.1).requires_grad_(True)
    b = (torch.randn(1, 16, device="cuda") * 0.1).requires_grad_(True)
    ball.mobius_add(x, b, 0.6).square().sum().backward()
    xr, br = x.detach().cpu().double().requires_grad_(True), b.detach().cpu().double().requires_grad_(True)
    om.pb_mobius_add(xr, br, 0.6).square().sum().backward()
    assert_close(b.grad.cpu().numpy(), br.grad.numpy(), what="broadcast bias row grad", rtol=2e-5, atol_scale=2e-5)
    assert_close(x.grad.cpu().numpy(), xr.grad.numpy(), what="broadcast rows grad", rtol=2e-5, atol_scale=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_hip_bias_encoder_grads_vs_reference(golden, mname, c):
    """HGCN.encode with use_bias = 1 and non-zero biases: forward and weight / bias gradients for a fixed cotangent against
    the reference's loss.backward() (layers/hyp_layers.py:103-107)"""
    import types
    from hypdiff_amd.models.encoders import HGCN
    g = golden("train_bias")
    kind, n, dim = "ego", 17, 16
    adj = gi.synthetic_graph(kind, n)
    idx, val, shape = adjacency.process_adj(oextras.mask_edges(adj, 0.05, 0.1, 1432)[0])
    k = f"encoder_bias/{kind}{n}/{mname}/"
    feat = n + (1 if mname == "Hyperboloid" else 0)
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=feat, dim=dim, num_layers=2,
                                 act="relu", dropout=0.0, bias=1, use_att=0, local_agg=0)
    enc = HGCN(torch.tensor([c]).cuda(), args).cuda().train()
    for i, lyr in enumerate(enc.layers):
        lyr.linear.load_state_dict({"weight": torch.from_numpy(g[k + f"W{i}"]), "bias": torch.from_numpy(g[k + f"b{i}"])})
    x = torch.eye(n).cuda()
    if mname == "Hyperboloid":
        x = torch.cat([torch.zeros_like(x[:, :1]), x], 1)
    adj_t = torch.sparse_coo_tensor(torch.from_numpy(idx).cuda(), torch.from_numpy(val).cuda(), shape, check_invariants=False)
    h = enc.encode(x, adj_t)
    assert_close(h.detach().cpu().numpy(), g[k + "h"], what=k + "h", rtol=1e-5, atol_scale=2e-6)
    (h * torch.from_numpy(g[k + "G"]).cuda()).sum().backward()
    for i, lyr in enumerate(enc.layers):
        assert_close(lyr.linear.weight.grad.cpu().numpy(), g[k + f"gradW{i}"], what=k + f"gradW{i}", rtol=2e-5, atol_scale=2e-5)
        assert_close(lyr.linear.bias.grad.cpu().numpy(), g[k + f"gradb{i}"], what=k + f"gradb{i}", rtol=2e-5, atol_scale=2e-5)
