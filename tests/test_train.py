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
from oracle import ddpm as oddpm, diffusion as odiff, golden_inputs as gi

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
