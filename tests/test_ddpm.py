"""Graph-level latent DDPM (ddpm.py / ddpm_utils.py, config 3): oracle vs reference-captured
vectors on CPU; HIP path vs the same vectors on the GPU."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import ddpm as oddpm

B, D, Hd, L, Tn = 10, 64, 64, 3, 1000


def _parts(g):
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    m = lambda pre: [(sd[f"{pre}.layers.{i}.weight"], sd[f"{pre}.layers.{i}.bias"]) for i in range(L + 2)]
    return sd, m("backbone.linear_model1"), sd["backbone.embedding_layer.weight"], m("backbone.linear_model2")


def test_oracle_ddpm_golden(golden):
    g = golden("ddpm")
    sd, m1, emb, m2 = _parts(g)
    x, eps, idx = (torch.from_numpy(g[k]) for k in ("x", "eps", "idx"))
    beta, ab = oddpm.schedule(Tn, 1e-4, 0.02)
    assert_close(ab, g["alpha_bars"], rtol=1e-6, what="alpha_bars")
    assert_close(oddpm.backbone(x, idx, m1, emb, m2), g["backbone"], what="backbone")
    assert_close(oddpm.backbone(x, torch.full((B,), 500), m1, emb, m2), g["infer_t500"], what="infer")
    for r in (0, 1):
        xt = oddpm.noise_forward(x, idx, eps, ab, restrict=bool(r))
        assert_close(oddpm.backbone(xt, idx, m1, emb, m2), g[f"train/restrict{r}/out"], what=f"train {r}")
    out = oddpm.backbone(oddpm.noise_forward(x, idx, eps, ab), idx, m1, emb, m2)
    assert_close((out - eps).square().mean(), g["loss"], what="loss")
    xs = x.clone()
    for i in range(3):
        t = Tn - 1 - i
        e = oddpm.backbone(xs, torch.full((B,), t), m1, emb, m2)
        xs = oddpm.sampler_step(xs, t, e, torch.from_numpy(g[f"sampler/noise{i}"]), beta, ab)
        assert_close(xs, g[f"sampler/step{i}"], what=f"sampler {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("gemm_mode", ["bf16x3", "f16x2", "exact"])
def test_hip_ddpm_golden(golden, gemm_mode, monkeypatch):
    from hypdiff_amd import ops
    from hypdiff_amd.ddpm import Model
    from hypdiff_amd.ddpm_utils import DDPMSampler
    monkeypatch.setattr(ops, "GEMM_MODE", gemm_mode)      # both fp32 GEMM kernels (24-layer limb chain)
    g = golden("ddpm")
    sd, _, _, _ = _parts(g)
    model = Model(D, Hd, L, Tn, 1e-4, 0.02).cuda().eval()
    model.load_state_dict(sd)
    x, eps, idx = (torch.from_numpy(g[k]).cuda() for k in ("x", "eps", "idx"))
    with torch.no_grad():
        assert_close(model.backbone(x, idx).cpu(), g["backbone"], what="backbone")
        assert_close(model(x, 500).cpu(), g["infer_t500"], what="infer")
        for r in (0, 1):
            out, e, ab = model(x, idx, get_target=True, restrict=bool(r), epsilon=eps, _train=True)
            assert_close(out.cpu(), g[f"train/restrict{r}/out"], what=f"train {r}")
            assert_close(ab.cpu(), g[f"train/restrict{r}/alpha_bar"], rtol=1e-6, what="alpha_bar")
        assert_close(model.loss_fn(x, idx=idx, epsilon=eps).cpu(), g["loss"], what="loss")
        s = DDPMSampler(1e-4, 0.02, Tn, model, "cuda", (D,))
        xs = x.clone()
        for i in range(3):
            xs = s.one_step(xs, Tn - 1 - i, torch.from_numpy(g[f"sampler/noise{i}"]).cuda())
            assert_close(xs.cpu(), g[f"sampler/step{i}"], what=f"sampler {i}")
        fin = s.sampling(7, x, only_final=True)
        assert fin.shape == (7, D) and bool(torch.isfinite(fin).all())
