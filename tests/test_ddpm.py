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


# ---- config 3 at its real size: hidden 512, 10 layers, B = 200 (diff_graphset.py:191) -----------------------------
BIG = dict(B=200, D=64, Hd=512, L=10, Tn=1000)


def _big_parts():
    from oracle import golden_inputs as gi
    sd = gi.ddpm_state(BIG["D"], BIG["Hd"], BIG["L"], BIG["Tn"])
    m = lambda pre: [(sd[f"{pre}.layers.{i}.weight"], sd[f"{pre}.layers.{i}.bias"]) for i in range(BIG["L"] + 2)]
    return sd, m("backbone.linear_model1"), sd["backbone.embedding_layer.weight"], m("backbone.linear_model2")


def test_oracle_ddpm_big_golden(golden):
    """the oracle's backbone / sampler step against the reference's outputs at hidden 512 x 10 layers"""
    g = golden("ddpm_big")
    sd, m1, emb, m2 = _big_parts()
    x, idx = torch.from_numpy(g["x"]), torch.from_numpy(g["idx"])
    beta, ab = oddpm.schedule(BIG["Tn"], 1e-4, 0.02)
    ex = g["backbone_f64"]
    assert_close(oddpm.backbone(x, idx, m1, emb, m2), g["backbone"], exact=ex, what="backbone 512x10")
    assert_close(oddpm.backbone(x, torch.full((BIG["B"],), 500), m1, emb, m2), g["infer_t500"], what="infer 512x10")
    xs = x.clone()
    for i in range(2):
        t = BIG["Tn"] - 1 - i
        e = oddpm.backbone(xs, torch.full((BIG["B"],), t), m1, emb, m2)
        xs = oddpm.sampler_step(xs, t, e, torch.from_numpy(g[f"sampler/noise{i}"]), beta, ab)
        assert_close(xs, g[f"sampler/step{i}"], what=f"sampler step {i} 512x10")


@pytest.mark.gpu
@pytest.mark.parametrize("gemm_mode", ["bf16x3", "f16x2", "exact"])
def test_hip_ddpm_big_golden(golden, gemm_mode, monkeypatch):
    """Backbone at the real size: one cooperative launch for the 24 layers (bf16x3 / f16x2 modes; `exact` keeps a
    launch per layer), against the reference's outputs and against the launch-per-layer path."""
    from hypdiff_amd import ops
    from hypdiff_amd.ddpm import Model
    from hypdiff_amd.ddpm_utils import DDPMSampler
    monkeypatch.setattr(ops, "GEMM_MODE", gemm_mode)
    g = golden("ddpm_big")
    sd, _, _, _ = _big_parts()
    model = Model(BIG["D"], BIG["Hd"], BIG["L"], BIG["Tn"], 1e-4, 0.02).cuda().eval()
    model.load_state_dict(sd)
    x, idx = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["idx"]).cuda()
    ex = g["backbone_f64"]
    with torch.no_grad():
        got = model.backbone(x, idx)
        assert_close(got.cpu(), g["backbone"], exact=ex, what="backbone 512x10")
        assert_close(model(x, 500).cpu(), g["infer_t500"], what="infer 512x10")
        if gemm_mode != "exact":
            ws = ops._chain_ws[(x.device.index, torch.cuda.current_stream().cuda_stream)]
            assert ops._lib.load().hd_chain_failed(ops._p(ws), ops._stream()) == 0
            monkeypatch.setattr(ops, "_NO_CHAIN_SMALL", True)
            per_layer = model.backbone(x, idx)
            monkeypatch.setattr(ops, "_NO_CHAIN_SMALL", False)
            assert_close(got.cpu(), per_layer.cpu(), exact=ex, what="one launch vs launch per layer")
            # ragged batches: 1 row, a partial tile, one group only, the largest batch of the one-launch path
            for b in (1, 17, 129, 512):
                xb = torch.randn(b, BIG["D"], device="cuda", generator=torch.Generator("cuda").manual_seed(b))
                ib = torch.randint(0, BIG["Tn"], (b,), device="cuda", generator=torch.Generator("cuda").manual_seed(b))
                one = model.backbone(xb, ib)
                monkeypatch.setattr(ops, "_NO_CHAIN_SMALL", True)
                ref = model.backbone(xb, ib)
                monkeypatch.setattr(ops, "_NO_CHAIN_SMALL", False)
                assert_close(one.cpu(), ref.cpu(), what=f"one launch vs launch per layer, B={b}", atol_scale=3e-6)
            bad = idx.clone()
            bad[5] = BIG["Tn"]                          # nn.Embedding would raise: the row comes back NaN
            out = model.backbone(x, bad)
            assert bool(torch.isnan(out[5]).all()) and bool(torch.isfinite(out[:5]).all()) and bool(torch.isfinite(out[6:]).all())
        s = DDPMSampler(1e-4, 0.02, BIG["Tn"], model, "cuda", (BIG["D"],))
        xs = x.clone()
        for i in range(2):
            xs = s.one_step(xs, BIG["Tn"] - 1 - i, torch.from_numpy(g[f"sampler/noise{i}"]).cuda())
            assert_close(xs.cpu(), g[f"sampler/step{i}"], what=f"sampler {i} 512x10")


@pytest.mark.gpu
def test_linear_chain_small_shapes():
    """hd_linear_chain_small_f32 on stacks that are not the Backbone: narrow first / last layers, a residual first
    layer, no bias, widths that leave k-steps and column tiles partly empty -- against fp64 on the host."""
    from hypdiff_amd import ops
    rs = np.random.RandomState(5)
    for rows, dims, epis in [(200, [64, 512, 512, 64], [1, 2, 0]), (33, [7, 32, 32, 5], [1, 2, 0]),
                             (512, [96, 96, 96, 160, 3], [1, 2, 1, 0]), (16, [512, 512, 512], [0, 2]), (5, [20, 13], [0]),
                             (257, [33, 64, 64], [1, 2])]:
        ws = [torch.from_numpy((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)).cuda()
              for i in range(len(epis))]
        bs = [None if i == 2 else torch.from_numpy(rs.standard_normal(dims[i + 1]).astype(np.float32)).cuda()
              for i in range(len(epis))]
        x = torch.from_numpy(rs.standard_normal((rows, dims[0])).astype(np.float32)).cuda()
        layers = list(zip(ws, bs, epis))
        assert ops.chain_small_ok(x, layers)
        got = ops.linear_chain_small(x, layers)
        h = x.double().cpu()
        for w, b, e in layers:
            z = h @ w.double().cpu().T + (0 if b is None else b.double().cpu())
            h = z if e == 0 else (torch.nn.functional.gelu(z) if e == 1 else h + torch.nn.functional.gelu(z))
        assert_close(got.cpu().numpy(), h.numpy(), what=f"chain {rows} x {dims}", atol_scale=3e-6)
    assert not ops.chain_small_ok(torch.zeros(513, 33, device="cuda"), layers)
    assert not ops.chain_small_ok(x, [(ws[1], bs[1], 2)])            # a residual first layer keeps the launch-per-layer path
    # 2100 launches on one workspace: the tag epoch wraps (2046) and the words are zeroed by the kernel itself
    want = ops.linear_chain_small(x, layers)
    for _ in range(2100):
        got = ops.linear_chain_small(x, layers)
    assert torch.equal(got, want)
    wsb = ops._chain_ws[(x.device.index, torch.cuda.current_stream().cuda_stream)]
    assert ops._lib.load().hd_chain_failed(ops._p(wsb), ops._stream()) == 0
    with pytest.raises(RuntimeError, match="rows must be"):
        ops._lib.call("hd_linear_chain_small_f32", None, 64, 513, None, 1, None, None, 0, None, 64, None, 0, ops._stream())


@pytest.mark.gpu
def test_chain_small_timeout_and_recovery():
    """The bounded wait of hd_linear_chain_small_f32, exercised: with HD_CHAIN_TEST_WITHHOLD=1 one producer never
    publishes its share of the input, its consumers run into the 2 s bound, the launch ends (no hang), its outputs are
    NaN, hd_chain_failed reports 1 exactly once -- and the NEXT launch on the same workspace is clean (the fail word is
    per launch), bit-identical to a launch before the failure."""
    import os
    import time
    from hypdiff_amd import ops
    rs = np.random.RandomState(11)
    dims, epis = [64, 512, 512, 64], [1, 2, 0]
    ws = [torch.from_numpy((rs.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32)).cuda() for i in range(3)]
    bs = [torch.from_numpy(rs.standard_normal(dims[i + 1]).astype(np.float32)).cuda() for i in range(3)]
    x = torch.from_numpy(rs.standard_normal((200, 64)).astype(np.float32)).cuda()
    layers = list(zip(ws, bs, epis))
    good = ops.linear_chain_small(x, layers)
    torch.cuda.synchronize()
    assert torch.isfinite(good).all() and not ops.chain_small_failed()
    os.environ["HD_CHAIN_TEST_WITHHOLD"] = "1"
    try:
        t0 = time.perf_counter()
        bad = ops.linear_chain_small(x, layers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        os.environ.pop("HD_CHAIN_TEST_WITHHOLD", None)
    assert 1.5 < dt < 30.0, dt                                  # ended by the bound, not by luck and not by a hang
    assert torch.isnan(bad[:16]).any()                           # the rows fed by the withheld tile
    again = ops.linear_chain_small(x, layers)                    # no query in between: recovery does not depend on one
    torch.cuda.synchronize()
    assert torch.equal(again, good)
    assert ops.chain_small_failed()                              # reported once ...
    assert not ops.chain_small_failed()                          # ... and cleared
    assert torch.equal(ops.linear_chain_small(x, layers), good)


@pytest.mark.gpu
def test_graphed_train_step_equals_eager():
    """DataParallelTrainer.capture: the whole ddpm.Model training step (loss, backward, one-launch Adam with its step
    count on the device) as one HIP graph.  3 warm-up steps + 2 replays leave exactly the parameters of 5 eager steps;
    versions are bumped per replay; with the draws left to the model, replays see fresh device random numbers."""
    import copy
    from hypdiff_amd.ddpm import Model
    from hypdiff_amd.dist import DataParallelTrainer
    torch.manual_seed(3)
    B, D = 200, 64
    m1 = Model(D, 128, 3, 100, 1e-4, 0.02).cuda().train()
    m2 = copy.deepcopy(m1)
    x = torch.randn(B, D, device="cuda")
    idx = torch.randint(0, 100, (B,), device="cuda")
    eps = torch.randn(B, D, device="cuda")
    t1, t2 = DataParallelTrainer(m1, lr=1e-3), DataParallelTrainer(m2, lr=1e-3)
    eager = [float(t1.step(lambda: m1.loss_fn(x, idx=idx, epsilon=eps))) for _ in range(5)]
    replay = t2.capture(lambda: m2.loss_fn(x, idx=idx, epsilon=eps), warmup=3)
    v0 = [p._version for p in m2.parameters()]
    graphed = [float(replay()) for _ in range(2)]
    assert graphed == eager[3:], (graphed, eager)
    assert all(torch.equal(a, b) for a, b in zip(m1.parameters(), m2.parameters()))
    assert all(p._version > v for p, v in zip(m2.parameters(), v0))
    assert eager[4] < eager[0]
    # draws inside the graph: every replay noises the batch afresh
    replay2 = t2.capture(lambda: m2.loss_fn(x), warmup=1)
    ls = [float(replay2()) for _ in range(4)]
    assert len(set(ls)) == 4 and all(np.isfinite(ls))
