"""hd_mlp_chain_limbs_f32 (csrc/gemm_chain.hip): the mlp denoiser's five Linear layers (models/direction_diffusion.py:56-69)
as ONE persistent launch must give, bit for bit, what five hd_gemm_limbs_f32 launches give (same tiles, same K order,
same write-out arithmetic) -- on the denoiser's own shapes at configs[1]'s row count, on ragged shapes (rows, N and K
off every tile boundary), and launch after launch on one workspace (the kernel cleans it itself)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layers(dims, seed, dev, bias=True):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(len(dims) - 1):
        k, n = dims[i], dims[i + 1]
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
        b = torch.randn(n, generator=g).to(dev) if bias else None
        out.append((w, b, 0 if i == len(dims) - 2 else 1))
    return out


def _per_layer(ops, a, layers):
    """one hd_gemm_limbs_f32 launch per layer (the per-layer kernel itself, not the chain kernel run layer by layer)"""
    xf = None
    via, ops.GEMM_VIA_CHAIN = ops.GEMM_VIA_CHAIN, False
    try:
        for i, (w, b, epi) in enumerate(layers):
            last = i == len(layers) - 1
            wl, _ = ops.weight_limbs(w, "bf16x3")
            xf, a = ops.gemm_limbs(a, wl, b, epi=epi, want_f32=last, want_limbs=not last)
    finally:
        ops.GEMM_VIA_CHAIN = via
    return xf


@pytest.mark.parametrize("rows,dims", [
    (15777, [513, 800, 900, 800, 700, 512]),        # configs[1]: the bench workload's denoiser
    (7850, [15, 800, 900, 800, 700, 14]),           # configs[0]: D = 14
    (4099, [33, 70, 129, 5]),                       # rows, N and K off every boundary; a 1-tile last layer
    (4096, [64, 64]),                               # single layer is refused by the wrapper -> per-layer path
    (5000, [100, 96, 32, 31, 200]),                 # tiles of fewer than 4 K slabs: the blocking successor path
])
def test_chain_equals_per_layer_launches(rows, dims):
    from hypdiff_amd import ops
    dev = "cuda:0"
    torch.manual_seed(1)
    x = torch.randn(rows, dims[0], device=dev)
    layers = _layers(dims, 7, dev)
    a = ops.limb_pack(x)
    ref = _per_layer(ops, a, layers)
    if len(layers) < 2:
        assert not ops.mlp_chain_ok(a, layers)
        return
    assert ops.mlp_chain_ok(a, layers)
    for rep in range(3):                              # launch after launch on the same workspace
        got = ops.mlp_chain(a, layers)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), (rep, float((got - ref).abs().max()))
    assert not ops.mlp_chain_failed()


def test_chain_matches_fp64_and_linear_chain_routes_to_it():
    """ops.linear_chain takes the one-launch path for a big bf16x3 stack; result within fp32 rounding of fp64"""
    from hypdiff_amd import ops
    dev = "cuda:0"
    rows, dims = 6000, [513, 800, 900, 800, 700, 512]
    torch.manual_seed(2)
    x = torch.randn(rows, dims[0], device=dev)
    layers = _layers(dims, 11, dev)
    mode = ops.GEMM_MODE
    ops.GEMM_MODE = "bf16x3"
    try:
        got = ops.linear_chain(x, layers)
    finally:
        ops.GEMM_MODE = mode
    h = x.double()
    for w, b, epi in layers:
        h = h @ w.double().T + b.double()
        if epi == 1:
            h = torch.nn.functional.gelu(h)
    err = float((got.double() - h).abs().max() / h.abs().max())
    assert err < 3e-6, err                              # five chained fp32 GEMMs


def test_chain_no_bias_and_many_launches_stay_clean():
    from hypdiff_amd import ops
    dev = "cuda:0"
    rows, dims = 9000, [128, 256, 192, 64]
    torch.manual_seed(3)
    x = torch.randn(rows, dims[0], device=dev)
    layers = _layers(dims, 5, dev, bias=False)
    a = ops.limb_pack(x)
    ref = _per_layer(ops, a, layers)
    outs = [ops.mlp_chain(a, layers) for _ in range(20)]      # back to back, no host sync in between
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref) for o in outs)
    assert not ops.mlp_chain_failed()


@pytest.mark.parametrize("rows,dims", [
    (15777, [516, 800, 900, 800, 700, 512]),        # the bench's training leg (cat(x, t) padded to a 16-byte row)
    (4100, [36, 72, 132, 8]),                       # ragged: the last column tiles narrow, a partial row block
])
def test_training_stack_through_the_chain_equals_per_layer_launches(rows, dims, monkeypatch):
    """LinearStack on limb images: forward (with the fp32 pre-activations kept) and the input-gradient GEMMs
    dz_{l-1} = (dz_l W_l) gelu'(z_{l-1}) as one-launch chains (epilogue 3) must give, bit for bit, the output and every
    parameter gradient of the per-layer launches (hd_gemm_limbs_train_f32)."""
    from hypdiff_amd import autograd as ag, ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, dims[0], generator=g).to(dev)
    gy = torch.randn(rows, dims[-1], generator=g).to(dev)
    kinds = tuple([ag.GELU] * (len(dims) - 2) + [ag.LIN])

    def run(no_chain):
        monkeypatch.setattr(ops, "_NO_MLP_CHAIN", no_chain)
        lin = [torch.nn.Linear(dims[i], dims[i + 1]).to(dev) for i in range(len(dims) - 1)]
        gg = torch.Generator().manual_seed(5)
        for m in lin:
            m.weight.data = (torch.randn(m.weight.shape, generator=gg) / m.weight.shape[1] ** 0.5).to(dev)
            m.bias.data = torch.randn(m.bias.shape, generator=gg).to(dev)
        y = ag.linear_stack(x, lin, kinds)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), [p.grad.clone() for m in lin for p in (m.weight, m.bias)]

    y0, g0 = run(True)
    y1, g1 = run(False)
    assert ops._mlp_chain_ws, "the chain did not run"
    assert torch.equal(y0, y1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert not ops.mlp_chain_failed()


def test_two_chains_at_once_on_two_streams():
    """Two launches in flight on one GPU, each with its own workspace (one per stream): no wait of the kernel depends on
    residency (an item depends only on items handed out before it), so they may share the CUs in any proportion."""
    from hypdiff_amd import ops
    dev = "cuda:0"
    torch.manual_seed(2)
    dims = [65, 200, 136, 72]
    xs = [torch.randn(6000, dims[0], device=dev), torch.randn(9000, dims[0], device=dev)]
    layers = _layers(dims, 11, dev)
    imgs = [ops.limb_pack(x) for x in xs]
    refs = [_per_layer(ops, a, layers) for a in imgs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(4):
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[k].append(ops.mlp_chain(imgs[k], layers))
    torch.cuda.synchronize()
    for k in range(2):
        for got in outs[k]:
            assert torch.equal(got, refs[k])
    assert not ops.mlp_chain_failed()


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [40, 1000, 15777])
def test_weight_gradients_in_one_launch_equal_one_launch_each(rows):
    """hd_gemm_tn_limbs_multi_f32 (the weight gradients of a backward pass as ONE launch of resident workgroups) cuts and
    sums every product as hd_gemm_tn_limbs_f32 does: results and column sums bit for bit, for ragged shapes, with outputs
    handed in or allocated, at a row count that needs no slices (40), a few (1000) and the bench's (15 777)."""
    from hypdiff_amd import autograd as ag, ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(rows)
    shapes = [(136, 72), (200, 136), (68, 200), (512, 700), (900, 800)]
    imgs = []
    for P, Q in shapes:
        a = torch.randn(rows, P, generator=g).to(dev)
        b = torch.randn(rows, Q, generator=g).to(dev)
        imgs.append((ops.limb_pack(a, "bf16x3"), ops.limb_pack(b, "bf16x3"), a, b))
    single = [ag.gemm_tn_limbs(ai, bi) for ai, bi, _, _ in imgs]
    outs = [None, torch.empty(200, 136, device=dev), None, torch.empty(512, 700, device=dev), None]
    css = [None, torch.empty(200, device=dev), None, None, torch.empty(900, device=dev)]
    multi = ag.gemm_tn_limbs_multi([(ai, bi, o, c) for (ai, bi, _, _), o, c in zip(imgs, outs, css)])
    torch.cuda.synchronize()
    for (w0, c0), (w1, c1), (_, _, a, b), o, c in zip(single, multi, imgs, outs, css):
        assert torch.equal(w0, w1) and torch.equal(c0, c1)
        if o is not None:
            assert w1.data_ptr() == o.data_ptr()
        if c is not None:
            assert c1.data_ptr() == c.data_ptr()
        ref = a.double().t() @ b.double()
        bound = 2e-6 * (a.double().abs().t() @ b.double().abs())
        assert bool(((w1.double()[: ref.shape[0], : ref.shape[1]] - ref).abs() <= bound + 1e-30).all())
    # a second call on the same stream reuses the workspace: same bits
    again = ag.gemm_tn_limbs_multi([(ai, bi, None, None) for ai, bi, _, _ in imgs])
    for (w0, c0), (w1, c1) in zip(single, again):
        assert torch.equal(w0, w1) and torch.equal(c0, c1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(15777, 512, 512), (4100, 272, 130 * 16)])
def test_single_gemm_as_a_chain_of_one_layer(shape, monkeypatch):
    """ops.gemm_limbs over many rows through the chain kernel's wide tiles (a chain of ONE layer) = hd_gemm_limbs_f32,
    bit for bit: fp32 output and limb image, with and without bias + GELU."""
    from hypdiff_amd import ops
    M, K, N = shape
    dev = "cuda:0"
    g = torch.Generator().manual_seed(M + N)
    a = ops.limb_pack(torch.randn(M, K, generator=g).to(dev), "bf16x3")
    w = ops.limb_pack(torch.randn(N, K, generator=g).to(dev) / K ** 0.5, "bf16x3")
    b = torch.randn(N, generator=g).to(dev)
    for bias, epi, limbs in ((None, 0, False), (b, 1, True)):
        monkeypatch.setattr(ops, "GEMM_VIA_CHAIN", False)
        c0, l0 = ops.gemm_limbs(a, w, bias, epi=epi, want_limbs=limbs)
        monkeypatch.setattr(ops, "GEMM_VIA_CHAIN", True)
        c1, l1 = ops.gemm_limbs(a, w, bias, epi=epi, want_limbs=limbs)
        torch.cuda.synchronize()
        assert torch.equal(c0, c1)
        if limbs:
            assert torch.equal(ops.limb_unpack(l0), ops.limb_unpack(l1))
    assert not ops.mlp_chain_failed()


@pytest.mark.gpu
def test_cached_weight_image_built_on_another_stream_is_awaited():
    """ops.weight_limbs / pad_k_cached fill their cached buffers on the stream that was current at the miss.  A hit on
    another stream right afterwards must wait for that fill (ops._tag / ops._ready): here the fill sits behind ~20 ms of
    busy-waiting on a side stream, and the main stream unpacks the image at once."""
    from hypdiff_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    w = torch.randn(900, 802, generator=g).to(dev)
    side = torch.cuda.Stream()
    ops.pad_k_cached(w, 804)                              # first sighting: a padded operand is cached at its second miss
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(40_000_000)
        img_side, _ = ops.weight_limbs(w)
        pad_side = ops.pad_k_cached(w, 804)
    img_main, _ = ops.weight_limbs(w)                     # hits while the side stream is still asleep
    pad_main = ops.pad_k_cached(w, 804)
    assert img_main is img_side and pad_main is pad_side
    back = ops.limb_unpack(img_main)                      # main stream
    cols = pad_main[:, :802].clone()
    torch.cuda.synchronize()
    assert torch.equal(back, w) and torch.equal(cols, w)


@pytest.mark.gpu
def test_weight_gradients_in_one_launch_edge_cases():
    """hd_gemm_tn_limbs_multi_f32: no rows -> zero gradients (as hd_gemm_tn_limbs_f32 and an empty torch sum); a single
    job; more than 8 jobs, no jobs and a misaligned workspace are refused with a message, nothing is launched."""
    from hypdiff_amd import _lib, autograd as ag, ops
    dev = "cuda:0"
    # R = 0: images of zero rows
    a0 = ops.limb_pack(torch.zeros(0, 72, device=dev), "bf16x3")
    b0 = ops.limb_pack(torch.zeros(0, 40, device=dev), "bf16x3")
    out = torch.full((72, 40), 7.0, device=dev)
    cs = torch.full((72,), 7.0, device=dev)
    (w, c), = ag.gemm_tn_limbs_multi([(a0, b0, out, cs)])
    torch.cuda.synchronize()
    assert w.data_ptr() == out.data_ptr() and float(w.abs().max()) == 0.0 and float(c.abs().max()) == 0.0
    # one job = the single entry point
    g = torch.Generator().manual_seed(1)
    a = ops.limb_pack(torch.randn(300, 72, generator=g).to(dev), "bf16x3")
    b = ops.limb_pack(torch.randn(300, 40, generator=g).to(dev), "bf16x3")
    (w1, c1), = ag.gemm_tn_limbs_multi([(a, b, None, None)])
    w0, c0 = ag.gemm_tn_limbs(a, b)
    assert torch.equal(w0, w1) and torch.equal(c0, c1)
    # refused calls
    job = _lib.HdTnJob(ops._p(a.buf), ops._p(b.buf), ops._p(w1), ops._p(c1), 72, 40, 40)
    nine = (_lib.HdTnJob * 9)(*([job] * 9))
    assert _lib.load().hd_gemm_tn_limbs_multi_workspace_floats(300, nine, 9) == -1
    ws = torch.zeros(1 << 20, device=dev)
    for arr, n, wsp in ((nine, 9, ops._p(ws)), (nine, 0, ops._p(ws))):
        with pytest.raises(RuntimeError, match="between 1 and 8 jobs"):
            _lib.call("hd_gemm_tn_limbs_multi_f32", 300, arr, n, wsp, ops._stream())
    with pytest.raises(RuntimeError, match="misaligned workspace"):
        _lib.call("hd_gemm_tn_limbs_multi_f32", 300, nine, 1, ops._p(ws[1:]), ops._stream())
    bad = _lib.HdTnJob(ops._p(a.buf), ops._p(b.buf), ops._p(w1), ops._p(c1), 70, 40, 40)      # P not a multiple of 4
    with pytest.raises(RuntimeError, match="multiples of 4"):
        _lib.call("hd_gemm_tn_limbs_multi_f32", 300, (_lib.HdTnJob * 1)(bad), 1, ops._p(ws), ops._stream())


@pytest.mark.gpu
def test_chain_gang_order_gives_the_same_bits():
    """HD_MLP_CHAIN_GANG hands the row blocks of a queue out in another order (gangs of G, column tile by column tile);
    the order must not change a bit.  The knob is read when the library is first used, so the check runs in a process of
    its own (tools/probes/chain_gang.py ... check: G = 2, 3, 5, 16 against G = 1 at the bench's shapes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "chain_gang.py"), "2,3,5,16", "check"],
                       capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "bit-identical" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_chain_timeout_poisons_its_own_launch_only():
    """The bounded wait of hd_mlp_chain_limbs_f32, exercised: with HD_MLP_CHAIN_TEST_WITHHOLD=1 the tiles of (layer 0, row
    block 0) never report completion, the tiles of layer 1 that need them run into the 2 s bound, the launch ends (no hang)
    with NaN outputs, hd_mlp_chain_failed reports 1 exactly once -- and the NEXT launch on the same workspace, with no query
    in between, is clean and bit-identical to a launch before the failure (the fail word is per launch)."""
    import os
    import time
    from hypdiff_amd import ops
    dev = "cuda:0"
    dims = [72, 200, 136, 68]
    layers = _layers(dims, 5, dev)
    a = ops.limb_pack(torch.randn(4500, dims[0], generator=torch.Generator().manual_seed(2)).to(dev), "bf16x3")
    good = ops.mlp_chain(a, layers)
    torch.cuda.synchronize()
    assert torch.isfinite(good).all() and not ops.mlp_chain_failed()
    os.environ["HD_MLP_CHAIN_TEST_WITHHOLD"] = "1"
    try:
        t0 = time.perf_counter()
        bad = ops.mlp_chain(a, layers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        os.environ.pop("HD_MLP_CHAIN_TEST_WITHHOLD", None)
    assert 1.5 < dt < 30.0, dt                                   # ended by the bound, not by luck and not by a hang
    assert torch.isnan(bad).all()                                # the failure is loud
    again = ops.mlp_chain(a, layers)                             # no query in between
    torch.cuda.synchronize()
    assert torch.equal(again, good)
    assert ops.mlp_chain_failed()                                # reported once ...
    assert not ops.mlp_chain_failed()                            # ... and cleared
    assert torch.equal(ops.mlp_chain(a, layers), good)
