"""GPU: config 5's SpMM paths -- hub rows (spmm_long: more than 512 stored entries, one workgroup
per row, 8 slices summed in slice order) and the non-LDS branch of spmm_rows (a block whose CSR
segment exceeds one 2048-entry stage) -- against the oracle (layers/hyp_layers.py:150-151).

Tolerance.  Rows of at most 512 entries are summed strictly in CSR order = the order the
reference's torch.spmm walks, with the product rounded before the add (no fma contraction): they
must equal the oracle BIT FOR BIT.  A hub row is summed as 8 contiguous slices added in slice
order, so it differs from the sequential sum by rounding only; both are within (deg * 2^-24) *
sum_k |val_k x_k| of the exact sum, and the test holds the kernel to 2^-22 * sum_k |val_k x_k|
against the fp64 sum (4 fp32 ulps of the row's magnitude sum; measured on MI355X: 0.09-0.15 of
that bound at degrees 513 ... 10 005; the test prints the figure)."""
import types

import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import adjacency, layers as olayers

pytestmark = pytest.mark.gpu

N, M_ATTACH = 20000, 5
HUBS = (513, 2049, 10000)
CLUSTER = (1000, 16, 400)           # 16 consecutive rows x 400 entries: > 2048 per block, none of them long


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def H(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def hub_graph():
    from hypdiff_amd import workloads
    adj = workloads.add_hubs(workloads.power_law_graph(N, M_ATTACH), HUBS, cluster=CLUSTER)
    idx, val, shape = adjacency.process_adj(adj)
    deg = np.bincount(idx[0], minlength=N)
    return dict(idx=idx, val=val, shape=shape, deg=deg)


def _exact_and_mag(g, x):
    """fp64 row sums and the per-element magnitude sums sum_k |val_k x_k| (the condition yardstick)"""
    idx, val = torch.from_numpy(g["idx"]), torch.from_numpy(g["val"]).double()
    xd = torch.from_numpy(x).double()
    ex = torch.zeros(N, x.shape[1], dtype=torch.float64).index_add_(0, idx[0], val[:, None] * xd[idx[1]])
    mag = torch.zeros(N, x.shape[1], dtype=torch.float64).index_add_(0, idx[0], (val[:, None] * xd[idx[1]]).abs())
    return ex.numpy(), mag.numpy()


def _check_rows(got, want, exact, mag, deg, what):
    long = deg > 512
    short = ~long
    nbit = int((got[short] != want[short]).sum())
    assert nbit == 0, f"{what}: {nbit} of {got[short].size} elements of rows <= 512 entries differ from the oracle"
    err_long = np.abs(got[long].astype(np.float64) - exact[long])
    bound = 2.0 ** -22 * mag[long] + 1e-30
    worst = float((err_long / bound).max())
    print(f"{what}: rows <= 512 entries bit-identical to the oracle ({got[short].size} elements); "
          f"{int(long.sum())} hub rows: max |err| / (2^-22 sum|val x|) = {worst:.3f}, "
          f"max |kernel - oracle| = {np.abs(got[long] - want[long]).max():.3e}")
    assert worst <= 1.0, f"{what}: hub rows exceed 2^-22 * sum|val x| by x{worst:.2f}"


def test_graph_has_hubs_and_oversized_blocks(hub_graph):
    """the workload really reaches both kernels' special paths"""
    from hypdiff_amd import ops
    g = hub_graph
    csr = ops.csr_from_coo(G(g["idx"]), G(g["val"]), g["shape"])
    assert csr.n_long >= len(HUBS) and int(g["deg"].max()) >= 10000
    assert np.array_equal(np.sort(H(csr.long_rows)), np.flatnonzero(g["deg"] > 512))
    rp = H(csr.rowptr).astype(np.int64)
    for rows_per_block in (4, 16, 64):          # lane groups of 64 / 16 / 4 per row (csrc/hd_common.cuh)
        seg = rp[np.minimum(np.arange(0, N, rows_per_block) + rows_per_block, N)] - rp[np.arange(0, N, rows_per_block)]
        assert (seg > 2048).any(), rows_per_block
    # 16 rows per block: the cluster's block is oversized although none of its rows is long
    b = CLUSTER[0] // 16
    assert rp[(b + 1) * 16] - rp[b * 16] > 2048 and (g["deg"][b * 16:(b + 1) * 16] <= 512).all()


@pytest.mark.parametrize("dt", ["f32", "f64"])
@pytest.mark.parametrize("d", [64, 256, 13])
def test_spmm_hub_rows_vs_oracle(hub_graph, d, dt):
    from hypdiff_amd import ops
    g = hub_graph
    tdt = torch.float32 if dt == "f32" else torch.float64
    rs = np.random.RandomState(d)
    x = rs.standard_normal((N, d)).astype(np.float32 if dt == "f32" else np.float64)
    csr = ops.csr_from_coo(G(g["idx"]), G(g["val"]), g["shape"])
    got = H(ops.spmm(csr, G(x)))
    want = olayers.spmm_coo(torch.from_numpy(g["idx"]), torch.from_numpy(g["val"]), g["shape"], torch.from_numpy(x)).numpy()
    assert got.dtype == want.dtype
    ex, mag = _exact_and_mag(g, x)
    if dt == "f32":
        _check_rows(got, want, ex, mag, g["deg"], f"spmm d={d}")
    else:
        assert_close(got, want, rtol=1e-12, atol_scale=1e-13, what=f"spmm f64 d={d}")
    # backward of torch.spmm w.r.t. the dense operand: adj^T @ g through the transposed CSR
    gt = ops.spmm(csr.transposed(), G(x))
    order = np.argsort(g["idx"][1], kind="stable")
    idx_t = np.vstack([g["idx"][1][order], g["idx"][0][order]])
    want_t = olayers.spmm_coo(torch.from_numpy(idx_t), torch.from_numpy(g["val"][order]), g["shape"],
                              torch.from_numpy(x)).numpy()
    assert csr.transposed().n_long > 0
    if dt == "f32":
        gT = dict(idx=idx_t, val=g["val"][order])
        exT, magT = _exact_and_mag(gT, x)
        _check_rows(H(gt), want_t, exT, magT, np.bincount(idx_t[0], minlength=N), f"spmm^T d={d}")
    else:
        assert_close(H(gt), want_t, rtol=1e-12, atol_scale=1e-13, what=f"spmm^T f64 d={d}")


@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
@pytest.mark.parametrize("d", [64, 256])
def test_agg_hub_rows_vs_oracle(hub_graph, mname, c, d):
    """hd_pb_agg / hd_hb_agg (expmap0 -> proj [-> HypAct] fused into the SpMM write-out), fuse_act 0 / 1,
    f32 and f64.  Tangent inputs small enough that tanh / sinh keep the aggregate's rounding visible."""
    from hypdiff_amd import ops
    from oracle import manifolds as om
    g = hub_graph
    pre = "pb_" if mname == "PoincareBall" else "hb_"
    abi = "pb" if mname == "PoincareBall" else "hb"
    rs = np.random.RandomState(d + 1)
    adj = (torch.from_numpy(g["idx"]), torch.from_numpy(g["val"]), g["shape"])
    for dt in (np.float32, np.float64):
        xt = (rs.standard_normal((N, d)) * 0.3).astype(dt)
        if abi == "hb":
            xt[:, 0] = 0
        csr = ops.csr_from_coo(G(g["idx"]), G(g["val"]), g["shape"])
        sup = olayers.spmm_coo(adj[0], adj[1], adj[2], torch.from_numpy(xt))
        want = getattr(om, pre + "proj")(getattr(om, pre + "expmap0")(sup, c), c)
        tol = dict(rtol=1e-5, atol_scale=2e-6) if dt == np.float32 else dict(rtol=1e-10, atol_scale=1e-11)
        got = ops.agg(abi, csr, G(xt), c)
        assert_close(H(got), want.numpy(), what=f"{abi}_agg d={d} {dt.__name__}", **tol)
        got = ops.agg(abi, csr, G(xt), c, 1.0, 1)
        want2 = olayers.hyp_act(mname, want, c, 1.0)
        assert_close(H(got), want2.numpy(), what=f"{abi}_agg+act d={d} {dt.__name__}", **tol)


@pytest.mark.parametrize("mname,c", [("PoincareBall", 0.6), ("Hyperboloid", 1.0)])
def test_sharded_encode_with_hubs(hub_graph, mname, c):
    """dist.sharded_hgcn_encode on the hub graph: world 1 through the function, world 2 / 3 simulated
    in-process (all-gather = concatenation of the padded blocks; the collective itself runs in
    tests/test_dist_gloo.py and tests/test_dist_nccl.py).  Shard rows keep the 1-GPU kernel's per-row
    summation order and the whole graph's form of the fused tail -> bit-identical to the unsharded encode (also
    where a shard alone would pick the other form: the Hyperboloid graph has long rows, a world-3 shard may have none);
    the unsharded encode against the oracle."""
    from hypdiff_amd import dist as hdist, ops
    from hypdiff_amd.models.encoders import HGCN, _lift
    g = hub_graph
    feat, dim = 64, 256
    rs = np.random.RandomState(9)
    x = (rs.standard_normal((N, feat)) * 0.1).astype(np.float32)
    if mname == "Hyperboloid":
        x = np.concatenate([np.zeros((N, 1), np.float32), x], 1)
    args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device="cuda:0", feat_dim=x.shape[1], dim=dim,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    torch.manual_seed(4)
    enc = HGCN(torch.tensor([c]).cuda(), args).cuda().eval()
    adj = torch.sparse_coo_tensor(G(g["idx"]), G(g["val"]), g["shape"], check_invariants=False)
    with torch.no_grad():
        full = enc.encode(G(x), adj)
        full_csr = ops.csr_of(adj)
        Ws = [l.linear.weight.detach().cpu() for l in enc.layers]
        ref = olayers.hgcn_encode(mname, torch.from_numpy(x), (torch.from_numpy(g["idx"]), torch.from_numpy(g["val"]),
                                                                g["shape"]), Ws, [c, c, c], prepend_time=False)
        assert_close(H(full), ref.numpy(), rtol=1e-5, atol_scale=2e-6, what=f"hub-graph encode {mname}")
        one = hdist.sharded_hgcn_encode(enc, G(x), hdist.RowShardedAdj(g["idx"], g["val"], g["shape"], 0, 1))
        assert torch.equal(one, full)
        for world in (2, 3):
            shards = [hdist.RowShardedAdj(g["idx"], g["val"], g["shape"], r, world) for r in range(world)]
            assert sum(hdist.shard_csr(s, "cuda").n_long for s in shards) >= len(HUBS)
            hs = [_lift(enc.manifold, G(x[s.lo:s.hi]), enc.curvatures[0]) for s in shards]
            for layer in enc.layers:
                xts = [layer.linear.forward(h, _tangent=True) for h in hs]
                gathered = torch.cat([torch.cat([xt, xt.new_zeros(s.rows_per_rank - xt.shape[0], xt.shape[1])])
                                      for s, xt in zip(shards, xts)])
                hs = [hdist.shard_aggregate(layer, hdist.shard_csr(s, "cuda"), gathered, s.hi - s.lo) for s in shards]
            # Shards keep the 1-GPU kernel's per-row summation order AND run the form of the fused tail (one / four rows
            # per wavefront, csrc/hd_spmm.cuh) that the whole graph takes (dist.shard_csr -> csr.whole): bit-identical.
            assert all(hdist.shard_csr(s, "cuda").whole == (N, full_csr.n_long) for s in shards)
            assert torch.equal(torch.cat(hs), full), world


def test_config5_full_size_properties():
    """Config 5 at its own size -- 10^6 nodes, m = 5 (nnz ~ 1.1e7 with self loops, natural hubs of ~10^4
    entries), dim 256 -- through size-independent properties: A @ 1 == 1 for the row-normalised
    adjacency (every row's weights sum to one), linearity, and agreement of the fused aggregate with
    the unfused composition."""
    from hypdiff_amd import ops, workloads
    import hypdiff_amd.manifolds as hm
    n, d = 1_000_000, 256
    adj = workloads.power_law_graph(n, 5)
    idx, val, shape = adjacency.process_adj(adj)
    csr = ops.csr_from_coo(G(idx), G(val), shape)
    assert csr.n_long > 0 and int((csr.rowptr[1:] - csr.rowptr[:-1]).max()) > 2048
    ones = torch.ones(n, d, device="cuda")
    out = ops.spmm(csr, ones)
    # a row is the fp32 sum of deg copies of fp32(1/deg): within (deg + 2) * 2^-24 of one
    deg = (csr.rowptr[1:] - csr.rowptr[:-1]).to(torch.float32)
    assert bool(((out - 1).abs().amax(dim=1) <= (deg + 2) * 2.0 ** -24).all())
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, d, device="cuda", generator=gen)
    y = torch.randn(n, d, device="cuda", generator=gen)
    ax, ay = ops.spmm(csr, x), ops.spmm(csr, y)
    lhs = ops.spmm(csr, 2 * x - 3 * y)
    err = (lhs - (2 * ax - 3 * ay)).abs().max()
    assert float(err) <= 1e-5, float(err)
    pb = hm.PoincareBall()
    xt = x * 0.05
    fused = ops.agg("pb", csr, xt, 0.6, 1.0, 1)
    sup = ops.spmm(csr, xt)
    unf = ops.hyp_act("pb", pb.proj(pb.expmap0(sup, 0.6), 0.6), 0.6, 1.0, 1)
    assert_close(H(fused[::997]), H(unf[::997]), rtol=1e-5, atol_scale=1e-6, what="fused vs unfused aggregate")
    assert bool(torch.isfinite(fused).all())
