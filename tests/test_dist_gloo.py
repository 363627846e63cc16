"""CPU, world_size = 2 over gloo: the multi-rank host logic of hypdiff_amd/dist.py against
the single-process result (the reference has no multi-GPU behaviour to compare with)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import adjacency, diffusion as odiff, golden_inputs as gi, layers as olayers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hypdiff_amd import dist as hdist
    torch.manual_seed(0)
    res = {}
    # (1) sharded normalization == global normalization
    x = torch.randn(101, 16, dtype=torch.float64)
    lo, hi = hdist.shard_range(101, rank, world)
    xs = x[lo:hi]
    m = torch.tensor([xs.sum().item(), (xs * xs).sum().item(), float(xs.numel())], dtype=torch.float64)
    hdist.allreduce_moments(m)
    mean = m[0] / m[2]
    sd = torch.sqrt((m[1] - m[0] * m[0] / m[2]) / (m[2] - 1))
    res["norm"] = torch.allclose((xs - mean) / sd, odiff.normalization(x)[lo:hi], rtol=1e-12, atol=1e-12)
    # (2) flat gradient bucket: mean over ranks of per-shard gradients == full-batch gradient
    lin = torch.nn.Linear(8, 4)
    torch.manual_seed(1)
    for p in lin.parameters():
        torch.nn.init.normal_(p)
    nb = 40 if world == 2 else 14 * world          # equal shards: the mean of the shard means is the batch mean
    data = torch.randn(nb, 8)
    bucket = hdist.GradBucket(lin.parameters())
    lo, hi = hdist.shard_range(nb, rank, world)
    lin(data[lo:hi]).pow(2).mean().backward()
    bucket.allreduce_mean_()
    ref = torch.nn.Linear(8, 4)
    ref.load_state_dict(lin.state_dict())
    ref(data).pow(2).mean().backward()
    res["grad"] = all(torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
                      for a, b in zip(lin.parameters(), ref.parameters()))
    # (3) row-sharded aggregation with tangent all-gather == full SpMM, bit for bit
    n, d = 203, 12
    adj = gi.synthetic_graph("ba", n)
    idx, val, shape = adjacency.process_adj(adj)
    xt = torch.from_numpy(np.random.RandomState(5).standard_normal((n, d)).astype(np.float32))
    full = olayers.spmm_coo(torch.from_numpy(idx), torch.from_numpy(val), shape, xt)
    sh = hdist.RowShardedAdj(idx, val, shape, rank, world)
    spmm = lambda i, v, s, xx: olayers.spmm_coo(torch.from_numpy(i), torch.from_numpy(v), s, xx)
    out = sh.aggregate(xt[sh.lo:sh.hi], spmm)
    res["agg"] = torch.equal(out, full[sh.lo:sh.hi])
    # ... and with the all-gather pipelined in column slabs (asynchronous collectives, one wait per slab)
    out = sh.aggregate_slabs(xt[sh.lo:sh.hi], spmm, n_slabs=3)
    res["agg_slabs"] = torch.equal(out, full[sh.lo:sh.hi])
    # (3b) the halo form on a graph with locality: only the referenced remote rows travel (all_to_all_single with the
    #      precomputed lists), the block is re-indexed, the result is the same bits
    n2 = 611
    idx2, val2, shape2 = adjacency.process_adj(gi.synthetic_graph("ring", n2))
    xt2 = torch.from_numpy(np.random.RandomState(6).standard_normal((n2, d)).astype(np.float32))
    full2 = olayers.spmm_coo(torch.from_numpy(idx2), torch.from_numpy(val2), shape2, xt2)
    sh2 = hdist.RowShardedAdj(idx2, val2, shape2, rank, world)
    res["halo_chosen"] = sh2.use_halo and 0 < sh2.halo_fraction < 0.5 and not sh.use_halo
    res["halo_volume"] = sh2.halo_bytes(d) < 0.2 * (world - 1) * sh2.rows_per_rank * d * 4
    res["halo_agg"] = torch.equal(sh2.aggregate(xt2[sh2.lo:sh2.hi], spmm), full2[sh2.lo:sh2.hi])
    res["halo_agg_slabs"] = torch.equal(sh2.aggregate_slabs(xt2[sh2.lo:sh2.hi], spmm, n_slabs=3), full2[sh2.lo:sh2.hi])
    # (4) DataParallelTrainer: 3 Adam steps on equal shards == 3 full-batch Adam steps; ranks start
    #     from different weights (rank 0's are broadcast) and the gradients stay bucket views
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))
    tr = hdist.DataParallelTrainer(net, lr=1e-2)
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    xb, yb = torch.randn(nb, 8, generator=torch.Generator().manual_seed(3)), torch.randn(
        nb, 8, generator=torch.Generator().manual_seed(4))
    lo, hi = hdist.shard_range(nb, rank, world)
    for _ in range(3):
        tr.step(lambda: torch.nn.functional.mse_loss(net(xb[lo:hi]), yb[lo:hi]))
        opt.zero_grad()
        torch.nn.functional.mse_loss(ref(xb), yb).backward()
        opt.step()
    res["trainer"] = all(torch.allclose(a, b, rtol=1e-5, atol=1e-6) for a, b in zip(net.parameters(), ref.parameters()))
    flat = tr.bucket.flat
    res["views"] = all(flat.data_ptr() <= p_.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
                       for p_ in net.parameters())
    # the same with the bucket cut into several chunks (sent from post-accumulate hooks while backward is
    # still running) and with the single blocking all-reduce: identical weights either way
    outs = []
    for overlap, chunk in ((True, 64), (False, 4 << 20)):
        torch.manual_seed(7)
        n2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 8))
        t2 = hdist.DataParallelTrainer(n2, lr=1e-2, overlap=overlap)
        if overlap:
            t2.bucket = hdist.GradBucket(n2.parameters(), chunk_bytes=chunk)
            t2.opt = torch.optim.Adam(t2.bucket.params, lr=1e-2)
            res["chunks"] = len(t2.bucket.chunks) >= 3
        for _ in range(3):
            t2.step(lambda: torch.nn.functional.mse_loss(n2(xb[lo:hi]), yb[lo:hi]))
        outs.append([p_.detach().clone() for p_ in n2.parameters()])
    # (two ranks: a + b either way, the same bits; with more, a ring all-reduce adds a chunk's elements in an order that
    # depends on where the chunk starts)
    same = torch.equal if world == 2 else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-7))
    res["overlap_equals_blocking"] = all(same(a, b) for a, b in zip(*outs))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res.values()), (rank, [k for k, v in res.items() if not v])


def test_world3_gloo():
    """three ranks: the last block is shorter than the others (padding rows), a rank has two peers to exchange with"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(3)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res.values()), (rank, [k for k, v in res.items() if not v])


def test_shard_range_covers():
    from hypdiff_amd.dist import shard_range
    for n in (0, 1, 7, 1197):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
