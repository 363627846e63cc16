"""GPU: the multi-GPU host logic of hypdiff_amd/dist.py with backend "nccl" (= RCCL) on the device.

The GPU box has one GPU, so the process group is a world of ONE rank with
dist.FORCE_COLLECTIVES set: every collective of the path is really issued to RCCL (all-reduce of the
moments, chunked asynchronous all-reduce of the gradient bucket fed by the HIP backward kernels,
all-gather of the tangent rows, whole and pipelined in column slabs) and must leave the single-GPU
results unchanged.  World sizes > 1 run over gloo in tests/test_dist_gloo.py (host logic) and
in-process in tests/test_spmm_hubs.py / test_gpu_parity.py (device kernels on shard CSRs)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    res = {}
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from hypdiff_amd import dist as hdist, workloads
        from hypdiff_amd.models.direction_diffusion import diffusion
        from hypdiff_amd.models.encoders import HGCN
        from oracle import adjacency
        hdist.FORCE_COLLECTIVES = True
        dev = "cuda:0"
        # (1) moments all-reduce through diffusion.moments (sync_moments)
        torch.manual_seed(0)
        dim, n = 64, 1000
        diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=200)
        assert all(p.device == torch.device(dev) for p in diff.parameters()) and diff.sqrt_alphas_cumprod.device == torch.device(dev)
        x = torch.randn(n, dim, device=dev)
        m0 = diff.moments(x).clone()
        diff.sync_moments = True
        m1 = diff.moments(x)
        res["moments"] = bool(torch.equal(m0, m1))
        # (2) DataParallelTrainer on the mlp denoiser: chunked overlapped all-reduce fed by the HIP backward
        #     kernels (GRAD_SINK) == plain backward + Adam without any collective
        rs = np.random.RandomState(1)
        emb = rs.standard_normal((n, dim)).astype(np.float32)
        emb *= 0.5 / np.linalg.norm(emb, axis=1, keepdims=True)
        labels = rs.randint(0, 3, size=n)
        u0 = torch.from_numpy((rs.standard_normal((3, dim)) * 0.05).astype(np.float32)).to(dev)
        t = torch.from_numpy(rs.randint(0, 200, size=n)).to(dev)
        noise = torch.from_numpy(rs.standard_normal((n, dim))).to(dev)
        import copy
        ref_mlp = copy.deepcopy(diff.mlp)
        ref2_mlp = copy.deepcopy(diff.mlp)
        tr = hdist.DataParallelTrainer(diff.mlp, lr=1e-3)
        tr.opt = torch.optim.Adam(tr.bucket.params, lr=1e-3)        # torch's Adam: the gradient path is bit-checked
        res["chunks"] = len(tr.bucket.chunks)
        loss_fn = lambda: diff(torch.from_numpy(emb).to(dev), None, labels, u0, restrict=True, graphset=True, t=t,
                               noise=noise.clone())
        diff.train()
        losses = [float(tr.step(loss_fn)) for _ in range(3)]
        mine = [p.detach().clone() for p in diff.mlp.parameters()]
        hdist.FORCE_COLLECTIVES = False
        diff.mlp = ref_mlp
        opt = torch.optim.Adam(ref_mlp.parameters(), lr=1e-3)
        ref_losses = []
        for _ in range(3):
            opt.zero_grad()
            l = loss_fn()
            l.backward()
            opt.step()
            ref_losses.append(float(l))
        res["trainer"] = all(torch.equal(a, b.detach()) for a, b in zip(mine, ref_mlp.parameters()))
        res["losses"] = losses == ref_losses and losses[2] < losses[0]
        hdist.FORCE_COLLECTIVES = True
        # (2b) the default optimizer (BucketAdam, one launch): same trajectory as torch's Adam to rounding
        diff.mlp = ref2_mlp
        tr2 = hdist.DataParallelTrainer(ref2_mlp, lr=1e-3)
        res["bucket_adam"] = isinstance(tr2.opt, hdist.BucketAdam)
        v0 = [p._version for p in ref2_mlp.parameters()]
        losses2 = [float(tr2.step(loss_fn)) for _ in range(3)]
        res["bucket_adam_versions"] = all(p._version > v for p, v in zip(ref2_mlp.parameters(), v0))
        res["bucket_adam_close"] = all(torch.allclose(a.detach(), b, rtol=1e-5, atol=2e-7) for a, b in
                                       zip(ref2_mlp.parameters(), mine)) and \
            np.allclose(losses2, ref_losses, rtol=1e-5)
        # (3) row-sharded encode: all-gather (whole, and pipelined in 4 column slabs) == unsharded encode
        g_n, feat, d = 3000, 16, 64
        adj = workloads.add_hubs(workloads.power_law_graph(g_n, 4), (600,))
        idx, val, shape = adjacency.process_adj(adj)
        xf = torch.from_numpy((rs.standard_normal((g_n, feat)) * 0.2).astype(np.float32)).to(dev)
        for mname, c in (("PoincareBall", 0.6), ("Hyperboloid", 1.0)):
            xin = torch.cat([torch.zeros_like(xf[:, :1]), xf], 1) if mname == "Hyperboloid" else xf
            args = types.SimpleNamespace(manifold=mname, c=c, cuda=0, device=dev, feat_dim=xin.shape[1], dim=d,
                                         num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
            enc = HGCN(torch.tensor([c], device=dev), args).to(dev).eval()
            at = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), shape,
                                         check_invariants=False)
            with torch.no_grad():
                full = enc.encode(xin, at)
                sh = hdist.RowShardedAdj(idx, val, shape, 0, 1)
                whole = hdist.sharded_hgcn_encode(enc, xin, sh, n_slabs=1)
                slabs = hdist.sharded_hgcn_encode(enc, xin, sh, n_slabs=4)
            res["gather_" + mname] = bool(torch.equal(whole, full))
            res["slabs_" + mname] = bool(torch.equal(slabs, full))
        # (4) the whole training step WITH its gradient all-reduces as one HIP graph (DataParallelTrainer.capture):
        #     replays == eager steps, bit for bit
        try:
            torch.manual_seed(5)
            import copy as _copy
            from hypdiff_amd.models.direction_diffusion import diffusion as _diffusion
            dg = _diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=200).train()
            de = _diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=200).train()
            de.mlp.load_state_dict(dg.mlp.state_dict())
            emb_d = torch.from_numpy(emb).to(dev)
            lab_d = labels
            mk = lambda dd: (lambda: dd(emb_d, None, lab_d, u0, restrict=True, graphset=True, t=t, noise=noise.clone()))
            trg = hdist.DataParallelTrainer(dg.mlp, lr=1e-3)
            tre = hdist.DataParallelTrainer(de.mlp, lr=1e-3)
            replay = trg.capture(mk(dg), warmup=2)
            for _ in range(3):
                replay()
            for _ in range(2 + 3):
                tre.step(mk(de))
            torch.cuda.synchronize()
            res["graphed_step_with_collectives"] = all(torch.equal(a.detach(), b.detach())
                                                       for a, b in zip(dg.mlp.parameters(), de.mlp.parameters()))
        except Exception:
            import traceback
            res["graphed_step_with_collectives_error"] = traceback.format_exc()[-1500:]
        torch.cuda.synchronize()
        dist.destroy_process_group()
    except Exception as e:          # report instead of hanging the parent
        import traceback
        res["error"] = traceback.format_exc()
    q.put(res)


def test_rccl_world_of_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert "error" not in res, res["error"]
    assert p.exitcode == 0
    assert res.pop("chunks") >= 1
    assert "graphed_step_with_collectives_error" not in res, res["graphed_step_with_collectives_error"]
    assert all(res.values()), res
