"""One rank of tests/test_two_ranks_one_gpu.py (not collected by pytest): both ranks run on cuda:0, the process group is gloo
with every collective staged through the host (bench.stage_collectives_through_host), the kernels are the library's."""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    bench.stage_collectives_through_host(dist)
    torch.cuda.set_device(0)
    dev = "cuda:0"
    from hypdiff_amd import dist as hdist, workloads
    from hypdiff_amd.models.direction_diffusion import diffusion
    from hypdiff_amd.models.encoders import HGCN
    res = {}

    # (1) data-parallel training of the mlp denoiser: gradients written into the bucket by the backward kernels, chunks
    #     all-reduced from the backward pass, moments all-reduced -- against ONE process on the concatenated rows
    dim, n, T = 64, 4224, 1000                     # >= 4096 rows per rank: the one-launch chains and the limb kernels run
    rs = np.random.RandomState(11)
    emb = rs.standard_normal((world * n, dim)).astype(np.float32)
    emb *= 0.5 / np.linalg.norm(emb, axis=1, keepdims=True)
    emb = torch.from_numpy(emb).to(dev)
    labels = rs.randint(0, 3, size=world * n).astype(np.int64)
    u0 = torch.from_numpy((rs.standard_normal((3, dim)) * 0.05).astype(np.float32)).to(dev)
    g = torch.Generator().manual_seed(5)
    t_all = torch.randint(0, T, (world * n,), generator=g).to(dev)
    noise_all = torch.randn(world * n, dim, generator=g, dtype=torch.float64).to(dev)
    torch.manual_seed(100 + rank)                  # different weights per rank: the trainer broadcasts rank 0's
    diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T).to(dev).train()
    diff.sync_moments = True
    tr = hdist.DataParallelTrainer(diff.mlp, lr=1e-3)
    start = {k: v.detach().clone() for k, v in diff.mlp.state_dict().items()}
    lo, hi = rank * n, (rank + 1) * n

    def shard_loss():
        return diff(emb[lo:hi], None, labels[lo:hi], u0, restrict=True, graphset=True, t=t_all[lo:hi],
                    noise=noise_all[lo:hi].clone())

    loss = tr.step(shard_loss)
    g_dp = tr.bucket.flat.detach().clone()
    res["chunks"] = len(tr.bucket.chunks)
    torch.manual_seed(1)
    ref = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T).to(dev).train()
    ref.mlp.load_state_dict(start)
    full_loss = ref(emb, None, labels, u0, restrict=True, graphset=True, t=t_all, noise=noise_all.clone())
    full_loss.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in ref.mlp.parameters() if p.requires_grad])
    losses = [torch.zeros(1, device=dev) for _ in range(world)]
    dist.all_gather(losses, loss.reshape(1).float())
    res["loss_mean_of_shards_vs_full"] = abs(float(sum(losses) / world) - float(full_loss)) / abs(float(full_loss))
    res["grad_rel_err"] = float((g_dp - g_ref).abs().max() / g_ref.abs().max())
    res["grad_nonzero"] = bool(g_ref.abs().max() > 0)
    for _ in range(2):
        tr.step(shard_loss)
    flat = torch.cat([p.detach().reshape(-1) for p in diff.mlp.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    res["params_identical_across_ranks"] = bool(all(torch.equal(both[0], b) for b in both[1:]))
    res["params_finite"] = bool(torch.isfinite(flat).all())

    # (2) row-sharded HGCN encode (config 5's path) through real collectives: all-gather form on a power-law graph, halo
    #     form on a ring; each against the 1-GPU encode of the whole graph, bit for bit
    from oracle import adjacency, golden_inputs as gi
    for name, n2 in (("ba", 6000), ("ring", 6000)):
        if name == "ba":
            big = workloads.normalize_adj(workloads.power_law_graph(n2, 5, seed=3)).tocoo()
            idx, val, shape = np.vstack((big.row, big.col)).astype(np.int64), big.data.astype(np.float32), (n2, n2)
        else:
            idx, val, shape = adjacency.process_adj(gi.synthetic_graph("ring", n2))
        d = 32
        x = torch.from_numpy((np.random.RandomState(9).standard_normal((n2, d)) * 0.1).astype(np.float32)).to(dev)
        args = types.SimpleNamespace(manifold="PoincareBall", c=0.6, cuda=0, device=dev, feat_dim=d, dim=d, num_layers=2,
                                     act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
        torch.manual_seed(3)
        enc = HGCN(torch.tensor([0.6], device=dev), args).to(dev).eval()
        adj = torch.sparse_coo_tensor(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), shape,
                                      check_invariants=False)
        sh = hdist.RowShardedAdj(idx, val, shape, rank, world)
        with torch.no_grad():
            full = enc.encode(x, adj)
            for slabs in (1, 4):
                got = hdist.sharded_hgcn_encode(enc, x[sh.lo:sh.hi].contiguous(), sh, n_slabs=slabs)
                res[f"{name}_slabs{slabs}_equal"] = bool(torch.equal(got, full[sh.lo:sh.hi]))
        res[f"{name}_halo_form"] = bool(sh.use_halo)
    torch.cuda.synchronize()
    dist.barrier()
    print("RESULT " + json.dumps({"rank": rank, **res}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
