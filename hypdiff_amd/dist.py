"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests).  The reference is single-device
(models/direction_diffusion.py:87); correctness here means equality with the 1-GPU result.

Three shardings:
  * graph sets (configs 1-3): whole graphs per rank, no data-path collective.  The only
    global quantity of the diffusion path is the mean / unbiased std of `normalization`
    (models/direction_diffusion.py:145-153): all-reduce the shard-local (sum, sumsq, count).
  * data-parallel training: one flat fp32 gradient bucket, one all-reduce per step.
  * one large graph (configs 4-5): contiguous row blocks of the CSR; HypLinear / HypAct are
    row-local, HypAgg needs the tangent rows of remote columns -> one all-gather of the
    (N/P, d) tangent block per layer, then the local SpMM over global column indices.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced [lo, hi) of n_items for `rank` (first n % P ranks get one more)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_moments(m, group=None):
    """Sum the (sum, sumsq, count) triple of diffusion.moments() over ranks, in place:
    24 bytes on the wire; every rank then standardises with the global mean/std and
    obtains exactly what a single GPU would compute on the concatenated rows (up to the
    fp64 summation order)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
    return m


class GradBucket:
    """All gradients of a module as ONE flat fp32 buffer (2.8 M elements for the mlp
    denoiser at D=512 ~ 11 MB): a single ring all-reduce per step instead of one per
    parameter -- xGMI links are point-to-point, so few large messages beat many small."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:       # gradients become views into the bucket: no pack/unpack copies
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def rebind(self):
        """autograd accumulates into an existing .grad in place, which keeps the views; if a
        gradient was replaced by a fresh tensor anyway, copy it back into the bucket."""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    def allreduce_mean_(self, group=None, async_op=False):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        self.flat.div_(dist.get_world_size(group))
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class DataParallelTrainer:
    """Data-parallel training step of a denoiser (diff.py:113-118 / diff_graphset.py:64-75 run
    loss -> backward -> Adam on one device; here every rank does that on its own shard of the
    graph set and the gradients are averaged in between).

    All gradients live in one flat fp32 bucket (the HIP backward kernels' outputs are
    accumulated into views of it), so the only collective of a step is ONE all-reduce of
    ~11 MB (mlp, D=512) / ~25 MB (ddpm.Backbone) over RCCL; the optimizer is torch's Adam on
    every rank (identical updates: weights never need a broadcast after the first)."""

    def __init__(self, module, lr=1e-3, group=None, optimizer=None, broadcast=True):
        self.module, self.group = module, group
        self.bucket = GradBucket(module.parameters())
        self.opt = optimizer if optimizer is not None else torch.optim.Adam(self.bucket.params, lr=lr)
        if broadcast and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            with torch.no_grad():
                for p in self.bucket.params:      # rank 0's initialisation everywhere
                    tmp = p.detach().clone()
                    dist.broadcast(tmp, src=src, group=group)
                    p.copy_(tmp)                  # in place on p itself: bumps p._version, so the caches keyed
                                                  # on it (limb images, padded / transposed copies) are rebuilt

    def step(self, loss_fn):
        """loss_fn() -> scalar loss of this rank's shard.  Returns the (local) loss tensor."""
        self.bucket.zero_()
        loss = loss_fn()
        loss.backward()
        self.bucket.rebind()
        self.bucket.allreduce_mean_(self.group)
        self.opt.step()
        return loss.detach()


class RowShardedAdj:
    """This rank's contiguous row block of a row-sorted COO adjacency, columns global.

    indices: int64 (2, nnz) numpy or tensor (global), values: (nnz,), shape (N, N).
    All ranks hold blocks of `rows_per_rank` rows (the last block is padded with empty
    rows) so the tangent all-gather is one equal-sized collective."""

    def __init__(self, indices, values, shape, rank, world_size):
        indices = np.asarray(indices)
        values = np.asarray(values)
        n = int(shape[0])
        self.n, self.rank, self.world = n, rank, world_size
        self.rows_per_rank = (n + world_size - 1) // world_size
        self.lo = min(n, rank * self.rows_per_rank)
        self.hi = min(n, self.lo + self.rows_per_rank)
        a = np.searchsorted(indices[0], self.lo, side="left")
        b = np.searchsorted(indices[0], self.hi, side="left")
        self.local_indices = np.vstack((indices[0, a:b] - self.lo, indices[1, a:b])).astype(np.int64)
        self.local_values = values[a:b]
        self.local_shape = (self.rows_per_rank, self.rows_per_rank * world_size)

    def gather_rows(self, x_local, group=None):
        """(rows_per_rank, d) block of every rank -> (rows_per_rank * P, d), rank-major = global
        row order.  Direct all-gather: on the full xGMI mesh every rank writes its block to
        its 7 peers concurrently."""
        if x_local.shape[0] != self.rows_per_rank:
            pad = torch.zeros(self.rows_per_rank - x_local.shape[0], x_local.shape[1], dtype=x_local.dtype,
                              device=x_local.device)
            x_local = torch.cat([x_local, pad], 0)
        if self.world == 1:
            return x_local
        out = torch.empty(self.rows_per_rank * self.world, x_local.shape[1], dtype=x_local.dtype,
                          device=x_local.device)
        dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
        return out

    def aggregate(self, xt_local, spmm, group=None):
        """HypAgg's support_t for the local rows: all-gather the tangent block, then
        spmm(local_adj, gathered).  `spmm(indices, values, shape, x)` is the device SpMM
        (ops.spmm on a CSR of the local block) -- injected so that the CPU tests can pass a
        host implementation.  Summation order per row is unchanged -> bit-identical to 1 GPU."""
        full = self.gather_rows(xt_local, group)
        out = spmm(self.local_indices, self.local_values, self.local_shape, full)
        return out[: self.hi - self.lo]


def sharded_hgcn_encode(encoder, x_local, sharded_adj, group=None):
    """HGCN.encode (models/encoders.py:115-119) for ONE graph whose rows are split over the ranks
    (config 5).  The input lift, HypLinear and HypAct are row-local; HypAgg needs the tangent rows
    of remote columns, so every layer is

        local rows:  lift / HypLinear tail -> tangent block (N/P, d)
        all ranks :  ONE all-gather of the tangent blocks (RCCL over xGMI)
        local rows:  CSR SpMM over global columns with expmap0 / proj / HypAct fused

    `x_local` holds this rank's rows of the features (Hyperboloid: zero time column already
    prepended).  Per row the summation order is that of the 1-GPU kernel, so the result is
    bit-identical to encoder.encode on the whole graph.  Inference path (torch.no_grad())."""
    from .models.encoders import _lift
    n_loc = sharded_adj.hi - sharded_adj.lo
    assert x_local.shape[0] == n_loc, (x_local.shape, n_loc)
    shard_csr(sharded_adj, x_local.device)
    csr = sharded_adj._csr
    h = _lift(encoder.manifold, x_local, encoder.curvatures[0])
    for layer in encoder.layers:
        xt = layer.linear.forward(h, _tangent=True)
        h = shard_aggregate(layer, csr, sharded_adj.gather_rows(xt, group), n_loc)
    return h


def shard_csr(sharded_adj, device):
    """device CSR of a rank's row block (columns global), built once"""
    from . import ops
    if not hasattr(sharded_adj, "_csr"):
        sharded_adj._csr = ops.csr_from_coo(torch.from_numpy(sharded_adj.local_indices).to(device),
                                            torch.from_numpy(np.ascontiguousarray(sharded_adj.local_values)).to(device),
                                            sharded_adj.local_shape)
    return sharded_adj._csr


def shard_aggregate(layer, csr, xt_full, n_loc):
    """HypAgg + HypAct of one HyperbolicGraphConvolution on a row block: `xt_full` = the
    all-gathered tangent rows of every rank, result = this rank's n_loc output rows"""
    from . import ops
    from .layers.hyp_layers import act_code
    code = act_code(layer.hyp_act.act)
    man = layer.agg.manifold
    if code is not None:
        out = ops.agg(man.abi, csr, xt_full, layer.agg.c, layer.hyp_act.c_out, code)
    else:
        out = layer.hyp_act.forward(ops.agg(man.abi, csr, xt_full, layer.agg.c))
    return out[:n_loc]
