"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPUs, "gloo" in the CPU tests).  The reference is single-device
(models/direction_diffusion.py:87); correctness here means equality with the 1-GPU result.

Three shardings:
  * graph sets (configs 1-3): whole graphs per rank, no data-path collective.  The only
    global quantity of the diffusion path is the mean / unbiased std of `normalization`
    (models/direction_diffusion.py:145-153): all-reduce the shard-local (sum, sumsq, count).
  * data-parallel training: one flat fp32 gradient bucket, one all-reduce per step.
  * one large graph (configs 4-5): contiguous row blocks of the CSR; HypLinear / HypAct are
    row-local, HypAgg needs the tangent rows of remote columns -> the (N/P, d) tangent block of
    every rank is all-gathered per layer IN COLUMN SLABS on RCCL's stream while the local SpMM
    (global column indices) already runs on the slabs that have landed.

Overlap: both collectives run asynchronously (torch's NCCL work handles): the gradient bucket is
reduced in chunks as soon as the backward kernels have produced them (last layers first) while the
GEMMs of the earlier layers still run; the tangent all-gather of slab k+1 is on the wire while
slab k is aggregated.
"""
import numpy as np
import torch
import torch.distributed as dist

# tests: issue the collectives even in a world of ONE rank, so that the RCCL code path executes on a
# single-GPU box (a world of one normally short-circuits every collective)
FORCE_COLLECTIVES = False
# bench.py: set to a list to have every wait for a collective bracketed by two HIP events on the compute stream,
# appended as (kind, event_before, event_after).  The first event completes when the kernels ahead of the wait have
# finished, the second when the collective has: the difference is the EXPOSED collective time (what the overlap
# did not hide).  Device tensors only; None = no events (the default: a record is a packet of its own in the queue).
WAIT_EVENTS = None


def _timed_wait(kind, works, on_cuda):
    ev = WAIT_EVENTS
    works = [w for w in works if w is not None]
    if ev is None or not on_cuda or not works:
        for w in works:
            w.wait()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for w in works:
        w.wait()
    e1.record()
    ev.append((kind, e0, e1))


def _collective(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


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
    if _collective(group):
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
    return m


class GradBucket:
    """All gradients of a module as ONE flat fp32 buffer (2.8 M elements for the mlp denoiser at
    D=512 ~ 11 MB; 6.4 M ~ 25 MB for ddpm.Backbone).

    The buffer is reduced in a few large chunks rather than per parameter -- xGMI links are
    point-to-point, so a ring all-reduce is per-link bound and few large messages beat many small
    ones -- and the chunks are cut along parameter boundaries at >= `chunk_bytes`, so that a chunk can
    leave as soon as the backward pass has produced its last gradient (`ready()`), overlapped with
    the GEMMs of the layers below.  Gradients are views into the buffer: no pack / unpack copies."""

    def __init__(self, params, chunk_bytes=4 << 20):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.offsets = {}
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self.offsets[id(p)] = (off, off + p.numel())
            off += p.numel()
        # chunks: runs of consecutive parameters, walked from the END of the parameter list (the order in
        # which backward produces gradients), closed once they hold chunk_bytes
        self.chunks, hi, acc = [], n, 0
        for p in reversed(self.params):
            acc += p.numel() * 4
            lo = self.offsets[id(p)][0]
            if acc >= chunk_bytes:
                self.chunks.append((lo, hi))
                hi, acc = lo, 0
        if hi > 0:
            self.chunks.append((0, hi))
        self._pending = {c: sum(1 for p in self.params if c[0] <= self.offsets[id(p)][0] < c[1]) for c in self.chunks}
        self._left, self._works, self._hooks, self._delivered = {}, [], [], set()

    def zero_(self):
        self.flat.zero_()

    def rebind(self):
        """autograd accumulates into an existing .grad in place, which keeps the views; if a
        gradient was replaced by a fresh tensor anyway, copy it back into the bucket."""
        for p in self.params:
            lo, hi = self.offsets[id(p)]
            view = self.flat[lo:hi].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view

    # ---- overlapped reduction -------------------------------------------------------------
    def arm(self, group=None):
        """Before backward: count down the parameters of every chunk as their gradients arrive
        (post-accumulate hooks) and launch the chunk's all-reduce when it is complete."""
        self._left = dict(self._pending)
        self._works = []
        self._delivered = set()
        self._group = group
        if not self._hooks:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def view_for(self, p):
        """The bucket view a backward kernel may write parameter p's gradient into directly (armed
        buckets only; hypdiff_amd.autograd.GRAD_SINK), or None."""
        if not self._left or id(p) not in self.offsets:
            return None
        if id(p) in self._delivered:          # before anything is overwritten
            raise RuntimeError("GradBucket: a second gradient for a parameter already delivered in this step (module "
                               "applied twice / shared weights): use DataParallelTrainer(..., overlap=False) for such a loss")
        lo, hi = self.offsets[id(p)]
        return self.flat[lo:hi].view_as(p)

    def wants_overlap(self):
        """True when an armed step has real all-reduces to start from the backward pass (more than one rank)."""
        return bool(self._left) and _collective(self._group)

    def delivered(self, p):
        """p's gradient has been WRITTEN (not accumulated) into view_for(p), enqueued on the current stream.  A
        parameter may be delivered once per armed step: a second delivery (the same module applied twice in one loss,
        shared weights) would overwrite the first contribution after its chunk may already be on the wire."""
        if id(p) in self._delivered:
            raise RuntimeError("GradBucket: a parameter's gradient was delivered twice in one step (module applied twice / "
                               "shared weights): run DataParallelTrainer(..., overlap=False) for such a loss")
        self._delivered.add(id(p))
        self._on_grad(p)

    def _on_grad(self, p):
        if not self._left or id(p) not in self.offsets:
            return
        lo = self.offsets[id(p)][0]
        for c in self.chunks:
            if c[0] <= lo < c[1]:
                self._left[c] -= 1
                if self._left[c] == 0:
                    self._launch(c)
                return

    def _launch(self, c):
        view = self.flat[c[0]:c[1]]
        # a gradient that autograd replaced instead of accumulating in place has to be copied in first
        for p in self.params:
            lo, hi = self.offsets[id(p)]
            if c[0] <= lo < c[1] and p.grad is not None and p.grad.data_ptr() != self.flat[lo:hi].data_ptr():
                self.flat[lo:hi].view_as(p).copy_(p.grad)
                p.grad = self.flat[lo:hi].view_as(p)
        if not _collective(self._group):      # one process: the bucket is only the sink of the backward kernels
            return
        view.div_(dist.get_world_size(self._group))
        self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self._group, async_op=True))

    def finish(self):
        """After backward: reduce whatever no hook has sent (parameters without a gradient this step)
        and wait for every chunk (the waits only order the current stream behind RCCL's)."""
        for c, left in list(self._left.items()):
            if left > 0:
                self._launch(c)
        self._left = {}
        _timed_wait("allreduce", self._works, self.flat.is_cuda)
        self._works = []

    def allreduce_mean_(self, group=None, async_op=False):
        """one blocking all-reduce of the whole bucket (the non-overlapped form)"""
        if not _collective(group):
            return None
        self.flat.div_(dist.get_world_size(group))
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class BucketAdam:
    """torch.optim.Adam with its defaults (no weight decay, no amsgrad: diff.py:107, diff_graphset.py:66) over a
    GradBucket of fp32 CUDA parameters, as ONE launch per step (hd_adam_step_f32) instead of torch's seven
    multi-tensor launches.  State = two flat arrays laid out like the bucket.  Every parameter of the bucket is updated
    on every step: one that received no gradient sees g = 0 (torch would skip it; the denoisers have no such
    parameter).  Results agree with torch's to rounding (same operation order, fused-multiply-add contraction aside)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        from . import _lib
        assert all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in bucket.params)
        self.bucket, self.lr, self.betas, self.eps, self.t = bucket, float(lr), tuple(betas), float(eps), 0
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        # {step count, -lr / (1 - b1^t), sqrt(1 - b2^t)} on the device, advanced by the launch itself: a captured HIP
        # graph of the step (DataParallelTrainer.capture) replays with the right bias corrections
        self.state = torch.zeros(3, dtype=torch.float64, device=bucket.flat.device)
        self._lib = _lib

    def step(self):
        from . import ops
        lib, b = self._lib, self.bucket
        self.t += 1
        ops._dev(b.flat)
        jobs = [lib.HdAdamJob(p.data_ptr(), b.offsets[id(p)][0], p.numel()) for p in b.params]
        for i in range(0, len(jobs), 128):
            part = jobs[i:i + 128]
            lib.call("hd_adam_step_f32", (lib.HdAdamJob * len(part))(*part), len(part), ops._p(b.flat), ops._p(self.exp_avg),
                     ops._p(self.exp_avg_sq), self.lr, self.betas[0], self.betas[1], self.eps, 1 if i == 0 else 0,
                     ops._p(self.state), ops._stream())
        # the kernel wrote the parameters behind autograd's back: bump their versions, which key the caches of
        # limb images and padded / transposed weight copies
        torch.autograd.graph.increment_version(b.params)


class DataParallelTrainer:
    """Data-parallel training step of a denoiser (diff.py:113-118 / diff_graphset.py:64-75 run
    loss -> backward -> Adam on one device; here every rank does that on its own shard of the
    graph set and the gradients are averaged in between).

    All gradients live in one flat fp32 bucket (the HIP backward kernels' outputs are accumulated
    into views of it).  With `overlap` (default) the bucket leaves in >= 4 MB chunks over RCCL while
    backward is still running -- 3 chunks for the mlp at D=512 (11 MB), 6 for ddpm.Backbone (25 MB);
    otherwise as ONE all-reduce after backward.  The optimizer is Adam on every rank (BucketAdam: one launch over the
    bucket; torch.optim.Adam for host tensors or when passed in) -- identical updates: weights never need a broadcast
    after the first."""

    def __init__(self, module, lr=1e-3, group=None, optimizer=None, broadcast=True, overlap=True):
        self.module, self.group, self.overlap = module, group, overlap
        self.bucket = GradBucket(module.parameters())
        if optimizer is not None:
            self.opt = optimizer
        elif self.bucket.params and all(p.is_cuda for p in self.bucket.params):
            self.opt = BucketAdam(self.bucket, lr=lr)
        else:                                     # host tensors (the gloo tests of the trainer's logic)
            self.opt = torch.optim.Adam(self.bucket.params, lr=lr)
        if broadcast and _collective(group):
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
        if self.overlap:
            from . import autograd as hag
            self.bucket.arm(self.group)
            hag.GRAD_SINK = self.bucket           # the stack's backward writes dW / db straight into the bucket
            try:
                loss.backward()
            finally:
                hag.GRAD_SINK = None
            self.bucket.finish()
        else:
            loss.backward()
            self.bucket.rebind()
            self.bucket.allreduce_mean_(self.group)
        self.opt.step()
        return loss.detach()

    def capture(self, loss_fn, warmup=3):
        """The whole step -- loss_fn, backward, optimizer -- as ONE HIP graph: returns replay() -> the loss tensor
        (a static tensor, overwritten by every replay).  For launch-bound steps (ddpm.Model at 200 rows is ~180
        launches of a few microseconds: 2.5 ms issued one by one, 1.3 ms of kernels).  Requirements: loss_fn reads
        the same device tensors on every call, draws its random numbers on the device (torch's CUDA generator is
        graph-safe; ddpm.Model does that while a capture is running), never reads anything back to the host; the
        optimizer is BucketAdam (its step count lives on the device).  With more than one rank (backend nccl = RCCL) the
        chunked gradient all-reduces are recorded into the graph as well -- they are stream operations -- so the graphed
        step survives data parallelism (every rank must capture and replay in lockstep).
        `warmup` eager steps run first on a side stream (allocations, caches).  The parameters' versions are bumped
        after every replay, as after an eager step."""
        coll = _collective(self.group)
        if coll and dist.get_backend(self.group) != "nccl":
            raise NotImplementedError("DataParallelTrainer.capture with more than one rank needs the nccl (RCCL) backend: "
                                      "its collectives are stream operations and can be recorded into the graph")
        if not isinstance(self.opt, BucketAdam):
            raise TypeError("DataParallelTrainer.capture needs the BucketAdam optimizer (device-side step count)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.step(loss_fn)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # with collectives in the step RCCL's watchdog thread polls events while the capture is open: thread-local
        # capture mode keeps its calls from invalidating the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if coll else "global"):
            loss = self.step(loss_fn)
        params = self.bucket.params

        def replay():
            graph.replay()
            self.opt.t += 1
            torch.autograd.graph.increment_version(params)
            return loss
        replay.graph = graph
        return replay


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
        # rows of the WHOLE graph above the SpMM's long-row threshold: the fused aggregate tail has two forms that
        # differ by rounding and picks by (rows, long rows); every shard runs the form the whole graph takes
        deg = np.bincount(indices[0].astype(np.int64), minlength=n) if indices.shape[1] else np.zeros(n, np.int64)
        self.n_long_whole = int((deg > 512).sum())
        self._build_halo(indices)

    HALO_MAX_FRACTION = 0.5

    def _build_halo(self, indices):
        """Halo exchange lists: which remote rows this rank's block references, per owner, and which
        of its own rows every peer references.  Every rank is constructed from the global COO, so both come from it
        without a request round; so does the decision -- halo exchange when EVERY rank references less than half of the
        rows it does not own (citation graphs), the equal-sized all-gather otherwise (Barabasi-Albert graphs reference
        ~80 %) -- which all ranks must take alike.  The local block is re-indexed for the halo form: local columns first
        (col - lo), then the referenced remote rows in (owner, id) order; the ORDER of a row's entries does not change,
        so its sum is formed exactly as on one GPU."""
        P, rpr, n = self.world, self.rows_per_rank, self.n
        self.halo_ids = [np.zeros(0, np.int64) for _ in range(P)]      # global ids this rank needs, per owner
        self.send_rows = [np.zeros(0, np.int64) for _ in range(P)]     # local rows of this rank that peer r needs
        self.halo_fraction, self.use_halo = 0.0, False
        if P == 1:
            return
        rows, cols = indices[0].astype(np.int64), indices[1].astype(np.int64)
        worst = 0.0
        for r in range(P):
            lo_r, hi_r = min(n, r * rpr), min(n, (r + 1) * rpr)
            a, b = np.searchsorted(rows, lo_r, side="left"), np.searchsorted(rows, hi_r, side="left")
            remote = np.unique(cols[a:b][(cols[a:b] < lo_r) | (cols[a:b] >= hi_r)])
            worst = max(worst, remote.size / max(1, n - (hi_r - lo_r)))
            if r == self.rank:
                owner = remote // rpr
                self.halo_ids = [remote[owner == q] for q in range(P)]
            else:
                mine = remote[(remote >= self.lo) & (remote < self.hi)]
                self.send_rows[r] = mine - self.lo
        self.halo_fraction = worst
        self.use_halo = worst < self.HALO_MAX_FRACTION
        if self.use_halo:
            total = sum(h.size for h in self.halo_ids)
            halo_all = np.concatenate(self.halo_ids) if total else np.zeros(0, np.int64)   # ascending: owners are ranges
            c = self.local_indices[1]
            local = (c >= self.lo) & (c < self.hi)
            newc = np.where(local, c - self.lo, rpr + np.searchsorted(halo_all, c))
            self.halo_indices = np.vstack((self.local_indices[0], newc)).astype(np.int64)
            self.halo_shape = (rpr, rpr + total)

    def _halo_exchange(self, x_local, group=None, async_op=False):
        """(rows_per_rank + halo, width) operand of the re-indexed block: this rank's rows, then the referenced rows of
        the peers, fetched by ONE all_to_all_single with the precomputed split sizes.  Returns (tensor, work)."""
        x_local = self._padded(x_local)
        w = x_local.shape[1]
        dev = x_local.device
        if not hasattr(self, "_send_idx") or self._send_idx.device != dev:
            cat = np.concatenate(self.send_rows) if sum(r.size for r in self.send_rows) else np.zeros(0, np.int64)
            self._send_idx = torch.from_numpy(cat).to(dev)
        send = x_local.index_select(0, self._send_idx)
        buf = torch.empty(self.halo_shape[1], w, dtype=x_local.dtype, device=dev)
        buf[: self.rows_per_rank].copy_(x_local)
        work = dist.all_to_all_single(buf[self.rows_per_rank:], send, [int(h.size) for h in self.halo_ids],
                                      [int(r.size) for r in self.send_rows], group=group, async_op=async_op)
        return buf, work

    def halo_bytes(self, width, itemsize=4):
        """bytes this rank receives per exchange of `width` columns (the all-gather receives (P - 1) blocks)"""
        return sum(int(h.size) for h in self.halo_ids) * width * itemsize

    def _padded(self, x_local):
        if x_local.shape[0] == self.rows_per_rank:
            return x_local
        pad = torch.zeros(self.rows_per_rank - x_local.shape[0], x_local.shape[1], dtype=x_local.dtype,
                          device=x_local.device)
        return torch.cat([x_local, pad], 0)

    def gather_rows(self, x_local, group=None):
        """(rows_per_rank, d) block of every rank -> (rows_per_rank * P, d), rank-major = global
        row order.  Direct all-gather: on the full xGMI mesh every rank writes its block to
        its 7 peers concurrently."""
        x_local = self._padded(x_local)
        if self.world == 1 and not FORCE_COLLECTIVES:
            return x_local
        out = torch.empty(self.rows_per_rank * self.world, x_local.shape[1], dtype=x_local.dtype,
                          device=x_local.device)
        dist.all_gather_into_tensor(out, x_local.contiguous(), group=group)
        return out

    def exchange_slabs(self, x_local, n_slabs, group=None):
        """gather_slabs in the halo form: per column slab an asynchronous all_to_all_single of the referenced rows
        only; the tensors are operands of the re-indexed block (shard_csr(..., halo=True))."""
        x_local = self._padded(x_local)
        d = x_local.shape[1]
        n_slabs = max(1, min(n_slabs, d // 4 if d >= 4 else 1))
        width = (-(-d // n_slabs) + 3) // 4 * 4
        out = []
        for lo in range(0, d, width):
            hi = min(d, lo + width)
            buf, work = self._halo_exchange(x_local[:, lo:hi], group, async_op=True)
            out.append((lo, hi, buf, work))
        return out

    def gather_slabs(self, x_local, n_slabs, group=None):
        """The same all-gather cut into column slabs, every slab an asynchronous collective of its own:
        returns [(col_lo, col_hi, gathered (rows_per_rank * P, width) tensor, work handle or None)].
        RCCL runs the slabs back to back on its own stream; the caller waits for slab k only (which
        orders the compute stream behind that one collective) and aggregates it while slabs k+1.. are
        still on the wire.  Slab widths are multiples of 4 columns (16-byte rows)."""
        x_local = self._padded(x_local)
        d = x_local.shape[1]
        n_slabs = max(1, min(n_slabs, d // 4 if d >= 4 else 1))
        width = -(-d // n_slabs)
        width = (width + 3) // 4 * 4
        out = []
        for lo in range(0, d, width):
            hi = min(d, lo + width)
            part = x_local[:, lo:hi].contiguous()
            if self.world == 1 and not FORCE_COLLECTIVES:
                out.append((lo, hi, part, None))
                continue
            full = torch.empty(self.rows_per_rank * self.world, hi - lo, dtype=x_local.dtype, device=x_local.device)
            work = dist.all_gather_into_tensor(full, part, group=group, async_op=True)
            out.append((lo, hi, full, work))
        return out

    def aggregate(self, xt_local, spmm, group=None):
        """HypAgg's support_t for the local rows: all-gather the tangent block, then
        spmm(local_adj, gathered).  `spmm(indices, values, shape, x)` is the device SpMM
        (ops.spmm on a CSR of the local block) -- injected so that the CPU tests can pass a
        host implementation.  Summation order per row is unchanged -> the support rows are bit-identical to 1 GPU."""
        if self.use_halo:
            buf, _ = self._halo_exchange(xt_local, group)
            return spmm(self.halo_indices, self.local_values, self.halo_shape, buf)[: self.hi - self.lo]
        full = self.gather_rows(xt_local, group)
        out = spmm(self.local_indices, self.local_values, self.local_shape, full)
        return out[: self.hi - self.lo]

    def aggregate_slabs(self, xt_local, spmm, n_slabs=4, group=None):
        """aggregate() with the all-gather pipelined against the SpMM in column slabs.  A feature
        column's sum does not involve the other columns, so the result is bit-identical."""
        n_loc = self.hi - self.lo
        outs = []
        if self.use_halo:
            for lo, hi, buf, work in self.exchange_slabs(xt_local, n_slabs, group):
                if work is not None:
                    work.wait()
                outs.append(spmm(self.halo_indices, self.local_values, self.halo_shape, buf)[:n_loc])
            return torch.cat(outs, 1)
        for lo, hi, full, work in self.gather_slabs(xt_local, n_slabs, group):
            if work is not None:
                work.wait()
            outs.append(spmm(self.local_indices, self.local_values, self.local_shape, full)[:n_loc])
        return torch.cat(outs, 1)


def sharded_hgcn_encode(encoder, x_local, sharded_adj, group=None, n_slabs=4):
    """HGCN.encode (models/encoders.py:115-119) for ONE graph whose rows are split over the ranks
    (config 5).  The input lift, HypLinear and HypAct are row-local; HypAgg needs the tangent rows
    of remote columns, so every layer is

        local rows:  lift / HypLinear tail -> tangent block (N/P, d)
        all ranks :  all-gather of the tangent blocks over RCCL / xGMI, in n_slabs column slabs
        local rows:  CSR SpMM over global columns on slab k while slab k+1 is on the wire, then
                     expmap0 / proj / HypAct on the assembled rows (hd_*_agg_tail)

    With one slab (or a world of one rank) the aggregate is the single fused kernel of the 1-GPU
    path, in the form (one or four rows per wavefront: ops.agg_form) that the WHOLE graph takes, so the
    result is bit-identical to encoder.encode on the whole graph.  `x_local` holds this rank's rows of
    the features (Hyperboloid: zero time column already prepended).  Per row and feature the summation
    order is that of the 1-GPU kernel.  The slab-pipelined path finishes the rows with hd_*_agg_tail,
    which evaluates the tail in the one-row form: bit-identical to the 1-GPU encode where that takes the
    one-row form too (fewer than 8192 rows, or a Hyperboloid graph with long rows), within a few ulp
    (2e-6 relative) of it otherwise.  Inference path (torch.no_grad())."""
    from .models.encoders import _lift
    n_loc = sharded_adj.hi - sharded_adj.lo
    assert x_local.shape[0] == n_loc, (x_local.shape, n_loc)
    halo = sharded_adj.use_halo
    csr = shard_csr(sharded_adj, x_local.device, halo=halo)
    h = _lift(encoder.manifold, x_local, encoder.curvatures[0])
    pipelined = n_slabs > 1 and (sharded_adj.world > 1 or FORCE_COLLECTIVES)
    for layer in encoder.layers:
        xt = layer.linear.forward(h, _tangent=True)
        if pipelined and xt.shape[1] >= 8 and xt.shape[1] % 4 == 0:
            slabs = sharded_adj.exchange_slabs(xt, n_slabs, group) if halo else sharded_adj.gather_slabs(xt, n_slabs, group)
            h = shard_aggregate_slabs(layer, csr, slabs, n_loc)
        elif halo:
            h = shard_aggregate(layer, csr, sharded_adj._halo_exchange(xt, group)[0], n_loc)
        else:
            h = shard_aggregate(layer, csr, sharded_adj.gather_rows(xt, group), n_loc)
    return h


def shard_csr(sharded_adj, device, halo=False):
    """device CSR of a rank's row block, built once: columns global, or (halo) re-indexed to [own rows | referenced
    remote rows] as RowShardedAdj._build_halo lays them out"""
    from . import ops
    attr = "_csr_halo" if halo else "_csr"
    if not hasattr(sharded_adj, attr):
        idx = sharded_adj.halo_indices if halo else sharded_adj.local_indices
        csr = ops.csr_from_coo(torch.from_numpy(idx).to(device),
                               torch.from_numpy(np.ascontiguousarray(sharded_adj.local_values)).to(device),
                               sharded_adj.halo_shape if halo else sharded_adj.local_shape)
        assert ops.CSR.LONG_ROW == 512
        csr.whole = (sharded_adj.n, sharded_adj.n_long_whole)
        setattr(sharded_adj, attr, csr)
    return getattr(sharded_adj, attr)


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


def shard_aggregate_slabs(layer, csr, slabs, n_loc):
    """shard_aggregate on column slabs as they arrive: plain SpMM of slab k into columns [lo, hi) of
    the support rows (hd_spmm_csr_strided), then the row tail (expmap0 -> proj [-> HypAct]) in one
    pass over the assembled support (hd_*_agg_tail) -- the same values as the fused aggregate."""
    from . import ops
    from .layers.hyp_layers import act_code
    code = act_code(layer.hyp_act.act)
    man = layer.agg.manifold
    d = slabs[-1][1]
    full0 = slabs[0][2]
    sup = torch.empty(csr.shape[0], d, dtype=full0.dtype, device=full0.device)
    whole = getattr(csr, "whole", None)
    thr = ops.agg_long_threshold(whole[0] if whole is not None else csr.shape[0])     # as the fused aggregate (ops.agg)
    for lo, hi, full, work in slabs:
        _timed_wait("allgather", [work], full.is_cuda)   # the compute stream waits for THIS slab's collective only
        ops.spmm_into(csr, full, sup, lo, long_thr=thr)
    out = ops.agg_tail(man.abi, sup, layer.agg.c, layer.hyp_act.c_out if code is not None else None, code)
    if code is None:
        out = layer.hyp_act.forward(out)
    return out[:n_loc]
