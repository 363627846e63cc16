#!/usr/bin/env python3
"""bench.py -- latent node-steps/sec (HGCN encode + DDPM step) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input:
    HGCN.encode (Hyperboloid, c=1, 2 layers, dim 512) over the block-diagonal batch of a
    SynEgo1000-shaped graph set (1197 graphs, sum N = 15 777 rows), followed by one
    direction-diffusion p_sample step (mlp denoiser + fused update, T = 1000 tables)
    over the (sum N, 512) latents.
value = rows processed by all ranks / wall time (inputs resident in HBM before timing).  The headline arrangement is
software-pipelined across batches: step i encodes graph-set batch i % 2 on a second HIP stream while p_sample advances the
latents of the OTHER batch on the main stream (the two phases of one batch are data-dependent in the reference pipeline,
diff.py:85-127; those of two batches are not); the streams are joined per step.  `--one-stream`: both phases on one
stream (what a caller without a pipeline gets; reported as the `one_stream` secondary otherwise).  No priming steps:
`warmup_effective` counts every untimed step issued before the K timed ones.

    python bench.py --gpus N --steps K --warmup W
N > 1: when WORLD_SIZE is not set this process is only a LAUNCHER -- it touches no GPU, starts N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), forwards rank 0's JSON line and exits
non-zero if any rank does.  Under torch.distributed.run (WORLD_SIZE set) it is a rank.  Each rank owns its own
graph-set shard of the same size (weak scaling, no data-path collective in the step).
`--dry --backend gloo`: the launcher / rendezvous / JSON merge without a GPU or a kernel (tests/test_bench_launcher.py).
The one-stream leg that runs ahead of a pipelined headline launches one 4 KB kernel per step on the side stream, so that its
hardware queue is live when the pipelined steps start (`config.side_queue_keepalive`; `--no-side-keepalive`).
More than one rank: a secondary leg that stops cannot cost the line (`--legs-timeout`).  HD_BENCH_SHARE_GPU=1 with
`--backend gloo`: every rank on cuda:0, collectives staged through the host -- a rehearsal of the N-rank path on a one-GPU
box, flagged in `config.rehearsal`, never a scaling figure.  HD_BENCH_VERBOSE=1: a progress line per leg and rank on stderr.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0
# dense MFMA peak per ALGORITHMIC fp32 flop of each GEMM kernel: the limb kernels issue six bf16
# MFMAs per fp32 product (csrc/gemm_limbs.hip, csrc/gemm_split.hip)
# MFMAs per fp32 product (csrc/gemm_limbs.hip, csrc/gemm_split.hip); the f16x2 kernel three fp16 MFMAs (same dense rate)
GEMM_PEAKS = {"mlp_chain": MFMA_BF16_PEAK_TFLOPS / 6, "gemm_limbs": MFMA_BF16_PEAK_TFLOPS / 6, "gemm_f32_bf16x3": MFMA_BF16_PEAK_TFLOPS / 6,
              "gemm_f16x2": MFMA_BF16_PEAK_TFLOPS / 3, "gemm_f32_mfma": MFMA_F32_PEAK_TFLOPS}
# kernel-name prefixes in the rocprofv3 PMC summary (the two limb formats are instantiations of one template)
GEMM_PMC_KEYS = {"mlp_chain": "mlp_chain", "gemm_limbs": "gemm_limbs", "gemm_f32_bf16x3": "gemm_f32_bf16x3",
                 "gemm_f16x2": "gemm_limbs<128, 8, true", "gemm_f32_mfma": "gemm_f32_mfma"}
PEAK_NOTES = {"mlp_chain": "dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 MFMAs per algorithmic fp32 product (the five denoiser "
                           "layers as ONE persistent launch, csrc/gemm_chain.hip)",
              "gemm_limbs": "dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 MFMAs per algorithmic fp32 product",
              "gemm_f32_bf16x3": "dense bf16 MFMA peak 2500 TFLOP/s / 6 bf16 MFMAs per algorithmic fp32 product",
              "gemm_f16x2": "dense fp16 MFMA peak 2500 TFLOP/s / 3 fp16 MFMAs per algorithmic fp32 product",
              "gemm_f32_mfma": "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"}
DTYPES = {"exact": "f32 (GEMMs on v_mfma_f32_32x32x2_f32: bit for bit an fmaf chain)",
          "bf16x3": "f32 (fp32-faithful: every GEMM operand element split exactly into 3 bf16 limbs, 6 exact limb "
                    "products per fp32 product accumulated in fp32, dropped terms < 2^-26 |a||w|; everything else "
                    "plain fp32 / fp64 as in the reference)",
          "f16x2": "f32 inputs/outputs, NARROWER-than-fp32 operands: inference GEMM operands (denoiser chain and "
                   "HypLinear) carried as 2 fp16 limbs of the power-of-two-scaled ROW (22 bits relative to the row "
                   "maximum), 3 exact limb products per fp32 product, fp32 accumulation"}


def gemm_summary(events):
    """Per GEMM kernel of the timed region: launches, HIP-event time, algorithmic flops / bytes."""
    per = {}
    for e in events:
        d = per.setdefault(e[4], dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
        d["launches"] += 1
        d["ms"] += e[0].elapsed_time(e[1])
        d["flops"] += e[2]
        d["bytes"] += e[3]
    return per


def timed_steps(w, T, steps, warmup, barrier, ops, dist, device, instrument=False):
    """W untimed + K timed steps between barriers.  instrument=False: nothing but the step's own
    launches goes to the stream (the headline).  instrument=True: every GEMM launch and the two
    phases of a step are bracketed by HIP events on the launch stream -- each record is a packet
    of its own in the queue and the 17 per step cost about 8 % of a 0.6 ms step, so the per-kernel
    durations of the roofline come from a second pass over the same K steps."""
    with torch.no_grad():
        for i in range(warmup):
            one_step(w, T - 1 - (i % T), None, i)
        barrier()
        ops.GEMM_EVENTS = [] if instrument else None
        phase = [] if instrument else None
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(w, T - 1 - (i % T), phase, i)
        barrier()
        elapsed = time.perf_counter() - t0
        events, ops.GEMM_EVENTS = ops.GEMM_EVENTS or [], None
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, events, phase or []


def progress(msg):
    """HD_BENCH_VERBOSE=1: one stderr line per leg and rank (where a multi-rank run is, should it stop)"""
    if os.environ.get("HD_BENCH_VERBOSE") == "1":
        sys.stderr.write("[bench rank %s %.1fs] %s\n" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, msg))
        sys.stderr.flush()


_T0 = time.perf_counter()


def stage_collectives_through_host(dist):
    """Rehearsal mode only (HD_BENCH_SHARE_GPU=1, backend gloo): every collective on device tensors becomes blocking copy
    to the host -> gloo on host tensors -> copy back.  gloo's own path for device tensors takes a fresh stream of torch's
    pool for every operation; with more HIP streams in use than hardware queues (GPU_MAX_HW_QUEUES = 4) its cross-stream
    waits never completed on the box (the second training step stopped; 16 queues: the row-sharded leg's eighth or so
    all-gather; tools/probes/rehearse_train.py).  RCCL runs every collective of a process group on ONE stream."""
    class Done:
        def wait(self):
            return True

    def host(t):
        return t.detach().cpu() if t.is_cuda else t

    raw = {n: getattr(dist, n) for n in ("all_reduce", "broadcast", "all_gather_into_tensor", "all_to_all_single", "all_gather")}

    def finish(pairs, async_op):
        for dev_t, host_t in pairs:
            if dev_t.is_cuda:
                dev_t.copy_(host_t)
        return Done() if async_op else None

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        h = host(t)
        raw["all_reduce"](h, op=op, group=group)
        return finish([(t, h)], async_op)

    def broadcast(t, src=0, group=None, async_op=False):
        h = host(t)
        raw["broadcast"](h, src=src, group=group)
        return finish([(t, h)], async_op)

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        ho = host(out)
        raw["all_gather_into_tensor"](ho, host(inp).contiguous(), group=group)
        return finish([(out, ho)], async_op)

    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        ho = host(out)
        raw["all_to_all_single"](ho, host(inp).contiguous(), output_split_sizes, input_split_sizes, group=group)
        return finish([(out, ho)], async_op)

    def all_gather(outs, inp, group=None, async_op=False):
        hs = [host(o) for o in outs]
        raw["all_gather"](hs, host(inp), group=group)
        return finish(list(zip(outs, hs)), async_op)

    dist.all_reduce, dist.broadcast, dist.all_gather_into_tensor = all_reduce, broadcast, all_gather_into_tensor
    dist.all_to_all_single, dist.all_gather = all_to_all_single, all_gather


def host_batch(seed, dim, feat=20):
    """One synthetic SynEgo1000-shaped graph-set batch, host side only (numpy): block-diagonal
    normalised COO, identity features with the Hyperboloid zero time column, labels, centres, latents."""
    from hypdiff_amd import workloads
    mats, sizes = workloads.ego_graph_set(seed=seed)
    idx, val, shape, feats = workloads.batch_graph_set(mats, sizes, feat)
    n = shape[0]
    x = np.concatenate([np.zeros((n, 1), np.float32), feats], axis=1)      # base_models.py:79-81
    rs = np.random.RandomState(seed)
    labels = rs.randint(0, 3, size=n).astype(np.int64)                     # host array, as the reference holds it
    u0 = (rs.standard_normal((3, dim)) * 0.05).astype(np.float32)
    img = rs.standard_normal((n, dim))                                     # fp64 state, as after step 1
    return dict(idx=idx, val=val, shape=shape, x=x, labels=labels, u0=u0, img=img, n=n, nnz=idx.shape[1], feat=feat + 1)


def build_workload(dim, T, seed, device):
    """Two batches of the same shape (seeds `seed`, `seed + 7919`): the pipelined headline encodes one while
    p_sample advances the latents of the other.  Both are resident in HBM before any timing starts."""
    from hypdiff_amd import ops
    from hypdiff_amd.models.direction_diffusion import diffusion
    from hypdiff_amd.models.encoders import HGCN
    hosts = [host_batch(seed, dim), host_batch(seed + 7919, dim)]
    feat = hosts[0]["feat"]
    torch.manual_seed(seed)
    args = types.SimpleNamespace(manifold="Hyperboloid", c=1.0, cuda=0, device=device, feat_dim=feat, dim=dim,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    enc = HGCN(torch.tensor([1.0], device=device), args).to(device).eval()
    diff = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T).eval()
    batches = []
    for hb in hosts:
        adj = torch.sparse_coo_tensor(torch.from_numpy(hb["idx"]).to(device), torch.from_numpy(hb["val"]).to(device),
                                      hb["shape"], check_invariants=False)
        x = torch.from_numpy(hb["x"]).to(device).contiguous()
        u0 = torch.from_numpy(hb["u0"]).to(device)
        img = torch.from_numpy(hb["img"]).to(device)
        with torch.no_grad():
            direction = diff.directions(u0)
            ops.csr_of(adj)
        labels_d = ops.labels_dev(hb["labels"], device)   # one upload, as p_sample_loop does for its T steps
        batches.append(dict(adj=adj, x=x, labels=labels_d, u0=u0, img=img, direction=direction))
    h0 = hosts[0]
    host = dict(idx=h0["idx"], val=h0["val"], shape=h0["shape"], x=torch.from_numpy(h0["x"]), labels=h0["labels"],
                u0=torch.from_numpy(h0["u0"]), img=torch.from_numpy(h0["img"]))
    w = dict(batches[0])
    w.update(enc=enc, diff=diff, batches=batches, n=h0["n"], nnz=h0["nnz"], nnz_other=hosts[1]["nnz"], host=host,
             feat=feat)
    return w


def one_step(w, t, ev=None, i=0):
    """One step: HGCN encode of one graph-set batch + one p_sample over (sum N, dim) latents.

    One stream (w["side_stream"] unset): both phases of batch 0 back to back.  Two streams (the
    headline): software pipeline ACROSS batches -- step i encodes batch i % 2 on the side stream while
    p_sample advances the latents of batch (i + 1) % 2 on the main stream; joined before the step returns."""
    side = w.get("side_stream")
    if side is None:
        idle = w.get("idle_side_stream")
        if idle is not None:
            # the side stream of the pipelined legs exists already: one 4 KB kernel per step keeps its hardware queue live
            # (a caller that pipelines has it live all along).  A queue that first becomes active when the pipelined steps
            # start costs them 5-10 % for their first ~12 steps (tools/probes/two_stream_ramp.py: 0.70-0.73 against 0.66 ms)
            with torch.cuda.stream(idle):
                w["tick"].add_(1)
        if ev is not None:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
        h = w["enc"].encode(w["x"], w["adj"])
        if ev is not None:
            e[1].record()
        pred, x0 = w["diff"].p_sample(w["img"], None, 1, w["direction"], w["labels"], w["u0"], t, None, True)
        if ev is not None:
            e[2].record()
            ev.append(e)
        return h, pred
    be, bd = w["batches"][i % 2], w["batches"][(i + 1) % 2]
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        h = w["enc"].encode(be["x"], be["adj"])
    pred, x0 = w["diff"].p_sample(bd["img"], None, 1, bd["direction"], bd["labels"], bd["u0"], t, None, True)
    main.wait_stream(side)
    # No h.record_stream(main): the block of h (allocated in the side stream's pool) is handed back when the caller drops
    # h, and whatever the side stream allocates next is enqueued behind the next step's side.wait_stream(main), i.e.
    # behind every use of h on the main stream that was issued before the drop -- the ordering record_stream would
    # enforce with events.  With record_stream the allocator may reuse a freed h only once an event on the main stream has
    # COMPLETED; the host runs tens of steps ahead of the GPU, so every step in flight took a fresh 32 MB block: 28
    # hipMalloc calls and 0.9 GB more reserved memory over the first 80 pipelined steps, which was the "settling" of the
    # two-stream arrangement (0.68 ms at step 10, 0.63 from step 50 on; tools/probes/two_stream_ramp.py).
    return h, pred


def cpu_baseline(w, dim, T):
    """The oracle (CPU restatement of the reference path) on the same workload, on the host cores
    (BASELINE.md section 3 / SURVEY.md 8d): the vectorised port (`kind: "port"`) and the loop-faithful
    form (`kind: "loop"`: tran_direction as the reference's per-row Python loop,
    models/direction_diffusion.py:158-161 -- the only per-row loop inside encode + p_sample), each with
    all host threads torch uses and with one thread; warm-ups + median of the timed steps,
    time.perf_counter.  Bounded: the all-thread legs are 3 warm-ups + 10 steps, the 1-thread legs 1 + 3."""
    from oracle import diffusion as odiff, layers as olayers, manifolds as om
    enc, diff, host = w["enc"], w["diff"], w["host"]
    Ws = [l.linear.weight.detach().cpu() for l in enc.layers]
    params = [(getattr(diff.mlp, f"fc{i}").weight.detach().cpu(), getattr(diff.mlp, f"fc{i}").bias.detach().cpu())
              for i in range(1, 6)]
    adj = (torch.from_numpy(host["idx"]), torch.from_numpy(host["val"]), host["shape"])
    tables = odiff.schedule(T)
    signs = odiff.sign_table(host["u0"])
    direction = list(om.pb_logmap0(host["u0"], 1.0).unbind(0))
    all_threads = torch.get_num_threads()

    def step(t, loop):
        olayers.hgcn_encode("Hyperboloid", host["x"], adj, Ws, [1.0, 1.0, 1.0], prepend_time=False)
        odiff.p_sample(host["img"], t, signs, host["labels"], tables, params, loop_direction=direction if loop else None)

    def leg(kind, threads, warm, reps):
        torch.set_num_threads(threads)
        try:
            with torch.no_grad():
                for i in range(warm):
                    step(T - 1 - i, kind == "loop")
                ts = []
                for i in range(reps):
                    t0 = time.perf_counter()
                    step(T - 1 - warm - i, kind == "loop")
                    ts.append(time.perf_counter() - t0)
        finally:
            torch.set_num_threads(all_threads)
        med = sorted(ts)[len(ts) // 2]
        return dict(value=w["n"] / med, unit="node-steps/s", cores=threads, kind=kind, ms_per_step=med * 1e3,
                    sample=f"{warm} warm-up + median of {reps} full steps of the same workload ({w['n']} rows), "
                           f"{'loop-faithful (per-row tran_direction)' if kind == 'loop' else 'vectorised'} torch-CPU "
                           f"oracle, {threads} thread{'s' if threads > 1 else ''}")

    main = leg("port", all_threads, 3, 10)
    main["others"] = [leg("loop", all_threads, 3, 10), leg("port", 1, 1, 3), leg("loop", 1, 1, 3)]
    return main


def train_leg(w, dim, device, steps, warmup, barrier, dist):
    """Secondary figure (outside the K timed headline steps): data-parallel TRAINING of the
    direction-diffusion denoiser on the same rows -- diffusion.forward (logmap at the cluster
    centres, standardisation, q_sample, mlp denoiser, MSE) -> backward (HIP kernels) -> one
    flat-bucket all-reduce over RCCL -> Adam; every rank trains on its own graph-set shard."""
    from hypdiff_amd.dist import DataParallelTrainer
    diff = w["diff"].train()
    rs = np.random.RandomState(7)
    emb = rs.standard_normal((w["n"], dim)).astype(np.float32)
    emb *= 0.5 / np.linalg.norm(emb, axis=1, keepdims=True)
    emb = torch.from_numpy(emb).to(device)
    if dist is not None:
        diff.sync_moments = True
    tr = DataParallelTrainer(diff.mlp, lr=1e-3)
    loss_fn = lambda: diff(emb, None, w["host"]["labels"], w["u0"], restrict=True, graphset=True)
    from hypdiff_amd import dist as hdist
    for _ in range(warmup):
        tr.step(loss_fn)
    barrier()
    hdist.WAIT_EVENTS = None if os.environ.get("HD_BENCH_NO_WAIT_EVENTS") == "1" else []
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.step(loss_fn)
        progress("training step %d issued" % i)
    barrier()
    dt = time.perf_counter() - t0
    waits, hdist.WAIT_EVENTS = hdist.WAIT_EVENTS or [], None
    exposed_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in waits) / steps
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    world = dist.get_world_size() if dist is not None else 1
    diff.eval()
    return {"metric": "denoiser training node-steps/s (forward + backward + gradient all-reduce + Adam)",
            "value": w["n"] * steps * world / dt, "unit": "node-steps/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "grad_bucket_bytes": int(tr.bucket.flat.numel() * 4), "final_loss": float(loss),
            "allreduce_bytes_per_step_per_gpu": (int(tr.bucket.flat.numel() * 4) + 24) if world > 1 else 0,
            "allreduce_chunks": len(tr.bucket.chunks),
            "exposed_collective_ms_per_step": exposed_ms,
            "exposed_note": "HIP events on the compute stream around the wait for the gradient chunks (rank 0): time the "
                            "all-reduce was not hidden behind the backward GEMMs; 0 in a world of one",
            "parallelism": "data-parallel x%d, gradient bucket all-reduced in %d chunks started from the backward pass"
                           % (world, len(tr.bucket.chunks))}


def sharded_leg(device, rank, world, steps, warmup, barrier, dist):
    """Secondary figure: config 5's path -- ONE power-law graph (Barabasi-Albert-like, m = 5, natural hubs
    of ~10^4 neighbours) whose rows are split over the ranks, 125 000 rows per rank (weak scaling: 10^6
    rows at 8 GPUs), features N(0, 0.1^2) of width 256, PoincareBall c = 0.6, HGCN 256 -> 256 -> 256.
    Per layer the tangent block of every rank is all-gathered over RCCL in 4 column slabs, pipelined
    against the local CSR SpMM (hypdiff_amd/dist.py).  value = rows of all ranks / wall time."""
    from hypdiff_amd import dist as hdist, workloads
    from hypdiff_amd.models.encoders import HGCN
    rows_per_rank, d = 125_000, 256
    n = rows_per_rank * world
    adj = workloads.power_law_graph(n, 5, seed=1432)
    big = workloads.normalize_adj(adj).tocoo()
    idx = np.vstack((big.row, big.col)).astype(np.int64)
    val = big.data.astype(np.float32)
    sh = hdist.RowShardedAdj(idx, val, (n, n), rank, world)
    rs = np.random.RandomState(100 + rank)
    x = torch.from_numpy((rs.standard_normal((sh.hi - sh.lo, d)) * 0.1).astype(np.float32)).to(device)
    args = types.SimpleNamespace(manifold="PoincareBall", c=0.6, cuda=0, device=device, feat_dim=d, dim=d,
                                 num_layers=2, act="relu", dropout=0.0, bias=0, use_att=0, local_agg=0)
    torch.manual_seed(1432)
    enc = HGCN(torch.tensor([0.6], device=device), args).to(device).eval()
    csr = hdist.shard_csr(sh, device, halo=sh.use_halo)
    with torch.no_grad():
        for _ in range(warmup):
            hdist.sharded_hgcn_encode(enc, x, sh)
        barrier()
        hdist.WAIT_EVENTS = []
        t0 = time.perf_counter()
        for _ in range(steps):
            h = hdist.sharded_hgcn_encode(enc, x, sh)
        barrier()
        dt = time.perf_counter() - t0
        waits, hdist.WAIT_EVENTS = hdist.WAIT_EVENTS, None
        exposed_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in waits) / steps
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    nnz_loc = int(sh.local_values.shape[0])
    return {"metric": "row-sharded HGCN encode nodes/s (config 5 shape)", "value": n * steps / dt, "unit": "nodes/s",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "rows_total": n, "rows_per_gpu": rows_per_rank, "dim": d,
            "nnz_per_gpu": nnz_loc, "hub_rows_per_gpu": int(csr.n_long), "finite": bool(torch.isfinite(h).all()),
            "exchange": "halo (all_to_all_single of the referenced remote rows)" if sh.use_halo else
                        "all-gather" + (" (every rank would need >= 50 % of the remote rows)" if world > 1 else ""),
            "remote_rows_referenced_max_fraction": sh.halo_fraction,
            "allgather_bytes_per_layer_per_gpu": rows_per_rank * d * 4 * (world - 1),
            "received_bytes_per_layer_per_gpu": sh.halo_bytes(d) if sh.use_halo else rows_per_rank * d * 4 * (world - 1),
            "exposed_collective_ms_per_step": exposed_ms,
            "exposed_note": "HIP events on the compute stream around each slab's wait (rank 0, summed over 2 layers x 4 "
                            "slabs): all-gather time not hidden behind the SpMM of the previous slab; 0 in a world of one",
            "parallelism": "rows sharded x%d, tangent all-gather per layer in 4 column slabs overlapped with the SpMM" % world}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, dry):
    """`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process stays a LAUNCHER.  It makes no GPU
    call (torch.cuda.device_count() does not initialise the device on this image; nothing of hypdiff_amd is imported),
    starts N fresh rank processes of this same script -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, exactly what torch.distributed.run would set -- and waits.  Rank 0's stdout is
    forwarded (its last line is the JSON line); the other ranks' stdout goes to stderr.  Exit status: the first
    non-zero status of any rank (the remaining ranks are then terminated by PID), else 0."""
    if not dry and os.environ.get("HD_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible\n")
            return 2
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HD_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else sys.stderr))
    rc = 0
    live = set(range(n))
    while live:
        time.sleep(0.1)
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                sys.stderr.write(f"bench.py: rank {r} exited with status {code}; stopping the other ranks\n")
                for q in live:
                    procs[q].terminate()          # exact PIDs of the children started above
    out0.seek(0)
    for line in out0.read().splitlines():         # the JSON line to stdout; anything else rank 0 printed (gloo's connection
        dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr      # notice goes to stdout) to stderr
        dst.write(line + "\n")
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


def start_legs_watchdog(out, rank, limit_s, first_leg):
    """A secondary leg that stops with more than one rank (a rank that left the leg early on an error, a collective that
    never completes) must not cost the headline: when the returned event has not been set `limit_s` seconds from now,
    rank 0 prints the line as it stands -- the unfinished leg (current_leg[0]) as an error entry -- and every rank ends
    its process (each rank runs its own watchdog).  limit_s <= 0: no watchdog.  Returns (done_event, current_leg)."""
    done, current_leg = threading.Event(), [first_leg]

    def watchdog():
        if done.wait(limit_s):
            return
        if rank == 0:
            out[current_leg[0]] = {"error": "leg not finished %d s after the headline steps (--legs-timeout): abandoned; the "
                                            "headline figures of this line were measured before it and are unaffected" % limit_s}
            out.setdefault("cpu_baseline", None)
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()
        sys.stderr.write("bench.py: rank %d: secondary leg '%s' exceeded --legs-timeout; exiting\n" % (rank, current_leg[0]))
        sys.stderr.flush()
        os._exit(0)

    if limit_s > 0:
        threading.Thread(target=watchdog, daemon=True).start()
    return done, current_leg


def dry_run(a, rank, world):
    """`--dry`: everything around the kernels -- rank environment, process group (`--backend`), host-side workload
    build of this rank's shard, barriers, max-over-ranks timing, per-rank gather, the JSON line -- without touching a
    GPU.  The "step" is a host checksum of the shard's adjacency, so value is NOT a performance number."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    hb = host_batch(1432 + rank, min(a.dim, 8))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    chk = 0
    for _ in range(a.steps):
        chk = int(hb["idx"].sum() % 1000003)
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    rows = torch.tensor([hb["n"]], dtype=torch.int64)
    per_rank = [rows.clone() for _ in range(world)]
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_gather(per_rank, rows)
    rows_per_gpu = [int(r.item()) for r in per_rank]
    stall = os.environ.get("HD_BENCH_DRY_STALL_RANK")      # test hook: that rank never joins the "leg" below
    out = None
    if rank == 0:
        out = ({"metric": "latent node-steps/sec (HGCN encode + DDPM step)",
                          "value": None, "unit": "node-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": float(el.item()) / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "synthetic", "dry": True,
                          "config": {"workload": "DRY RUN: launcher / rendezvous / JSON merge only, no kernel ran; "
                                                 "value is not a performance number",
                                     "rows_per_gpu": rows_per_gpu, "backend": a.backend, "checksum": chk,
                                     "launched_by": "bench.py launcher" if os.environ.get("HD_BENCH_LAUNCHED") else
                                                    ("external launcher" if world > 1 else "single process")}})
    if stall is not None and world > 1:
        done, _ = start_legs_watchdog(out, rank, a.legs_timeout, "dry_stall")
        if rank == int(stall):
            time.sleep(3600)
        dist.barrier()                                     # never completes: the watchdog ends every rank
        done.set()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-stream", action="store_true",
                    help="headline with both phases of a step on ONE HIP stream (default: software pipeline across two "
                         "batches on two streams, joined per step)")
    ap.add_argument("--no-side-keepalive", action="store_true",
                    help="the one-stream leg ahead of a pipelined headline leaves the side stream untouched (default: one "
                         "4 KB kernel per step on it, so that its hardware queue is live when the pipelined steps start)")
    ap.add_argument("--no-overlap-leg", action="store_true",
                    help="skip the secondary leg that repeats the steps with the other stream arrangement")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the comparison legs on the other fp32 GEMM kernels")
    ap.add_argument("--train-steps", type=int, default=20, help="steps of the secondary training leg (0 = skip)")
    ap.add_argument("--sharded-steps", type=int, default=5, help="steps of the secondary row-sharded leg (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="headline steps only (profiling runs)")
    ap.add_argument("--legs-timeout", type=int, default=420,
                    help="more than one rank: seconds the secondary legs may take in all before the line is printed without "
                         "the unfinished one (0 = no limit)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo with --dry)")
    ap.add_argument("--dry", action="store_true", help="no GPU, no kernels: launcher / rendezvous / JSON plumbing only")
    a = ap.parse_args()
    if a.no_secondary:
        a.no_overlap_leg = a.no_exact_leg = True
        a.train_steps = a.sharded_steps = 0

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:], a.dry))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.dry:
        sys.exit(dry_run(a, rank, world))
    if a.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: running the {world} rank(s) of the launcher\n")
    dist = None
    # HD_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a box with ONE GPU -- every rank on cuda:0, backend gloo (RCCL
    # refuses two ranks on one device); the line says so (config.rehearsal) and its value is not a scaling figure
    share = os.environ.get("HD_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share or world == 1 else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group(a.backend, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend)
            if share:
                stage_collectives_through_host(dist)
    device = f"cuda:{dev_index}"
    if os.environ.get("HD_BENCH_VERBOSE") == "1":      # a run that stops: every thread's Python stack after HD_BENCH_DUMP_S seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("HD_BENCH_DUMP_S", "600")), exit=False)
    progress("process group up (world %d, backend %s), device %s" % (world, a.backend if world > 1 else "-", device))

    from hypdiff_amd import ops
    T = a.timesteps
    w = build_workload(a.dim, T, 1432 + rank, device)
    progress("workload resident")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Headline: software pipeline across two batches -- the encode of batch i % 2 on a second HIP stream while p_sample
    # advances the other batch's latents (the HBM-bound encode kernels run in the partly empty tile rounds of the
    # denoiser GEMMs); joined per step.  The per-kernel pass below stays on one stream: a kernel's duration there is
    # its own, not that of two kernels sharing the chip.  No priming steps: what runs before the K timed steps is the
    # secondary leg in the other arrangement (a leg of its own, reported) and the W warm-up steps; `warmup_effective`
    # counts both.
    side = None if a.one_stream else torch.cuda.Stream()
    rows_total = w["n"] * a.steps * world
    other = None
    untimed_before = 0
    if not a.no_overlap_leg:
        # secondary, run FIRST: the same K steps in the other stream arrangement.  It comes before the headline on
        # purpose and the line says so (config.order, warmup_effective): a run of 25 steps is 18 ms, less than the chip
        # takes to leave its idle clocks, and whichever leg came first measured 8-10 % slow in short runs.
        if side is None:
            w["side_stream"] = torch.cuda.Stream()
        elif not a.no_side_keepalive:
            w["idle_side_stream"], w["tick"] = side, torch.zeros(1024, device=device)
        try:
            k3 = max(a.steps, 100)                     # its own step count: at least 100, i.e. >= 70 ms of load
            w3 = min(a.warmup, 3)
            el3, _, _ = timed_steps(w, T, k3, w3, barrier, ops, dist, device)
            untimed_before += k3 + w3
            other = ("two_streams" if side is None else "one_stream",
                     {"value": w["n"] * k3 * world / el3, "unit": "node-steps/s", "steps": k3, "ms_per_step": el3 / k3 * 1e3,
                      "note": ("same steps as a software pipeline across two batches on two HIP streams, joined per step"
                               if side is None else "both phases of ONE batch on one HIP stream: what a caller without a "
                               "pipeline gets") + "; run before the headline steps (first on the idle chip)"})
        finally:
            w.pop("side_stream", None)
            w.pop("idle_side_stream", None)
    progress("secondary stream-arrangement leg done")
    if side is not None:
        w["side_stream"] = side
    try:
        elapsed, _, _ = timed_steps(w, T, a.steps, a.warmup, barrier, ops, dist, device)
    finally:
        w.pop("side_stream", None)
    untimed_before += a.warmup
    progress("headline steps done")
    el_inst, events, phase = timed_steps(w, T, a.steps, min(a.warmup, 3), barrier, ops, dist, device, instrument=True)
    rows_per_gpu = [w["n"]]
    if dist is not None:
        mine = torch.tensor([w["n"]], dtype=torch.int64, device=device)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        rows_per_gpu = [int(r.item()) for r in per_rank]
        rows_total = sum(rows_per_gpu) * a.steps

    # dominant kernel: the fp32 GEMM of the denoiser / HypLinear (live HIP-event timing per launch)
    per = gemm_summary(events)
    dom = max(per, key=lambda k: per[k]["ms"])
    d = per[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
    peak = GEMM_PEAKS[dom]
    # HBM bytes per launch from the PMC passes of the same command (profiles/, collected with
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE; cannot be sampled from inside the process)
    traffic, traffic_src = None, None
    tp = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_bench_hbm_traffic.json") for r in (3, 2))
               if os.path.exists(q)), None)
    if tp is not None:
        try:
            ks = json.load(open(tp))["kernels"]
            hit = [v for k, v in ks.items() if k.startswith(GEMM_PMC_KEYS[dom])]
            if hit:
                traffic = int(sum(v["hbm_bytes_corrected"] * v["launches"] for v in hit) / sum(v["launches"] for v in hit))
                traffic_src = "profiles/%s (rocprofv3 PMC, per-launch mean)" % os.path.basename(tp)
        except Exception:
            traffic = None
    gemm_ms = sum(v["ms"] for v in per.values())
    out = {
        "metric": "latent node-steps/sec (HGCN encode + DDPM step)",
        "value": rows_total / elapsed, "unit": "node-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "warmup_effective": untimed_before,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPES[ops.GEMM_MODE],
        "data": "synthetic",
        "config": {"workload": "configs[1]: SynEgo1000-shaped graph set (1197 ego graphs, sum N=%d, nnz=%d), "
                               "Hyperboloid c=1 HGCN encode (21->%d->%d) + direction-diffusion p_sample "
                               "(mlp denoiser, T=%d tables)" % (w["n"], w["nnz"], a.dim, a.dim, T),
                   "rows_per_gpu": rows_per_gpu, "dim": a.dim, "timesteps": T,
                   "parallelism": "graph-set shards x%d" % world, "gemm_mode": ops.GEMM_MODE,
                   "launched_by": "bench.py launcher" if os.environ.get("HD_BENCH_LAUNCHED") else
                                  ("external launcher" if world > 1 else "single process"),
                   "streams": "one HIP stream (both phases of one batch back to back)" if side is None else
                              "two HIP streams, software-pipelined across batches: step i encodes batch i % 2 while "
                              "p_sample advances the latents of batch (i + 1) % 2; joined per step",
                   "priming_steps": 0,
                   **({"rehearsal": "HD_BENCH_SHARE_GPU=1: all %d ranks share cuda:0 over backend %s; exercises the N-rank "
                                    "code path only, NOT a scaling measurement" % (world, a.backend)} if share and world > 1 else {}),
                   "side_queue_keepalive": bool(other and side is not None and not a.no_side_keepalive),
                   "order": ("secondary leg in the other stream arrangement (%d untimed-for-the-headline steps%s), then "
                             % (untimed_before - a.warmup,
                                "; each of them also launches one 4 KB kernel on the side stream of the pipelined legs, so that "
                                "its hardware queue is live when they start" if side is not None and not a.no_side_keepalive
                                else "") if other else "") +
                            "W warm-up + K timed headline steps, then the instrumented per-kernel pass, then the other legs; "
                            "warmup_effective = every step issued before the K timed ones"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "traffic_unit": "HBM bytes per launch (mean)", "traffic_source": traffic_src,
                     "peak_note": PEAK_NOTES[dom],
                     "achieved_over_f32_mfma_peak": achieved / MFMA_F32_PEAK_TFLOPS,
                     "timing": "HIP events on the launch stream around every GEMM launch of a second, instrumented "
                               "pass over the same K steps, both phases on ONE stream so that a launch's duration is "
                               "the kernel's own (%.4f ms per step with the 17 event records per step in the queue; "
                               "the headline K steps carry none)" % (el_inst / a.steps * 1e3),
                     "algorithmic_bytes_per_launch": d["bytes"] / max(1, d["launches"]), "launches": d["launches"],
                     "kernel_ms_per_step": d["ms"] / a.steps, "all_gemm_ms_per_step": gemm_ms / a.steps,
                     "other_gemm_kernels": {k: {"launches": v["launches"], "TFLOP/s": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                                                "ms_per_step": v["ms"] / a.steps}
                                            for k, v in per.items() if k != dom},
                     "memory_bound_kernels": "profiles/r02_kernel_roofline.jsonl (tools/bench_kernels.py)"},
    }
    enc_ms = sum(e[0].elapsed_time(e[1]) for e in phase) / max(1, len(phase))
    dif_ms = sum(e[1].elapsed_time(e[2]) for e in phase) / max(1, len(phase))
    out["breakdown"] = {"encode_ms": enc_ms, "encode_nodes_per_s_per_gpu": w["n"] / (enc_ms * 1e-3) if enc_ms else None,
                        "p_sample_ms": dif_ms,
                        "ddpm_node_steps_per_s_per_gpu": w["n"] / (dif_ms * 1e-3) if dif_ms else None,
                        "note": "HIP-event time of the two phases of one step on rank 0, instrumented pass (SURVEY.md 8d)"}
    # From here on only secondary legs run; with more than one rank a watchdog keeps a leg that stops from costing the headline
    legs_done, current_leg = start_legs_watchdog(out, rank, a.legs_timeout if world > 1 else 0, "other_gemm_modes")
    if not a.no_exact_leg:
        # the same K steps under the other fp32 GEMM kernels, for comparison
        mode = ops.GEMM_MODE
        out["other_gemm_modes"] = {}
        try:
            for m2 in ("bf16x3", "exact", "f16x2"):
                if m2 == mode:
                    continue
                ops.GEMM_MODE = m2
                if side is not None:
                    w["side_stream"] = side              # the same stream arrangement as the headline
                try:
                    el2, _, _ = timed_steps(w, T, a.steps, min(a.warmup, 3), barrier, ops, dist, device)
                finally:
                    w.pop("side_stream", None)
                _, ev2, _ = timed_steps(w, T, a.steps, 1, barrier, ops, dist, device, instrument=True)
                per2 = gemm_summary(ev2)
                k2 = max(per2, key=lambda k: per2[k]["ms"])
                ach2 = per2[k2]["flops"] / (per2[k2]["ms"] * 1e-3) / 1e12
                out["other_gemm_modes"][m2] = {
                    "value": rows_total / el2, "unit": "node-steps/s", "ms_per_step": el2 / a.steps * 1e3,
                    "roofline": {"bound": "mfma", "kernel": k2, "achieved": ach2, "peak": GEMM_PEAKS[k2],
                                 "unit": "TFLOP/s", "frac": ach2 / GEMM_PEAKS[k2]},
                    "dtype": DTYPES[m2], "note": "same workload and steps with HD_GEMM_MODE=" + m2}
        finally:
            ops.GEMM_MODE = mode
    if other is not None:
        out[other[0]] = other[1]
    if a.train_steps > 0:
        try:
            progress("instrumented pass and comparison legs done; training leg")
            current_leg[0] = "train"
            out["train"] = train_leg(w, a.dim, device, a.train_steps, 3, barrier, dist)
        except Exception as e:                      # never lose the headline line to the secondary leg
            out["train"] = {"error": repr(e)}
    if a.sharded_steps > 0:
        try:
            progress("row-sharded leg")
            current_leg[0] = "sharded"
            out["sharded"] = sharded_leg(device, rank, world, a.sharded_steps, 2, barrier, dist)
        except Exception as e:
            out["sharded"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        progress("CPU baseline")
        out["cpu_baseline"] = cpu_baseline(w, a.dim, T)
    elif rank == 0:
        out["cpu_baseline"] = None
    legs_done.set()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
