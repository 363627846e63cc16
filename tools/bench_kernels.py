#!/usr/bin/env python3
"""Per-kernel HBM roofline of the memory-bound manifold / SpMM / diffusion kernels.

Sizes follow config 5 (N = 10^6 rows, dim 256, PoincareBall, power-law graph m=5) so every
operand (1 GB) is far beyond the 256 MiB Infinity Cache, plus the config-2 shape
(15 777 x 512).  Prints one JSON line per kernel: algorithmic bytes
/ HIP-event time, against the 8 TB/s HBM3E peak.
"""
import argparse
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import manifolds as hm, ops, workloads  # noqa: E402
from hypdiff_amd.models.direction_diffusion import diffusion  # noqa: E402

PEAK = 8000.0


power_law_graph = workloads.power_law_graph


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, shape, nbytes, sec, extra=None):
    gbs = nbytes / sec / 1e9
    row = {"kernel": name, "shape": shape, "us": round(sec * 1e6, 1), "alg_bytes": int(nbytes),
           "GB/s": round(gbs, 1), "frac_hbm_peak": round(gbs / PEAK, 3)}
    if extra:
        row.update(extra)
    print(json.dumps(row), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    pb, hb = hm.PoincareBall(), hm.Hyperboloid()
    c = 0.6
    with torch.no_grad():
        for (n, d) in ((a.rows, a.dim), (15777, 512), (a.rows, 64)):
            x = torch.randn(n, d, device=dev) * 0.1
            p = pb.proj(pb.expmap0(x, c), c)
            q = pb.proj(pb.expmap0(torch.randn(n, d, device=dev) * 0.1, c), c)
            ph = hb.expmap0(hb.proj_tan0(x, 1.0), 1.0)
            B = n * d * 4
            sh = [n, d]
            report("pb_expmap0_proj", sh, 2 * B, timeit(lambda: ops.map1("hd_pb_expmap0_proj", x, c), a.iters))
            report("pb_logmap0", sh, 2 * B, timeit(lambda: pb.logmap0(p, c), a.iters))
            report("pb_proj", sh, 2 * B, timeit(lambda: pb.proj(p, c), a.iters))
            report("pb_hyp_act", sh, 2 * B, timeit(lambda: ops.hyp_act("pb", p, c, c, 1), a.iters))
            report("pb_mobius_add", sh, 3 * B, timeit(lambda: pb.mobius_add(p, q, c), a.iters))
            report("pb_sqdist", sh, 2 * B + 4 * n, timeit(lambda: pb.sqdist(p, q, c), a.iters))
            report("hb_expmap0_proj", sh, 2 * B, timeit(lambda: ops.map1("hd_hb_expmap0_proj", x, 1.0), a.iters))
            report("hb_logmap0", sh, 2 * B, timeit(lambda: hb.logmap0(ph, 1.0), a.iters))
            report("hb_hyp_act", sh, 2 * B, timeit(lambda: ops.hyp_act("hb", ph, 1.0, 1.0, 1), a.iters))
            # diffusion elementwise kernels
            import types
            dm = diffusion(types.SimpleNamespace(r=2.0, t=1.0), d, num_timesteps=1000)
            u0 = torch.randn(3, d, device=dev) * 0.05
            direction = dm.directions(u0)
            labels = np.random.RandomState(0).randint(0, 3, size=n)
            t = torch.randint(0, 1000, (n,), device=dev)
            x64, nz = x.double(), torch.randn(n, d, device=dev, dtype=torch.float64)
            report("q_sample(f64,f64,restrict)", sh, n * d * (8 + 8 + 8 + 8) + 12 * n,
                   timeit(lambda: dm.q_sample(x64, t, direction, nz, labels, u0, restrict=True), a.iters))
            x0 = torch.randn(n, d, device=dev)
            lab = ops.labels_dev(labels, dev)
            signs = dm._signs(direction)
            out = torch.empty(n, d, dtype=torch.float64, device=dev)
            H = dm._host

            def upd():
                ops._lib.call("hd_p_sample_update", ops._p(x64), 1, ops._p(x0), ops._p(lab), ops._p(signs), 3,
                              float(H["sqrt_recip_alphas_cumprod"][500]), float(H["sqrt_recipm1_alphas_cumprod"][500]),
                              float(H["sqrt_alphas_cumprod"][499]), float(H["minus_alphas_cumprod"][499]), 500, 499,
                              ops._p(out), n, d, 1.0, 500.0, 1.0, ops._stream())
            report("p_sample_update(f64 x_t)", sh, n * d * (8 + 4 + 4 + 8) + 4 * n, timeit(upd, a.iters))
            report("moments+standardize(f32)", sh, 3 * B, timeit(lambda: dm.normalization(x), a.iters))
            # the 64-lane-per-row producers / tails below are what the product path runs from 128 (producers) /
            # 256 (tail) columns on; narrower rows take the generic row kernels measured above (ops.limb_rows_fmt,
            # manifolds/hyperboloid.py), so they are not timed outside their range
            if 128 <= d <= 1023:
                imgb = ops.limb_empty(n, d + 1, dev, "bf16x3")
                nsl = (d + 1 + 31) // 32 * 32

                def mpackb():
                    ops._lib.call("hd_mlp_pack_input_bf16x3", ops._p(x64), 1, None, 2, 37.0, ops._p(imgb.buf), n, d, ops._stream())
                report("mlp_pack_input_bf16x3(f64 x)", sh, n * (8 * d + 6 * nsl), timeit(mpackb, a.iters))
                dsl = (d + 31) // 32 * 32
                report("row_pack_bf16x3(hb logmap0)", sh, n * (4 * d + 6 * dsl), timeit(lambda: ops.row_pack(ph, 1, 1.0, "bf16x3"), a.iters))
                del imgb
            if 128 <= d <= 1535:
                img = ops.limb_empty(n, d + 1, dev, "f16x2", zero_amax=False)
                nsl = (d + 1 + 31) // 32 * 32

                def mpack():
                    ops._lib.call("hd_mlp_pack_input_f16x2", ops._p(x64), 1, None, 2, 37.0, ops._p(img.buf), ops._p(img.exp),
                                  ops._p(img.amax), n, d, ops._stream())
                report("mlp_pack_input_f16x2(f64 x)", sh, n * (8 * d + 4 * nsl + 8), timeit(mpack, a.iters))
                dsl = (d + 31) // 32 * 32
                report("row_pack_f16x2(hb logmap0)", sh, n * (4 * d + 4 * dsl + 8), timeit(lambda: ops.row_pack(ph, 1, 1.0), a.iters))
            if 128 <= d <= 512 and d % 16 == 0:
                tl = torch.empty_like(x)

                def tail():
                    ops._lib.call("hd_hb_linear_tail_rows_f32", ops._p(x), ops._p(tl), n, d, ops.curv(1.0), 3, ops._stream())
                report("hb_linear_tail_rows(flags=3)", sh, 2 * B, timeit(tail, a.iters))
                xin = hb.proj(hb.expmap0(torch.cat([torch.zeros(n, 1, device=dev), torch.randn(n, 23, device=dev) * 0.3], 1), 1.0), 1.0)
                wt = torch.randn(24, d, device=dev) * 0.1

                def smallk():
                    ops._lib.call("hd_hb_linear_smallk_f32", ops._p(xin), ops._p(wt), ops._p(tl), n, 24, d, ops.curv(1.0), 3,
                                  ops._stream())
                report("hb_linear_smallk fp32-fma (24 -> d, flags=3)", sh, n * 4 * (24 + d) + 4 * 24 * d, timeit(smallk, a.iters))
                wimg = ops.smallk_image_cached(wt.t().contiguous())

                def smallk_mfma():
                    ops._lib.call("hd_hb_linear_smallk_bf16x3_f32", ops._p(xin), ops._p(wimg), ops._p(tl), n, 24, d, ops.curv(1.0), 3,
                                  ops._stream())
                report("hb_linear_smallk bf16x3-mfma (24 -> d, flags=3)", sh, n * 4 * (24 + d) + 4 * 24 * d, timeit(smallk_mfma, a.iters))
                del tl, xin
            del x, p, q, ph, x64, nz, x0, out
            torch.cuda.empty_cache()

        # HypAgg on the power-law graph (config 5 shape) and on the ego set (config 2 shape)
        for tag, (n, d) in (("powerlaw_m5", (a.rows, a.dim)), ("ego_set", (15777, 512))):
            if tag == "powerlaw_m5":
                adj = power_law_graph(n, 5)
                nadj = workloads.normalize_adj(adj).tocoo()
                idx = np.vstack((nadj.row, nadj.col)).astype(np.int64)
                val = nadj.data.astype(np.float32)
            else:
                mats, sizes = workloads.ego_graph_set()
                idx, val, _, _ = workloads.batch_graph_set(mats, sizes, 20)
            csr = ops.csr_from_coo(torch.from_numpy(idx).to(dev), torch.from_numpy(val).to(dev), (n, n))
            nnz = idx.shape[1]
            xt = torch.randn(n, d, device=dev) * 0.1
            alg = 4 * d * (nnz + n) + 8 * nnz + 4 * n
            deg = np.diff(csr.rowptr.cpu().numpy())
            extra = {"nnz": int(nnz), "max_deg": int(deg.max())}
            report(f"spmm_csr[{tag}]", [n, d], alg, timeit(lambda: ops.spmm(csr, xt), a.iters), extra)
            report(f"pb_agg+act[{tag}]", [n, d], alg, timeit(lambda: ops.agg("pb", csr, xt, c, c, 1), a.iters), extra)
            report(f"hb_agg+act[{tag}]", [n, d], alg, timeit(lambda: ops.agg("hb", csr, xt, 1.0, 1.0, 1), a.iters), extra)
            src = torch.from_numpy(idx).to(dev)
            vals = torch.from_numpy(val).to(dev)
            report(f"csr_from_coo[{tag}]", [n, nnz], 20 * nnz + 4 * n,
                   timeit(lambda: ops.csr_from_coo(src, vals, (n, n)), 5), extra)


if __name__ == "__main__":
    main()
