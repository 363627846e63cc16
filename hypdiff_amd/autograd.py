"""Training path of the denoisers: torch.autograd.Function wrappers whose forward AND
backward are HIP kernels of libhypdiff_hip.so (no eager-PyTorch arithmetic on activations).

The reference trains its denoisers with plain autograd over nn.Linear / F.gelu
(diff.py:113-118 -> models/direction_diffusion.py:335-363; diff_graphset.py:64-75 ->
ddpm.py:71-85).  Here one Function covers a whole Linear(+GELU)(+residual) stack:

  forward   per layer one MFMA GEMM; bias, exact-erf GELU and the residual add leave the
            accumulators fused, and the pre-activation z is stored next to the output.
  backward  per layer  dW = dz^T @ x   (hd_gemm_tn_f32: rows split over workgroups, partial
                                        tiles summed in fixed order -> bit-reproducible)
                       db = colsum(dz) (hd_colsum_f32)
                       g  = dz @ W (+ skip gradient), and dz of the layer below =
                            g * gelu'(z_below) from the same accumulators (hd_gemm_f32_ex, epi 3)

PyTorch owns memory, the autograd graph and RNG (and the optimizer unless dist.BucketAdam is used); nothing else.
"""
import os

import torch

from . import _lib, ops

GELU, RES, LIN = 1, 2, 0        # layer kinds = epilogue codes of hd_gemm_f32

# Set by dist.DataParallelTrainer around loss.backward(): an object with view_for(param) -> fp32 view of
# the flat gradient bucket (or None) and delivered(param).  LinearStack.backward then writes a layer's
# dW / db straight into the bucket and reports it, so the bucket's chunk can start its all-reduce while
# the layers below are still in their GEMMs.  Parameters are assumed to enter the graph once per step.
GRAD_SINK = None
# the weight gradients of a chained backward pass as one launch (hd_gemm_tn_limbs_multi_f32); False: one launch per layer
TN_MULTI = os.environ.get("HD_TN_MULTI", "1") != "0"


def _ws(n, dtype, device):
    return torch.empty(max(int(n), 1), dtype=dtype, device=device)


def gemm_ex(a, w, bias, resid, aux, c, c2, epi):
    """hd_gemm_f32_ex / hd_gemm_f32_bf16x3 (ops.GEMM_MODE) on fp32 device tensors; a (M,K), w (N,K); c / c2 may be None."""
    M, K = a.shape
    N = w.shape[0]
    if K % 4 or a.stride(0) % 4 or w.stride(0) % 4 or a.data_ptr() % 16 or w.data_ptr() % 16 or a.stride(1) != 1 \
            or w.stride(1) != 1:
        kp = (K + 3) // 4 * 4
        a, w = ops._pad_k(a, kp), ops._pad_k(w, kp)
    ref = c if c is not None else c2
    ev = ops.GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    sym = ops.gemm_symbol()
    nws = _lib.load().hd_gemm_splitk_workspace_floats(M, N, a.shape[1]) if sym == "hd_gemm_f32_bf16x3" else 0
    if nws > 0:
        # few tiles, long K (the 200-row batches of the graph-level DDPM): the K range is split over workgroups
        ws = _ws(nws, torch.float32, a.device)
        _lib.call(sym + "_splitk", ops._p(a), ops._p(w), ops._p(bias), ops._p(resid), ops._p(aux), ops._p(c), ops._p(c2),
                  M, N, a.shape[1], a.stride(0), w.stride(0), ref.stride(0), epi, ops._p(ws), ops._stream())
    else:
        _lib.call(sym, ops._p(a), ops._p(w), ops._p(bias), ops._p(resid), ops._p(aux), ops._p(c), ops._p(c2),
                  M, N, a.shape[1], a.stride(0), w.stride(0), ref.stride(0), epi, ops._stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N),
                   "gemm_f32_mfma" if ops.GEMM_MODE == "exact" else "gemm_f32_bf16x3"))


def gemm_tn(a, b, out=None, colsum_out=None, want_colsum=False):
    """out[P,Q] = a[R,P]^T @ b[R,Q] (weight gradient).  With `colsum_out` / `want_colsum` the bias gradient
    colsum(a) comes back as well (one launch on the limb kernel, hd_colsum_f32 otherwise): returns (out, colsum)."""
    if colsum_out is not None or want_colsum:
        return _gemm_tn_cs(a, b, out, colsum_out)
    return _gemm_tn_cs(a, b, out, None, False)[0]


def _gemm_tn_cs(a, b, out, cs, want_cs=True):
    R, P = a.shape
    Q = b.shape[1]
    if P % 4 or Q % 4 or a.stride(0) % 4 or b.stride(0) % 4 or a.data_ptr() % 16 or b.data_ptr() % 16:
        # odd widths (e.g. the D = 14 latents of config 1): zero-padded copies, result sliced back
        res = gemm_tn(ops._pad_k(a, (P + 3) // 4 * 4), ops._pad_k(b, (Q + 3) // 4 * 4))[:P, :Q]
        if out is None:
            out = res.contiguous()
        else:
            out.copy_(res)
        return out, (colsum(a, out=cs) if want_cs else None)
    if out is not None and (out.data_ptr() % 16 or not out.is_contiguous()):
        res, cs = _gemm_tn_cs(a, b, None, cs, want_cs)        # the kernel stores 16-byte rows
        out.copy_(res)
        return out, cs
    if out is None:
        out = torch.empty(P, Q, dtype=torch.float32, device=a.device)
    # fp32-faithful limb arithmetic on the bf16 matrix cores unless GEMM_MODE asks for the exact fp32 MFMA
    limbs = ops.GEMM_MODE != "exact" and out.stride(0) % 4 == 0
    sym = "hd_gemm_tn_bf16x3" if limbs else "hd_gemm_tn"
    ws = _ws(getattr(_lib.load(), sym + "_workspace_floats")(R, P, Q), torch.float32, a.device)
    fused = limbs and want_cs and (cs is None or (cs.data_ptr() % 16 == 0 and cs.is_contiguous()))
    if fused and cs is None:
        cs = torch.empty(P, dtype=torch.float32, device=a.device)
    ev = ops.GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if limbs:
        _lib.call(sym + "_f32", ops._p(a), ops._p(b), ops._p(out), ops._p(cs) if fused else None, R, P, Q, a.stride(0),
                  b.stride(0), out.stride(0), ops._p(ws), ops._stream())
    else:
        _lib.call(sym + "_f32", ops._p(a), ops._p(b), ops._p(out), R, P, Q, a.stride(0), b.stride(0), out.stride(0),
                  ops._p(ws), ops._stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, 2.0 * R * P * Q, 4.0 * (R * P + R * Q + P * Q)))
    if want_cs and not fused:
        cs = colsum(a, out=cs)
    return out, cs


def gemm_tn_limbs(a_img, b_img, out=None, colsum_out=None):
    """(out[P,Q], colsum[P]) = (a^T @ b, colsum(a)) from the bf16x3 limb images of a (R, P) and b (R, Q)
    (hd_gemm_tn_limbs_f32).  P, Q are the image widths rounded up to multiples of 4 (pad columns are zero in an image)."""
    R = a_img.rows
    assert b_img.rows == R and a_img.fmt == b_img.fmt == "bf16x3"
    P, Q = (a_img.cols + 3) // 4 * 4, (b_img.cols + 3) // 4 * 4
    dev = a_img.buf.device
    if out is None or out.data_ptr() % 16 or not out.is_contiguous() or tuple(out.shape) != (P, Q):
        res, cs = gemm_tn_limbs(a_img, b_img, torch.empty(P, Q, dtype=torch.float32, device=dev), None)
        if out is None:
            return res, cs
        out.copy_(res[: out.shape[0], : out.shape[1]])
        if colsum_out is not None:
            colsum_out.copy_(cs[: colsum_out.shape[0]])
        return out, colsum_out if colsum_out is not None else cs
    own_cs = colsum_out is None or colsum_out.data_ptr() % 16 or not colsum_out.is_contiguous() or colsum_out.numel() != P
    cs = torch.empty(P, dtype=torch.float32, device=dev) if own_cs else colsum_out
    ws = _ws(_lib.load().hd_gemm_tn_bf16x3_workspace_floats(R, P, Q), torch.float32, dev)
    ev = ops.GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("hd_gemm_tn_limbs_f32", ops._p(a_img.buf), ops._p(b_img.buf), ops._p(out), ops._p(cs), R, P, Q, out.stride(0),
              ops._p(ws), ops._stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, 2.0 * R * P * Q, 6.0 * (R * P + R * Q) + 4.0 * P * Q))
    if own_cs and colsum_out is not None:
        colsum_out.copy_(cs[: colsum_out.numel()])
        cs = colsum_out
    return out, cs


def gemm_tn_limbs_multi(jobs):
    """[(a^T @ b, colsum(a)) for (a_img, b_img, out, colsum_out) in jobs] in ONE launch (hd_gemm_tn_limbs_multi_f32):
    products over the same rows that do not depend on each other -- the weight gradients of a backward pass.  Every
    product is cut and summed exactly as gemm_tn_limbs cuts and sums it (bit-identical).  out / colsum_out: fp32
    tensors of the padded shapes (P, Q) / (P,), contiguous and 16-byte aligned, or None (allocated here)."""
    if not jobs:
        return []
    R = jobs[0][0].rows
    dev = jobs[0][0].buf.device
    res = []
    arr = (_lib.HdTnJob * len(jobs))()
    flops = nbytes = 0.0
    for i, (a_img, b_img, out, cs) in enumerate(jobs):
        assert a_img.rows == R and b_img.rows == R and a_img.fmt == b_img.fmt == "bf16x3"
        P, Q = (a_img.cols + 3) // 4 * 4, (b_img.cols + 3) // 4 * 4
        fits = out is not None and out.data_ptr() % 16 == 0 and out.is_contiguous() and tuple(out.shape) == (P, Q)
        o = out if fits else torch.empty(P, Q, dtype=torch.float32, device=dev)
        cfits = cs is not None and cs.data_ptr() % 16 == 0 and cs.is_contiguous() and cs.numel() == P
        c = cs if cfits else torch.empty(P, dtype=torch.float32, device=dev)
        arr[i] = _lib.HdTnJob(ops._p(a_img.buf), ops._p(b_img.buf), ops._p(o), ops._p(c), P, Q, o.stride(0))
        res.append((o, c, out, cs, fits, cfits))
        flops += 2.0 * R * P * Q
        nbytes += 6.0 * (R * P + R * Q) + 4.0 * P * Q
    ws = _ws(_lib.load().hd_gemm_tn_limbs_multi_workspace_floats(R, arr, len(jobs)), torch.float32, dev)
    ev = ops.GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("hd_gemm_tn_limbs_multi_f32", R, arr, len(jobs), ops._p(ws), ops._stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, flops, nbytes))
    outs = []
    for o, c, out, cs, fits, cfits in res:
        if out is not None and not fits:
            out.copy_(o[: out.shape[0], : out.shape[1]])
            o = out
        if cs is not None and not cfits:
            cs.copy_(c[: cs.numel()])
            c = cs
        outs.append((o, c))
    return outs


def transpose(w):
    out = torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=w.device)
    _lib.call("hd_transpose_f32", ops._p(w), ops._p(out), w.shape[0], w.shape[1], w.stride(0), out.stride(0),
              ops._stream())
    return out


def transpose_multi(ws):
    """[w^T for w in ws] in one launch per 64 matrices (hd_transpose_multi_f32)."""
    outs, jobs = [], []
    for w in ws:
        o = torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=w.device)
        outs.append(o)
        jobs.append(_lib.HdTransposeJob(w.data_ptr(), o.data_ptr(), w.shape[0], w.shape[1], w.stride(0), o.stride(0)))
    for i in range(0, len(jobs), 64):
        part = jobs[i:i + 64]
        _lib.call("hd_transpose_multi_f32", (_lib.HdTransposeJob * len(part))(*part), len(part), ops._stream())
    return outs


def colsum(x, out=None):
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _ws(_lib.load().hd_colsum_workspace_floats(M, N), torch.float32, x.device)
    _lib.call("hd_colsum_f32", ops._p(x), ops._p(out), M, N, x.stride(0), ops._p(ws), ops._stream())
    return out


def gemm_limbs_train(a, w, bias, aux, epi, want_z=False, want_f32=True, want_limbs=False):
    """hd_gemm_limbs_train_f32 on bf16x3 limb images: returns (z, c, cl).  Forward (epi 0 / 1): z = a w^T + bias
    (kept for the backward pass), c / cl = the activated value as fp32 / as the next layer's operand image.
    Backward (epi 3): a = image of dz, w = image of W^T, aux = z of the layer below: c / cl = (dz W) * gelu'(aux)."""
    M, K, N = a.rows, a.cols, w.rows
    assert w.cols == K, ((a.rows, a.cols), (w.rows, w.cols))
    dev = a.buf.device
    # fp32 rows padded to 64 bytes: a lane quartet's 64-byte piece of a row is then one whole sector for the
    # non-temporal stores of the write-out (N = 900, 700: unaligned rows measured 25 % slower launches)
    ld = (N + 15) // 16 * 16
    z = torch.empty(M, ld, dtype=torch.float32, device=dev)[:, :N] if want_z else None
    c = torch.empty(M, ld, dtype=torch.float32, device=dev)[:, :N] if want_f32 else None
    cl = ops.limb_empty(M, N, dev, "bf16x3") if want_limbs else None
    assert aux is None or (tuple(aux.shape) == (M, N) and aux.stride() == (ld, 1) and aux.data_ptr() % 16 == 0)
    ev = ops.GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("hd_gemm_limbs_train_f32", ops._p(a.buf), ops._p(w.buf), ops._p(bias), ops._p(aux), ops._p(z), ops._p(c),
              ops._p(cl.buf) if cl is not None else None, M, N, K, ld, epi, ops._stream())
    if ev is not None:
        e1.record()
        kp, npad = (K + 31) // 32 * 32, (N + 31) // 32 * 32
        ev.append((e0, e1, 2.0 * M * N * K, 6.0 * (M * kp + N * kp) + 4.0 * M * N * ((z is not None) + (c is not None)
                   + (aux is not None)) + (6.0 * M * npad if cl is not None else 0.0), "gemm_limbs"))
    return z, c, cl


def _limb_stack_ok(x, kinds, ws_):
    """The stack can run on the persistent limb GEMM: fp32-faithful limb arithmetic allowed (not GEMM_MODE "exact"),
    no residual layers (the mlp denoiser: GELU ... GELU, LIN), rows of 16-byte multiples, no gradient for x."""
    # (every layer but the last is GELU: the input-gradient GEMM of _backward_limbs multiplies by gelu'(z_below) in its
    # write-out; a LIN layer in the middle has no z_below and takes the on-the-fly kernels instead)
    return (ops.GEMM_MODE != "exact" and not x.requires_grad and all(k == GELU for k in kinds[:-1]) and kinds[-1] == LIN
            and x.shape[0] > 0 and all(w.shape[0] % 4 == 0 for w in ws_))


class LinearStack(torch.autograd.Function):
    """y = stack(x): layer l computes z = x_l W_l^T + b_l and x_{l+1} = z (LIN), gelu(z)
    (GELU) or x_l + gelu(z) (RES).  `x` is a contiguous fp32 (M, K0) tensor with K0 % 4 == 0;
    W_0 may be narrower than K0 (the packed, zero-padded cat(x, t) input of the mlp denoiser):
    it is zero-padded for the GEMM and its gradient is sliced back."""

    @staticmethod
    def forward(ctx, x, kinds, *params):
        ops._dev(x)
        ws_, bs_ = params[0::2], params[1::2]
        assert len(ws_) == len(kinds) == len(bs_)
        ctx.limbs = _limb_stack_ok(x, kinds, ws_)
        if ctx.limbs:
            # operands travel as bf16x3 limb images (split once per element by the producing GEMM's write-out) and the
            # backward pass reads the same images (the weight gradient contracts over image rows): the only fp32
            # activations kept are the pre-activations z (gelu')
            zs, a_imgs = [], []
            # the input, every W_l and (for the input gradients of the backward pass) every W_l^T above the first
            # layer become limb images in ONE launch: the weights change every step, so nothing can be cached
            L = len(kinds)
            train = any(p.requires_grad for p in params)
            imgs = ops.limb_pack_multi([(x[:, : ws_[0].shape[1]], False)] + [(w.detach(), False) for w in ws_] +
                                       ([(w.detach(), True) for w in ws_[1:]] if train else []))
            a, wimgs = imgs[0], imgs[1:1 + L]
            ctx.wt_imgs = [None] + imgs[1 + L:] if train else None
            h = None
            ctx.chain = ops.mlp_chain_train_ok(a, L)
            if ctx.chain:
                # all layers in ONE persistent launch (csrc/gemm_chain.hip): same tiles' arithmetic as the per-layer
                # launches, bit-identical outputs
                outs = ops._mlp_chain_launch(a, [
                    {"w": wimgs[l], "bias": bs_[l].detach(), "epi": 1 if kind == GELU else 0, "img": l < L - 1,
                     "f32": l == L - 1, "z": kind == GELU and train} for l, kind in enumerate(kinds)])
                a_imgs = [a] + [o[0] for o in outs[:-1]]
                zs = [o[2] for o in outs]
                h = outs[-1][1]
            else:
                for l, kind in enumerate(kinds):
                    last = l == L - 1
                    a_imgs.append(a)
                    z, h, a = gemm_limbs_train(a, wimgs[l], bs_[l].detach(), None, 1 if kind == GELU else 0,
                                               want_z=kind == GELU and train, want_f32=last, want_limbs=not last)
                    zs.append(z)
            ctx.x_imgs = a_imgs if train else None
            ctx.kinds = kinds
            ctx.params = params
            ctx.need_x = False
            ctx.wshapes = [tuple(w.shape) for w in ws_]
            ctx.save_for_backward(*[z for z in zs if z is not None])
            ctx.nz = [z is not None for z in zs]
            return h
        xs, zs, wpad = [], [], []
        h = x
        for l, kind in enumerate(kinds):
            w = ws_[l].detach()
            if w.shape[1] != h.shape[1] or w.stride(0) % 4 or w.data_ptr() % 16:
                w = ops._pad_k(w, h.shape[1])
            wpad.append(w)
            out = torch.empty(h.shape[0], w.shape[0], dtype=torch.float32, device=h.device)
            z = torch.empty_like(out) if kind != LIN else None
            gemm_ex(h, w, bs_[l].detach(), h if kind == RES else None, None, out, z, kind)
            xs.append(h)
            zs.append(z)
            h = out
        ctx.kinds = kinds
        ctx.params = params            # the parameter objects themselves (GRAD_SINK looks them up by identity)
        ctx.need_x = x.requires_grad
        ctx.wshapes = [tuple(w.shape) for w in ws_]
        ctx.save_for_backward(*xs, *[z for z in zs if z is not None], *wpad)
        ctx.nz = [z is not None for z in zs]
        return h

    @staticmethod
    def backward(ctx, gy):
        kinds, L = ctx.kinds, len(ctx.kinds)
        if ctx.limbs:
            return LinearStack._backward_limbs(ctx, gy)
        saved = list(ctx.saved_tensors)
        xs = saved[:L]
        nzc = sum(ctx.nz)
        zl = saved[L:L + nzc]
        ws_ = saved[L + nzc:]
        zs, it = [], iter(zl)
        for has in ctx.nz:
            zs.append(next(it) if has else None)
        gy = gy.contiguous()
        grads = [None] * (2 * L)
        # dz of the top layer
        top = L - 1
        if kinds[top] == LIN:
            dz, g_up = gy, None
        else:       # stack ending in an activation: dz = gy * gelu'(z) through the same epilogue
            raise NotImplementedError("LinearStack: the last layer must be linear")
        lo = 0 if ctx.need_x else 1
        wts = [None] * lo + transpose_multi(ws_[lo:])     # (K_l, N_l): contraction over N_l; every W^T in one launch
        for l in range(top, -1, -1):
            x_l, w = xs[l], ws_[l]
            shp = ctx.wshapes[l]
            sink = GRAD_SINK
            wp, bp = ctx.params[2 * l], ctx.params[2 * l + 1]
            wv = sink.view_for(wp) if sink is not None else None
            bv = sink.view_for(bp) if sink is not None else None
            if wv is not None and bv is not None:
                if shp[1] == x_l.shape[1]:
                    gemm_tn(dz, x_l, out=wv, colsum_out=bv)
                else:                                       # zero-padded first layer: slice the pad columns off
                    wv.copy_(gemm_tn(dz, x_l, colsum_out=bv)[0][:, :shp[1]])
                sink.delivered(bp)
                sink.delivered(wp)
            else:
                dw, grads[2 * l + 1] = gemm_tn(dz, x_l, want_colsum=True)
                grads[2 * l] = dw if tuple(dw.shape) == shp else dw[:, :shp[1]].contiguous()
            if l == 0 and not ctx.need_x:
                break
            wt = wts[l]
            M, Kl = x_l.shape
            resid = g_up if kinds[l] == RES else None
            if l == 0:
                gx = torch.empty(M, Kl, dtype=torch.float32, device=gy.device)
                gemm_ex(dz, wt, None, resid, None, gx, None, 3)
                g_up = gx
                break
            keep_g = kinds[l - 1] == RES
            g = torch.empty(M, Kl, dtype=torch.float32, device=gy.device) if keep_g else None
            dz_below = torch.empty(M, Kl, dtype=torch.float32, device=gy.device)
            gemm_ex(dz, wt, None, resid, zs[l - 1], g, dz_below, 3)
            g_up, dz = g, dz_below
        gx = g_up if ctx.need_x else None
        return (gx, None, *grads)

    @staticmethod
    def _backward_limbs(ctx, gy):
        """Backward pass on limb images only: dz travels down as an image (written by the input-gradient launch's
        write-out, (dz @ W_l) * gelu'(z_below)); dW_l = dz^T x_l and db_l = colsum(dz) come from the dz image and the
        image of x_l that the forward pass fed to layer l (hd_gemm_tn_limbs_f32), straight into the gradient bucket
        when one is armed."""
        kinds, L = ctx.kinds, len(ctx.kinds)
        zl = iter(ctx.saved_tensors)
        zs = [next(zl) if has else None for has in ctx.nz]
        if kinds[L - 1] != LIN:
            raise NotImplementedError("LinearStack: the last layer must be linear")
        grads = [None] * (2 * L)
        dz_img = ops.limb_pack(gy.contiguous(), "bf16x3")
        sink = GRAD_SINK
        dz_imgs = None
        if ctx.chain and L >= 3:
            # the input-gradient GEMMs dz_{l-1} = (dz_l W_l) gelu'(z_{l-1}), l = L-1 .. 1, are row-parallel like the
            # forward pass: one persistent launch; the weight-gradient GEMMs then read the dz images it left
            outs = ops._mlp_chain_launch(dz_img, [{"w": ctx.wt_imgs[l], "epi": 3, "img": True, "aux": zs[l - 1]}
                                                  for l in range(L - 1, 0, -1)])
            dz_imgs = {L - 1: dz_img}
            for k, l in enumerate(range(L - 1, 0, -1)):
                dz_imgs[l - 1] = outs[k][0]
        if dz_imgs is not None and TN_MULTI:
            # every dz image exists: the L weight gradients are independent products over the same rows -- one launch
            # (two when a gradient bucket has all-reduces to overlap: the top layers' chunks go on the wire while the
            # lower layers' gradients are still being multiplied)
            order = list(range(L - 1, -1, -1))
            split = sink is not None and sink.wants_overlap() and L >= 4
            for grp in ([order[:2], order[2:]] if split else [order]):
                jobs, views = [], []
                for l in grp:
                    wp, bp = ctx.params[2 * l], ctx.params[2 * l + 1]
                    wv = sink.view_for(wp) if sink is not None else None
                    bv = sink.view_for(bp) if sink is not None else None
                    if wv is None or bv is None:
                        wv = bv = None
                    jobs.append((dz_imgs[l], ctx.x_imgs[l], wv, bv))
                    views.append(wv is not None)
                for l, (dw, db), in_bucket in zip(grp, gemm_tn_limbs_multi(jobs), views):
                    if in_bucket:
                        sink.delivered(ctx.params[2 * l + 1])
                        sink.delivered(ctx.params[2 * l])
                    else:
                        shp = ctx.wshapes[l]
                        grads[2 * l] = dw if tuple(dw.shape) == shp else dw[: shp[0], : shp[1]].contiguous()
                        grads[2 * l + 1] = db if db.numel() == shp[0] else db[: shp[0]].contiguous()
            ctx.x_imgs = ctx.wt_imgs = None
            return (None, None, *grads)
        for l in range(L - 1, -1, -1):
            if dz_imgs is not None:
                dz_img = dz_imgs[l]
            shp = ctx.wshapes[l]
            wp, bp = ctx.params[2 * l], ctx.params[2 * l + 1]
            wv = sink.view_for(wp) if sink is not None else None
            bv = sink.view_for(bp) if sink is not None else None
            if wv is not None and bv is not None:
                gemm_tn_limbs(dz_img, ctx.x_imgs[l], out=wv, colsum_out=bv)
                sink.delivered(bp)
                sink.delivered(wp)
            else:
                dw, db = gemm_tn_limbs(dz_img, ctx.x_imgs[l])
                grads[2 * l] = dw if tuple(dw.shape) == shp else dw[: shp[0], : shp[1]].contiguous()
                grads[2 * l + 1] = db if db.numel() == shp[0] else db[: shp[0]].contiguous()
            if l == 0:
                break
            if dz_imgs is None:
                _, _, dz_img = gemm_limbs_train(dz_img, ctx.wt_imgs[l], None, zs[l - 1], 3, want_f32=False, want_limbs=True)
        ctx.x_imgs = ctx.wt_imgs = None
        return (None, None, *grads)


def linear_stack(x, layers, kinds):
    """layers: list of nn.Linear; kinds: tuple of GELU/RES/LIN."""
    params = []
    for lyr in layers:
        params += [lyr.weight, lyr.bias]
    return LinearStack.apply(x, tuple(kinds), *params)


class EmbeddingAdd(torch.autograd.Function):
    """h + table[idx] (ddpm.py:39-41).  Backward: the table gradient is a segment sum of the
    rows of dL/dh grouped by idx (hd_embedding_grad_f32: a device sort, then every table row sums its run in
    ascending row order -- deterministic, no atomics, nothing read back to the host)."""

    @staticmethod
    def forward(ctx, h, table, idx):
        if h.is_contiguous():
            out = h
            ctx.mark_dirty(h)
        else:
            out = h.contiguous()
        tab = table.detach()
        _lib.call("hd_embedding_add", ops._p(out), ops._p(tab), ops._p(idx), tab.shape[0], out.shape[0], out.shape[1],
                  ops._stream())
        ctx.save_for_backward(idx)
        ctx.T = tab.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        order = torch.sort(idx, stable=True)
        gt = torch.empty(ctx.T, g.shape[1], dtype=torch.float32, device=g.device)
        _lib.call("hd_embedding_grad_f32", ops._p(g), ops._p(order.values), ops._p(order.indices.contiguous()), g.shape[0],
                  g.shape[1], ctx.T, ops._p(gt), ops._stream())
        return g, gt, None


class MSELoss(torch.autograd.Function):
    """F.mse_loss(a, b) / (a - b).square().mean() with b a constant target."""

    @staticmethod
    def forward(ctx, a, b):
        ops._dev(a, b)
        a, b = a.contiguous(), b.detach().to(torch.float32).contiguous()
        assert a.shape == b.shape and a.dtype == torch.float32
        d = torch.empty_like(a)
        ws = _ws(_lib.load().hd_mse_workspace_doubles(), torch.float64, a.device)
        loss = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.call("hd_mse_forward_f32", ops._p(a), ops._p(b), ops._p(d), a.numel(), ops._p(ws), ops._p(loss),
                  ops._stream())
        ctx.save_for_backward(d)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        (d,) = ctx.saved_tensors
        gl = gl.to(torch.float32).reshape(1).contiguous()
        out = torch.empty_like(d)
        _lib.call("hd_mse_backward_f32", ops._p(d), ops._p(gl), ops._p(out), d.numel(), ops._stream())
        return out, None


def mse_loss(a, b):
    return MSELoss.apply(a, b)


def wants_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def chain_wants_grad(chain, *tensors):
    """wants_grad over the weights and biases of a [(weight, bias, epi), ...] list and further tensors.  The callers used
    to pass *module.parameters(): a walk of the module tree, 25-60 us per call and several calls per step, paid in
    inference too because the arguments are built before wants_grad can look at torch.is_grad_enabled()."""
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or \
        any((w is not None and w.requires_grad) or (b is not None and b.requires_grad) for w, b, _ in chain)


# ---- encoder (HGCN / HNN) training: lp_train.py:66-82 --------------------------------------
MAP_CODES = {"proj": 0, "expmap0": 1, "logmap0": 2, "expmap0_proj": 3}


def _f32_only(x, what):
    if x.dtype != torch.float32:
        raise NotImplementedError(f"{what}: the backward kernels are fp32 (the reference trains in fp32); "
                                  f"got {x.dtype} -- run fp64 under torch.no_grad()")


class RowMap(torch.autograd.Function):
    """proj / expmap0 / logmap0 / input lift of either manifold with its VJP kernel."""

    @staticmethod
    def forward(ctx, x, abi, op, cv):
        x2, shape = ops._rows(x)
        out = torch.empty_like(x2)
        _lib.call(f"hd_{abi}_{op}" + ops._sfx(x2), ops._p(x2), ops._p(out), x2.shape[0], x2.shape[1], cv, ops._stream())
        ctx.save_for_backward(x2)
        ctx.meta = (abi, MAP_CODES[op], cv, shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        (x2,) = ctx.saved_tensors
        abi, which, cv, shape = ctx.meta
        g2 = g.reshape(x2.shape).contiguous()
        out = torch.empty_like(x2)
        _lib.call(f"hd_{abi}_map_bwd" + ops._sfx(x2), which, ops._p(x2), ops._p(g2), ops._p(out), x2.shape[0],
                  x2.shape[1], cv, ops._stream())
        return out.view(shape), None, None, None


class RowMap2(torch.autograd.Function):
    """mobius_add / expmap / logmap of the Poincare ball on two row operands of equal shape (kernel argument order:
    ops.map2) with the VJP kernel hd_pb_map2_bwd (both gradients in one pass; _lambda_x carries no gradient, as in
    manifolds/poincare.py:34)."""

    CODES = {"hd_pb_mobius_add": 0, "hd_pb_expmap": 1, "hd_pb_logmap": 2}

    @staticmethod
    def forward(ctx, a, b, name, cv):
        a2, shape = ops._rows(a)
        b2, _ = ops._rows(b)
        out = torch.empty_like(a2)
        _lib.call(name + ops._sfx(a2), ops._p(a2), ops._p(b2), ops._p(out), a2.shape[0], a2.shape[1], cv, ops._stream())
        ctx.save_for_backward(a2, b2)
        ctx.meta = (RowMap2.CODES[name], cv, shape)
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        which, cv, shape = ctx.meta
        g2 = g.reshape(a2.shape).contiguous()
        da, db = torch.empty_like(a2), torch.empty_like(b2)
        _lib.call("hd_pb_map2_bwd" + ops._sfx(a2), which, ops._p(a2), ops._p(b2), ops._p(g2), ops._p(da), ops._p(db),
                  a2.shape[0], a2.shape[1], cv, ops._stream())
        return da.view(shape), db.view(shape), None, None


class SqdistFn(torch.autograd.Function):
    """Manifold.sqdist of two row operands (one scalar per row) with hd_{pb,hb}_sqdist_bwd"""

    @staticmethod
    def forward(ctx, a, b, name, cv):
        a2, shape = ops._rows(a)
        b2, _ = ops._rows(b)
        out = torch.empty(a2.shape[0], dtype=a2.dtype, device=a2.device)
        _lib.call(name + ops._sfx(a2), ops._p(a2), ops._p(b2), ops._p(out), a2.shape[0], a2.shape[1], cv, ops._stream())
        ctx.save_for_backward(a2, b2)
        ctx.meta = (name, cv, shape)
        return out.view(shape[:-1])

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        name, cv, shape = ctx.meta
        g1 = g.reshape(-1).contiguous()
        da, db = torch.empty_like(a2), torch.empty_like(b2)
        _lib.call(name + "_bwd" + ops._sfx(a2), ops._p(a2), ops._p(b2), ops._p(g1), ops._p(da), ops._p(db), a2.shape[0],
                  a2.shape[1], cv, ops._stream())
        return da.view(shape), db.view(shape), None, None


class HbProjTan0(torch.autograd.Function):
    """hyperboloid.py:64-68: zero the time column; the VJP is the same map on the gradient"""

    @staticmethod
    def forward(ctx, u, cv):
        ctx.cv = cv
        return HbProjTan0._run(u, cv)

    @staticmethod
    def _run(u, cv):
        u2 = u.contiguous()
        out = torch.empty_like(u2)
        _lib.call("hd_hb_proj_tan0" + ops._sfx(u2), ops._p(u2), ops._p(out), u2.shape[0], u2.shape[1], cv, ops._stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return HbProjTan0._run(g, ctx.cv), None


class HypLinearFn(torch.autograd.Function):
    """proj(mobius_matvec(W, x)) [+ logmap0] (layers/hyp_layers.py:98-101, :134) and its VJP:
    one row kernel for the tail, the TN MFMA GEMM for dW, the NT GEMM for dx."""

    @staticmethod
    def forward(ctx, x, w, abi, cv, flags):
        _f32_only(x, "HypLinear autograd")
        x2 = x.contiguous()
        wd = w.detach().to(torch.float32)
        if abi == "pb":
            lhs = x2
            aux = ops.rownorm(x2)
        else:
            lhs = torch.empty_like(x2)
            _lib.call("hd_hb_logmap0_f32", ops._p(x2), ops._p(lhs), x2.shape[0], x2.shape[1], cv, ops._stream())
            aux = None
        mx = ops.gemm(lhs, wd)
        out = torch.empty_like(mx)
        if abi == "pb":
            _lib.call("hd_pb_matvec_epilogue_f32", ops._p(mx), ops._p(aux), ops._p(out), mx.shape[0], mx.shape[1], cv,
                      flags, ops._stream())
            ctx.save_for_backward(x2, wd, mx, aux)
        else:
            _lib.call("hd_hb_linear_epilogue_f32", ops._p(mx), ops._p(out), mx.shape[0], mx.shape[1], cv, flags,
                      ops._stream())
            ctx.save_for_backward(x2, wd, mx, lhs)
        ctx.meta = (abi, cv, flags, x.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        abi, cv, flags, need_x = ctx.meta
        x2, wd, mx, aux = ctx.saved_tensors
        g = g.contiguous()
        n, dout = mx.shape
        dmx = torch.empty_like(mx)
        if abi == "pb":
            dxn = torch.empty(n, dtype=torch.float32, device=g.device)
            _lib.call("hd_pb_linear_tail_bwd_f32", ops._p(mx), ops._p(aux), ops._p(g), ops._p(dmx), ops._p(dxn), n, dout,
                      cv, flags, ops._stream())
            lhs = x2
        else:
            _lib.call("hd_hb_linear_tail_bwd_f32", ops._p(mx), ops._p(g), ops._p(dmx), n, dout, cv, flags, ops._stream())
            lhs = aux
        dw = gemm_tn(dmx, lhs)
        dx = None
        if need_x:
            dl = ops.gemm(dmx, transpose(wd))                 # (n, d_in) = dmx @ W
            dx = torch.empty_like(x2)
            if abi == "pb":
                _lib.call("hd_rownorm_bwd_add_f32", ops._p(x2), ops._p(dxn), ops._p(dl), ops._p(dx), n, x2.shape[1],
                          ops._stream())
            else:
                _lib.call("hd_hb_map_bwd_f32", 2, ops._p(x2), ops._p(dl), ops._p(dx), n, x2.shape[1], cv, ops._stream())
        return dx, dw, None, None, None


class HypAggFn(torch.autograd.Function):
    """[HypAct](proj(expmap0(spmm(adj, xt)))) (layers/hyp_layers.py:150-151, :170-173).  Training
    forward = hd_spmm_csr (aggregate s kept) + the tail kernel; backward = tail VJP + SpMM with
    the transposed CSR (rows summed in index order: deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, xt, csr, abi, cin, cout, act_code, fuse):
        _f32_only(xt, "HypAgg autograd")
        s = ops.spmm(csr, xt)
        out = torch.empty_like(s)
        _lib.call(f"hd_{abi}_agg_tail_f32", ops._p(s), ops._p(out), s.shape[0], s.shape[1], cin, cout, act_code, fuse,
                  ops._stream())
        ctx.save_for_backward(s)
        ctx.meta = (csr, abi, cin, cout, act_code, fuse)
        return out

    @staticmethod
    def backward(ctx, g):
        (s,) = ctx.saved_tensors
        csr, abi, cin, cout, act_code, fuse = ctx.meta
        g = g.contiguous()
        ds = torch.empty_like(s)
        _lib.call(f"hd_{abi}_agg_tail_bwd_f32", ops._p(s), ops._p(g), ops._p(ds), s.shape[0], s.shape[1], cin, cout,
                  act_code, fuse, ops._stream())
        return ops.spmm(csr.transposed(), ds), None, None, None, None, None, None


class EdgeDecodeFn(torch.autograd.Function):
    """LPModel.decode (models/base_models.py:152-159): gather both endpoints, sqdist and the
    Fermi-Dirac map in one kernel; backward = one VJP kernel writing the two endpoint
    gradients of every pair, summed per node in pair order by hd_embedding_grad_f32 (no atomics, no read-back)."""

    @staticmethod
    def forward(ctx, h, idx, abi, cv, r, t):
        _f32_only(h, "LPModel.decode autograd")
        h = h.contiguous()
        out = torch.empty(idx.shape[0], dtype=h.dtype, device=h.device)
        _lib.call(f"hd_{abi}_edge_decode_f32", ops._p(h), ops._p(idx), h.shape[0], ops._p(out), idx.shape[0], h.shape[1],
                  cv, 1, r, t, ops._stream())
        ctx.save_for_backward(h, idx)
        ctx.meta = (abi, cv, r, t)
        return out

    @staticmethod
    def backward(ctx, g):
        h, idx = ctx.saved_tensors
        abi, cv, r, t = ctx.meta
        E = idx.shape[0]
        g = g.reshape(-1).contiguous()
        ge = torch.empty(2 * E, h.shape[1], dtype=h.dtype, device=h.device)
        _lib.call(f"hd_{abi}_edge_decode_bwd_f32", ops._p(h), ops._p(idx), h.shape[0], ops._p(g), ops._p(ge), E, h.shape[1],
                  cv, 1, r, t, ops._stream())
        # per-node sum of the 2E endpoint gradients in ascending pair order: a device sort and one segment-sum launch
        # (the kernel of the embedding-table gradient), nothing read back to the host
        node = torch.cat([idx[:, 0], idx[:, 1]])
        order = torch.sort(node, stable=True)
        gh = torch.empty_like(h)
        _lib.call("hd_embedding_grad_f32", ops._p(ge), ops._p(order.values), ops._p(order.indices.contiguous()), 2 * E,
                  h.shape[1], h.shape[0], ops._p(gh), ops._stream())
        return gh, None, None, None, None, None


# ---- gcndec denoiser (models/direction_diffusion.py:17-44) -----------------------------------
class GCNConvGelu(torch.autograd.Function):
    """act(A_norm @ (x W^T) + bias) with A_norm = D^-1/2 (A + I) D^-1/2 as a device CSR (rows =
    target nodes).  PARITY UNPINNED: torch_geometric.nn.GCNConv is not in the reference tree; this
    follows its published definition (Kipf & Welling; linear first, bias after the propagation)."""

    @staticmethod
    def forward(ctx, x, w, b, csr, act):
        x = x.contiguous()
        wd, bd = w.detach(), b.detach()
        xw = ops.gemm(x, wd)
        agg = ops.spmm(csr, xw)
        out = torch.empty_like(agg)
        z = torch.empty_like(agg) if (any(ctx.needs_input_grad) and act) else None
        _lib.call("hd_bias_gelu_f32", ops._p(agg), ops._p(bd), ops._p(out), ops._p(z), agg.shape[0], agg.shape[1],
                  1 if act else 0, ops._stream())
        ctx.save_for_backward(x, wd, z)
        ctx.meta = (csr, act, x.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wd, z = ctx.saved_tensors
        csr, act, need_x = ctx.meta
        g = g.contiguous()
        if act:
            dz = torch.empty_like(g)
            _lib.call("hd_gelu_bwd_f32", ops._p(g), ops._p(z), ops._p(dz), g.numel(), ops._stream())
        else:
            dz = g
        db = colsum(dz)
        dxw = ops.spmm(csr.transposed(), dz)
        dw = gemm_tn(dxw, x)
        dx = ops.gemm(dxw, transpose(wd)) if need_x else None
        return dx, dw, db, None, None
