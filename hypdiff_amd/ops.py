"""Tensor-level wrappers over the C ABI (device pointers in, device tensors out).

PyTorch is plumbing here: it owns the device memory and the stream.  Every function
enqueues HIP kernels of libhypdiff_hip.so on the current stream and returns without
synchronising.  Inputs must live on the GPU; nothing here computes on the CPU.
"""
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import HdCurv

ACT_CODES = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "elu": 4, "leaky_relu": 5}

_curv_cache = {}
GEMM_EVENTS = None      # set to a list by bench.py to time every hd_gemm_f32 launch
# fp32 GEMM kernels.  "bf16x3" (default) = fp32-faithful 3-limb split on the bf16 matrix cores: every
# operand element keeps its own 24 significant bits, 6 exact limb products per fp32 product
# (csrc/gemm_limbs.hip, csrc/gemm_split.hip).  "exact" = v_mfma_f32_32x32x2_f32 (csrc/gemm.hip), bit
# for bit an fmaf chain.  "f16x2" = opt-in: inference GEMMs carry two fp16 limbs of the row scaled by
# a power of two (22 bits relative to the ROW maximum -- block floating point, narrower than IEEE
# fp32 element by element, although the contraction's error stays inside the fp32 bound), 3 limb
# products per fp32 product; residual chains and training stay on bf16x3.
GEMM_MODE = os.environ.get("HD_GEMM_MODE", "bf16x3")


def gemm_symbol():
    if GEMM_MODE in ("bf16x3", "f16x2"):
        return "hd_gemm_f32_bf16x3"
    if GEMM_MODE == "exact":
        return "hd_gemm_f32_ex"
    raise ValueError("hypdiff_amd.ops.GEMM_MODE must be 'bf16x3', 'f16x2' or 'exact', not %r" % (GEMM_MODE,))


def curv(c):
    """hd_curv for a python-float or 1-element-tensor curvature.

    Precision rules of the reference's expressions `c ** 0.5`, `1. / c`, `K ** 0.5`
    (manifolds/poincare.py:26, manifolds/hyperboloid.py:36-37): double for a python
    float, the tensor's own dtype for a tensor (layers/hyp_layers.py:35).  Tensor
    curvatures are read back once per (storage, version) and cached."""
    if isinstance(c, HdCurv):
        return c
    if torch.is_tensor(c):
        if c.requires_grad and torch.is_grad_enabled():
            # args.c=None makes the curvatures nn.Parameters (layers/hyp_layers.py:30-32); the kernels take c
            # as a host scalar and none of the backward kernels returns d/dc -- refuse rather than train with
            # a silently frozen curvature
            raise NotImplementedError(
                "trainable curvature (args.c=None) has no d/dc kernels: fix c, or run under torch.no_grad()")
        # keyed by object identity and guarded by a weak reference + version counter: a
        # freed tensor's id / storage address can be reused by a different curvature
        hit = _curv_cache.get(id(c))
        if hit is not None and hit[0]() is c and hit[1] == c._version:
            return hit[2]
        ch = c.detach().reshape(-1)[:1].cpu()
        K = 1.0 / ch
        val = HdCurv(float(ch), float(ch ** 0.5), float(K), float(K ** 0.5))
        if len(_curv_cache) > 256:
            _curv_cache.clear()
        _curv_cache[id(c)] = (weakref.ref(c), c._version, val)
        return val
    c = float(c)
    return HdCurv(c, c ** 0.5, 1.0 / c, (1.0 / c) ** 0.5)


def _sfx(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError(f"hypdiff_amd kernels take float32/float64 rows, got {t.dtype}")


def _dev(*ts):
    """Every operand on ONE GPU, and that GPU the current device: the kernels are enqueued on the
    current device's stream, so a pointer into another GPU's memory would be a cross-device access
    without peer mapping (one process per GPU: models are built on the rank's own device)."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("hypdiff_amd runs on the MI355X only: tensor is on %s (no CPU fallback)" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError("hypdiff_amd: operand on %s but the current device is cuda:%d (build the model and "
                               "its inputs on the rank's own GPU, after torch.cuda.set_device)" % (t.device, cur))


def _no_grad(*ts):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        raise NotImplementedError(
            "this hypdiff_amd op has no backward kernel (autograd covers the denoisers, HypLinear / HypAgg / HypAct, "
            "proj / expmap0 / logmap0, mobius_add / expmap / logmap / sqdist): call it under torch.no_grad()")


# the current HIP stream's handle.  torch.cuda.current_stream().cuda_stream builds a Stream object per call: 2.7 us, paid
# for every kernel launch (0.5 ms of config 3's 180-launch eager training step); the raw query is 0.07 us
_raw_stream_fn = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device_fn = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream():
    if _raw_stream_fn is not None and _cur_device_fn is not None:
        return _raw_stream_fn(_cur_device_fn())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(raw_stream())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _rows(x):
    """View x as contiguous (rows, dim); returns (x2d, original shape)."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]) if x.dim() != 2 else x
    return x2.contiguous(), shape


def map1(name, x, c):
    _dev(x)
    if torch.is_grad_enabled() and x.requires_grad:
        from . import autograd as hag
        _, abi, op = name.split("_", 2)
        if op == "proj_tan0":
            return hag.HbProjTan0.apply(x, curv(c))
        if op in hag.MAP_CODES:
            return hag.RowMap.apply(x, abi, op, curv(c))
    _no_grad(x)
    x2, shape = _rows(x)
    out = torch.empty_like(x2)
    _lib.call(name + _sfx(x2), _p(x2), _p(out), x2.shape[0], x2.shape[1], curv(c), _stream())
    return out.view(shape)


def map2(name, a, b, c):
    """Row-wise binary map; a and b broadcast against each other like torch would.  Differentiable w.r.t. both
    operands (the broadcast is a torch op: a (1, d) bias row gets the sum of its rows' gradients)."""
    _dev(a, b)
    if a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(dt), b.to(dt)
    if a.shape != b.shape:
        a, b = torch.broadcast_tensors(a, b)
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        from . import autograd as hag
        if name in hag.RowMap2.CODES:
            return hag.RowMap2.apply(a, b, name, curv(c))
    _no_grad(a, b)
    a2, shape = _rows(a)
    b2, _ = _rows(b)
    out = torch.empty_like(a2)
    _lib.call(name + _sfx(a2), _p(a2), _p(b2), _p(out), a2.shape[0], a2.shape[1], curv(c), _stream())
    return out.view(shape)


def sqdist(name, a, b, c):
    _dev(a, b)
    if a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = a.to(dt), b.to(dt)
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        from . import autograd as hag
        if a.shape != b.shape:
            a, b = torch.broadcast_tensors(a, b)
        return hag.SqdistFn.apply(a, b, name, curv(c))
    a2, shape = _rows(a)
    b2, _ = _rows(b)
    out = torch.empty(a2.shape[0], dtype=a2.dtype, device=a2.device)
    _lib.call(name + _sfx(a2), _p(a2), _p(b2), _p(out), a2.shape[0], a2.shape[1], curv(c), _stream())
    return out.view(shape[:-1])


def map_at(name, centres, labels_dev, x, c):
    """logmap/expmap with the base point gathered per row: centres[labels[row]]."""
    _dev(centres, labels_dev, x); _no_grad(centres, x)
    x2, shape = _rows(x)
    if labels_dev.numel() != x2.shape[0]:
        raise ValueError("map_at: %d labels for %d rows (one label per row; repeat the labels of a node over its "
                         "extra leading axes)" % (labels_dev.numel(), x2.shape[0]))
    ce = centres.to(x2.dtype).contiguous()
    out = torch.empty_like(x2)
    _lib.call(name + _sfx(x2), _p(ce), _p(labels_dev), ce.shape[0], _p(x2), _p(out), x2.shape[0], x2.shape[1],
              curv(c), _stream())
    return out.view(shape)


def hyp_act(manifold, x, c_in, c_out, act_code):
    _dev(x); _no_grad(x)
    x2, shape = _rows(x)
    out = torch.empty_like(x2)
    _lib.call(f"hd_{manifold}_hyp_act" + _sfx(x2), _p(x2), _p(out), x2.shape[0], x2.shape[1], curv(c_in),
              curv(c_out), act_code, _stream())
    return out.view(shape)


def rownorm(x2):
    out = torch.empty(x2.shape[0], dtype=x2.dtype, device=x2.device)
    _lib.call("hd_rownorm" + _sfx(x2), _p(x2), _p(out), x2.shape[0], x2.shape[1], _stream())
    return out


def _pad_k(t, k_pad):
    """zero-pad the last dim of a 2-D tensor to k_pad columns (16-byte rows for the GEMM)"""
    if t.shape[1] == k_pad and t.is_contiguous() and t.data_ptr() % 16 == 0:
        return t
    if t.shape[1] < k_pad:
        return torch.nn.functional.pad(t, (0, k_pad - t.shape[1]))      # one launch (zeros + slice assignment were two)
    out = torch.zeros(t.shape[0], k_pad, dtype=t.dtype, device=t.device)
    out[:, : t.shape[1]] = t
    return out


# Cached device buffers (padded / transposed / limb-split weights, sparse feature rows) are filled by kernels on the stream
# that was current at the miss.  A hit on ANOTHER stream must not read them before those kernels have finished: every entry
# carries (stream, event recorded behind the fill), and a hit from a different stream waits for the event unless it has
# completed (the common case: one query, no wait).
def _tag():
    if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
        return None
    st = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(st)
    return [st.cuda_stream, ev]


def _ready(tag):
    # steady state: the fill completed long ago and the entry says so (no stream query, no event query: a stack of 24
    # small layers makes 24 hits per call and is bound by the host)
    if tag is None or tag[1] is None:
        return
    if torch.cuda.is_current_stream_capturing():
        return
    if tag[1].query():
        tag[1] = None
        return
    st = torch.cuda.current_stream()
    if st.cuda_stream != tag[0]:
        st.wait_event(tag[1])


_pad_cache = {}
_pad_seen = {}


def pad_k_cached(t, k_pad):
    """_pad_k for operands that persist between calls (weights, input features): the padded copy
    is rebuilt only when the tensor object, its version counter or its storage changes."""
    if t.shape[1] == k_pad and t.is_contiguous() and t.data_ptr() % 16 == 0:
        return t
    hit = _pad_cache.get(id(t))
    if hit is not None and hit[0]() is t and hit[1] == (t._version, t.data_ptr(), k_pad):
        _ready(hit[3])
        return hit[2]
    out = _pad_k(t.detach(), k_pad)
    # An operand enters the cache at its SECOND miss as the same live object: the activations of inner layers come through
    # here too (ops.gemm pads whatever it is given), are new tensors on every call and would only push the weights out
    # (and pay for an event each: config 1's encode, 14-wide rows, is bound by the host)
    seen = _pad_seen.get(id(t))
    if seen is not None and seen() is t:
        if len(_pad_cache) > 64:
            _pad_cache.clear()
        _pad_cache[id(t)] = (weakref.ref(t), (t._version, t.data_ptr(), k_pad), out, _tag() if out.is_cuda else None)
    else:
        if len(_pad_seen) > 256:
            _pad_seen.clear()
        _pad_seen[id(t)] = weakref.ref(t)
    return out


_wt_cache = {}


def transposed_cached(w):
    """w^T as a contiguous tensor, rebuilt when the parameter changes (keyed like pad_k_cached)."""
    hit = _wt_cache.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()):
        _ready(hit[3])
        return hit[2]
    out = w.detach().t().contiguous()
    if len(_wt_cache) > 64:
        _wt_cache.clear()
    _wt_cache[id(w)] = (weakref.ref(w), (w._version, w.data_ptr()), out, _tag() if out.is_cuda else None)
    return out


def smallk_image_cached(w):
    """hd_smallk_pack_w_f32 image of a weight (d_out, d_in <= 32), rebuilt when the parameter changes."""
    key = ("smallk", id(w))
    hit = _wt_cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()):
        _ready(hit[3])
        return hit[2]
    wd = w.detach()
    if wd.stride(1) != 1:
        wd = wd.contiguous()
    img = torch.empty(_lib.load().hd_smallk_image_bytes(wd.shape[0]) // 4, dtype=torch.int32, device=wd.device)
    _lib.call("hd_smallk_pack_w_f32", _p(wd), wd.stride(0), wd.shape[0], wd.shape[1], _p(img), _stream())
    if len(_wt_cache) > 64:
        _wt_cache.clear()
    _wt_cache[key] = (weakref.ref(w), (w._version, w.data_ptr()), img, _tag())
    return img


def match_padded(x2, m):
    """A weight whose input carries zero pad columns (models/encoders._lift pads the features to
    16-byte rows once): pad the weight's K to the same width (cached per parameter version)."""
    if x2.shape[1] != m.shape[1]:
        assert x2.shape[1] == (m.shape[1] + 3) // 4 * 4, (x2.shape, m.shape)
        m = pad_k_cached(m, x2.shape[1])
    return m


def gemm(a, w, bias=None, epi=0, resid=None, out=None):
    """epi(a @ w.T + bias): fp32 on the exact-f32 matrix cores, fp64 on the vector ALU."""
    _dev(a, w, bias)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if a.dtype == torch.float64:
        assert epi == 0
        a, w = a.contiguous(), w.contiguous()
        _lib.call("hd_gemm_f64", _p(a), _p(w), _p(bias), _p(out), M, N, K, K, K, out.stride(0), _stream())
        return out
    if K % 4:
        kp = (K + 3) // 4 * 4
        a, w = pad_k_cached(a, kp), pad_k_cached(w, kp)
    a = a if a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 else a.contiguous()
    w = w if w.stride(1) == 1 and w.stride(0) % 4 == 0 and w.data_ptr() % 16 == 0 else w.contiguous()
    ev = GEMM_EVENTS
    if ev is not None:          # bench.py: live HIP-event timing of the dominant kernel
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    sym = gemm_symbol()
    nws = _lib.load().hd_gemm_splitk_workspace_floats(M, N, a.shape[1]) if sym == "hd_gemm_f32_bf16x3" else 0
    if nws > 0:       # few tiles or a long K on a chip's worth of tiles: the K range is split over workgroups
        ws = torch.empty(nws, dtype=torch.float32, device=a.device)
        _lib.call(sym + "_splitk", _p(a), _p(w), _p(bias), _p(resid), None, _p(out), None, M, N, a.shape[1], a.stride(0),
                  w.stride(0), out.stride(0), epi, _p(ws), _stream())
    else:
        _lib.call(sym, _p(a), _p(w), _p(bias), _p(resid), None, _p(out), None, M, N, a.shape[1], a.stride(0),
                  w.stride(0), out.stride(0), epi, _stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N),
                   "gemm_f32_mfma" if GEMM_MODE == "exact" else "gemm_f32_bf16x3"))
    return out


# ---- limb images: fp32 operands pre-split into limbs (csrc/gemm_limbs.hip) ---------------------
class LimbImage:
    """A (rows, cols) fp32 matrix stored as a limb image (include/hypdiff.h): fmt "bf16x3" (three
    bf16 limbs) or "f16x2" (two fp16 limbs of the row scaled by 2^exp[row]; amax = bit patterns
    of the exact row maxima of |x|, which bound the rows of the next product)."""

    __slots__ = ("buf", "rows", "cols", "fmt", "exp", "amax")

    def __init__(self, buf, rows, cols, fmt="bf16x3", exp=None, amax=None):
        self.buf, self.rows, self.cols, self.fmt, self.exp, self.amax = buf, rows, cols, fmt, exp, amax


def limb_empty(rows, cols, device, fmt="bf16x3", zero_amax=True, scales=None):
    """scales: optional (2, rows) int32 tensor for the row exponents / row maxima of an f16x2
    image (the maxima zeroed by the caller when a GEMM epilogue is going to collect them)."""
    q = "hd_limb_image_bytes" if fmt == "bf16x3" else "hd_limb_image_bytes_f16x2"
    nbytes = getattr(_lib.load(), q)(rows, cols)
    buf = torch.empty(max(nbytes // 4, 4), dtype=torch.int32, device=device)
    if fmt == "bf16x3":
        return LimbImage(buf, rows, cols)
    if scales is None:
        scales = (torch.zeros if zero_amax else torch.empty)(2, max(rows, 1), dtype=torch.int32, device=device)
    return LimbImage(buf, rows, cols, fmt, scales[0], scales[1])


def limb_pack(x, fmt="bf16x3"):
    """fp32 (rows, cols) row-major (any row stride) -> LimbImage."""
    _dev(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    img = limb_empty(x.shape[0], x.shape[1], x.device, fmt, zero_amax=False)
    if fmt == "bf16x3":
        _lib.call("hd_limb_pack_f32", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(img.buf), _stream())
    else:
        _lib.call("hd_limb_pack_f16x2_f32", _p(x), x.stride(0), x.shape[0], x.shape[1], _p(img.buf), _p(img.exp),
                  _p(img.amax), _stream())
    return img


def limb_pack_multi(items):
    """[(fp32 matrix, transposed), ...] -> [LimbImage, ...] (bf16x3) in one launch per 16 matrices: the image of the
    matrix itself or of its transpose (hd_limb_pack_multi_f32)."""
    imgs, jobs = [], []
    for x, tr in items:
        _dev(x)
        assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
        rows, cols = (x.shape[1], x.shape[0]) if tr else (x.shape[0], x.shape[1])
        img = limb_empty(rows, cols, x.device, "bf16x3")
        imgs.append(img)
        jobs.append(_lib.HdPackJob(x.data_ptr(), img.buf.data_ptr(), x.stride(0), x.shape[0], x.shape[1], 1 if tr else 0))
    for i in range(0, len(jobs), 16):
        part = jobs[i:i + 16]
        _lib.call("hd_limb_pack_multi_f32", (_lib.HdPackJob * len(part))(*part), len(part), _stream())
    return imgs


def limb_unpack(img):
    out = torch.empty(img.rows, img.cols, dtype=torch.float32, device=img.buf.device)
    if img.fmt == "bf16x3":
        _lib.call("hd_limb_unpack_f32", _p(img.buf), img.rows, img.cols, _p(out), out.stride(0), _stream())
    else:
        _lib.call("hd_limb_unpack_f16x2_f32", _p(img.buf), _p(img.exp), img.rows, img.cols, _p(out), out.stride(0),
                  _stream())
    return out


_wlimb_cache = {}


def weight_limbs(w, fmt="bf16x3", need_norm=False):
    """(LimbImage, max row 2-norm or None) of a weight matrix, rebuilt when the parameter changes
    (keyed like `curv`).  The norm (f16x2 chains only: one read-back per parameter version) feeds
    the row bound of the products written as f16x2 images."""
    key = (id(w), fmt)
    hit = _wlimb_cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr():
        img, wn = hit[3]
        _ready(hit[4])
        if need_norm and wn is None:
            wn = float(w.detach().double().norm(dim=1).max())
            _wlimb_cache[key] = hit[:3] + ((img, wn), hit[4])
        return img, wn
    wd = w.detach()
    img = limb_pack(wd if wd.stride(1) == 1 else wd.contiguous(), fmt)
    wn = float(wd.double().norm(dim=1).max()) if need_norm else None
    if len(_wlimb_cache) > 256:
        _wlimb_cache.clear()
    _wlimb_cache[key] = (weakref.ref(w), w._version, w.data_ptr(), (img, wn), _tag())
    return img, wn


_bmax_cache = {}


def _bias_absmax(b):
    if b is None:
        return 0.0
    hit = _bmax_cache.get(id(b))
    if hit is not None and hit[0]() is b and hit[1] == b._version and hit[2] == b.data_ptr():
        return hit[3]
    v = float(b.detach().abs().max())
    if len(_bmax_cache) > 256:
        _bmax_cache.clear()
    _bmax_cache[id(b)] = (weakref.ref(b), b._version, b.data_ptr(), v)
    return v


def gemm_limbs(a, w, bias=None, epi=0, resid=None, want_f32=True, want_limbs=False, out=None, w_norm=None,
               bias_max=None, out_scales=None):
    """epi(a @ w.T + bias) from limb images; returns (fp32 tensor or None, LimbImage or None).
    f16x2 with want_limbs: w_norm = max row 2-norm of w, bias_max = max|bias| (bounds of the output rows)."""
    M, K, N = a.rows, a.cols, w.rows
    assert w.cols == K and a.fmt == w.fmt, ((a.rows, a.cols, a.fmt), (w.rows, w.cols, w.fmt))
    dev = a.buf.device
    fmt = a.fmt
    c = None
    if want_f32:
        c = out if out is not None else torch.empty(M, N, dtype=torch.float32, device=dev)
    ldc = c.stride(0) if c is not None else (resid.stride(0) if resid is not None else N)
    if resid is not None and c is not None:
        assert resid.stride(0) == c.stride(0)
    if (fmt == "bf16x3" and GEMM_VIA_CHAIN and not _NO_MLP_CHAIN and resid is None and epi in (0, 1)
            and M >= MLP_CHAIN_MIN_ROWS and N % 16 == 0):
        # many rows: the one-launch chain's 128 x 128 tiles, as a chain of one layer (same bits as hd_gemm_limbs_f32)
        o = _mlp_chain_launch(a, [{"w": w, "bias": bias, "epi": epi, "img": want_limbs, "f32": c if want_f32 else False}],
                              label="gemm_limbs")[0]
        return c, o[0]
    cl = limb_empty(M, N, dev, fmt, scales=out_scales) if want_limbs else None
    ev = GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if fmt == "bf16x3":
        _lib.call("hd_gemm_limbs_f32", _p(a.buf), _p(w.buf), _p(bias), _p(resid), _p(c), _p(cl.buf) if cl else None,
                  M, N, K, ldc, epi, _stream())
    else:
        assert resid is None and epi in (0, 1), "f16x2 limb GEMM: epilogues 0 / 1 only"
        bmul = badd = 0.0
        if cl is not None:
            if w_norm is None:
                w_norm = float(limb_unpack(w).double().norm(dim=1).max())
            if bias_max is None:
                bias_max = _bias_absmax(bias)
            kp = (K + 31) // 32 * 32
            bmul, badd = kp ** 0.5 * w_norm * (1.0 + 2.0 ** -10), bias_max * (1.0 + 2.0 ** -10)
        _lib.call("hd_gemm_f16x2_f32", _p(a.buf), _p(a.exp), _p(a.amax), _p(w.buf), _p(w.exp), _p(bias), _p(c),
                  _p(cl.buf) if cl else None, _p(cl.exp) if cl else None, _p(cl.amax) if cl else None,
                  M, N, K, ldc, epi, bmul, badd, _stream())
    if ev is not None:
        e1.record()
        kp, npad = (K + 31) // 32 * 32, (N + 31) // 32 * 32          # limb images: 6 / 4 bytes per element
        eb = 6.0 if fmt == "bf16x3" else 4.0
        ev.append((e0, e1, 2.0 * M * N * K, eb * (M * kp + N * kp) + (4.0 * M * N if c is not None else 0.0)
                   + (eb * M * npad if cl is not None else 0.0), "gemm_limbs" if fmt == "bf16x3" else "gemm_f16x2"))
    return c, cl


ROW_PACK_MAX_DIM = {"f16x2": 1536, "bf16x3": 1024}      # LDS stage of the one-pass row producers


def row_pack(x2, mode, c=None, fmt="f16x2"):
    """fp32 rows -> LimbImage in one pass (hd_row_pack_f16x2_f32 / hd_row_pack_bf16x3_f32): mode 1 =
    Hyperboloid logmap0(x, c), mode 2 = x itself (returns (image, row norms))."""
    _dev(x2)
    assert x2.dim() == 2 and x2.dtype == torch.float32 and x2.is_contiguous() and x2.shape[1] <= ROW_PACK_MAX_DIM[fmt]
    n, d = x2.shape
    img = limb_empty(n, d, x2.device, fmt, zero_amax=False)
    norms = torch.empty(n, dtype=torch.float32, device=x2.device) if mode == 2 else None
    if fmt == "f16x2":
        _lib.call("hd_row_pack_f16x2_f32", _p(x2), mode, curv(c if c is not None else 1.0), _p(img.buf), _p(img.exp),
                  _p(img.amax), _p(norms), n, d, _stream())
    else:
        _lib.call("hd_row_pack_bf16x3_f32", _p(x2), mode, curv(c if c is not None else 1.0), _p(img.buf), _p(norms),
                  n, d, _stream())
    return (img, norms) if mode == 2 else img


def limb_rows_fmt(x2, m):
    """The limb format HypLinear's x @ W^T runs in (inference, fp32, rows short enough for the one-pass row
    producers), or None for the plain GEMM of the fp32 rows."""
    fmt = GEMM_MODE if GEMM_MODE in ROW_PACK_MAX_DIM else None
    if (fmt and x2.dtype == torch.float32 and m.dtype == torch.float32 and x2.shape[0] > 0
            and 128 <= x2.shape[1] <= ROW_PACK_MAX_DIM[fmt]):      # narrower rows waste the 64-lane producer
        return fmt
    return None


def linear_chain(x, layers):
    """Inference-only stack of nn.Linear layers: layers = [(weight, bias, epi), ...] with epi 0
    none, 1 erf-GELU, 2 x + GELU(layer(x)) (ddpm.py:20-22); x fp32 (M, K) on the device.

    GEMM_MODE "f16x2" / "bf16x3": activations travel between layers as limb images written by the
    previous GEMM's epilogue (split once per element), weights are split once per parameter
    version; fp32 copies are written only where a residual or the caller needs them.  Chains with
    a residual epilogue run in the bf16x3 format under either mode.
    GEMM_MODE "exact": one hd_gemm_f32_ex per layer."""
    if not isinstance(x, LimbImage):
        _dev(x)
    if GEMM_MODE == "exact":
        h = x
        for w, b, epi in layers:
            h = gemm(h, w.detach(), None if b is None else b.detach(), epi=epi, resid=h if epi == 2 else None)
        return h
    gemm_symbol()
    if isinstance(x, LimbImage):                      # operand packed by the caller (mlp input)
        fmt, a, xf = x.fmt, x, None
        assert all(epi in (0, 1) for _, _, epi in layers)
    else:
        fmt = "f16x2" if GEMM_MODE == "f16x2" and all(epi in (0, 1) for _, _, epi in layers) else "bf16x3"
        a, xf = limb_pack(x, fmt), x
    # f16x2: the row exponents / maxima of every intermediate image in one zero-filled allocation
    if fmt == "bf16x3" and mlp_chain_ok(a, layers):
        return mlp_chain(a, layers)
    sc = torch.zeros(len(layers) - 1, 2, max(a.rows, 1), dtype=torch.int32, device=a.buf.device) \
        if fmt == "f16x2" and len(layers) > 1 else None
    for i, (w, b, epi) in enumerate(layers):
        last = i == len(layers) - 1
        need_f32 = last or layers[i + 1][2] == 2
        resid = None
        if epi == 2:
            resid = xf if xf.is_contiguous() else xf.contiguous()
        wl, wn = weight_limbs(w, fmt, need_norm=fmt == "f16x2" and not last)
        bd = None if b is None else b.detach()
        xf, a = gemm_limbs(a, wl, bd, epi=epi, resid=resid, want_f32=need_f32, want_limbs=not last, w_norm=wn,
                           bias_max=_bias_absmax(b) if fmt == "f16x2" and not last else None,
                           out_scales=sc[i] if sc is not None and not last else None)
    return xf


# ---- the whole stack as ONE persistent launch over many rows (csrc/gemm_chain.hip) ----
MLP_CHAIN_MIN_ROWS = 4096                                     # below this the per-layer launches are not round-bound
_NO_MLP_CHAIN = bool(os.environ.get("HD_NO_MLP_CHAIN"))       # A/B knob: one hd_gemm_limbs_f32 launch per layer
GEMM_VIA_CHAIN = os.environ.get("HD_GEMM_VIA_CHAIN", "0") != "0"   # a single limb GEMM over many rows as a chain of one layer
# (same bits; measured neutral in the bench step: 0.6414-0.6421 against 0.6419-0.6425 ms on one stream, so the plain launch stays)
_mlp_chain_ws = {}


def mlp_chain_ok(a, layers):
    """hd_mlp_chain_limbs_f32 applies: bf16x3 image input, 2..8 layers with epilogues none / GELU only, enough rows."""
    return (not _NO_MLP_CHAIN and isinstance(a, LimbImage) and a.fmt == "bf16x3" and a.rows >= MLP_CHAIN_MIN_ROWS
            and 2 <= len(layers) <= 8 and all(epi in (0, 1) for _, _, epi in layers)
            and all(w.dtype == torch.float32 for w, _, _ in layers))


def _mlp_chain_launch(a, specs, label="mlp_chain"):
    """One hd_mlp_chain_limbs_f32 launch.  specs: one dict per layer with w (LimbImage of the weight), bias, epi and the
    outputs wanted: img (limb image), f32 (fp32 tensor, or a tensor to write into), z (fp32 pre-activation), aux (epi 3:
    z of the layer below).  fp32 rows are padded to 64 bytes as gemm_limbs_train's are.  Returns [(img, f32, z), ...]."""
    dev = a.buf.device
    M = a.rows
    arr = (_lib.HdMlpChainLayer * len(specs))()
    keep, outs, k_in, flops, nbytes = [], [], a.cols, 0.0, 0.0
    for i, sp in enumerate(specs):
        wl, bd, epi = sp["w"], sp.get("bias"), sp["epi"]
        assert wl.cols == k_in, (wl.cols, k_in)
        n = wl.rows
        ld = (n + 15) // 16 * 16
        img = limb_empty(M, n, dev, "bf16x3") if sp.get("img") else None
        c = sp.get("f32")
        if c is False:
            c = None
        elif c is True:
            c = torch.empty(M, ld, dtype=torch.float32, device=dev)[:, :n]
        z = torch.empty(M, ld, dtype=torch.float32, device=dev)[:, :n] if sp.get("z") else None
        aux = sp.get("aux")
        pitches = {t.stride(0) for t in (c, z, aux) if t is not None}
        assert len(pitches) <= 1, pitches
        ldc = pitches.pop() if pitches else n
        keep += [wl, bd, img, c, z, aux]
        outs.append((img, c, z))
        arr[i] = _lib.HdMlpChainLayer(_p(wl.buf), _p(bd), _p(img.buf) if img is not None else None, _p(c), n, k_in, ldc,
                                      epi, _p(z), _p(aux))
        kp, npad = (k_in + 31) // 32 * 32, (n + 31) // 32 * 32
        flops += 2.0 * M * n * k_in
        nbytes += (6.0 * (M * kp + n * kp) + 4.0 * M * n * ((c is not None) + (z is not None) + (aux is not None))
                   + (6.0 * M * npad if img is not None else 0.0))
        k_in = n
    key = (dev.index, raw_stream(), (M + 127) // 128, len(specs))
    ws = _mlp_chain_ws.get(key)
    if ws is None:
        # zero-filled once: the kernel leaves it clean for the next launch on this stream
        if len(_mlp_chain_ws) > 64:
            _mlp_chain_ws.clear()
        ws = _mlp_chain_ws[key] = torch.zeros(_lib.load().hd_mlp_chain_workspace_bytes(M, len(specs)) // 4 + 4,
                                              dtype=torch.int32, device=dev)
    ev = GEMM_EVENTS
    if ev is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call("hd_mlp_chain_limbs_f32", _p(a.buf), M, arr, len(specs), _p(ws), ws.numel() * 4, _stream())
    if ev is not None:
        e1.record()
        ev.append((e0, e1, flops, nbytes, label))
    return outs


def mlp_chain(a, layers, out=None):
    """layers = [(weight, bias, epi), ...] over the limb image `a`: every layer's tiles in one launch, a row block
    flows from layer to layer as soon as its own tiles are complete.  Bit-identical to the per-layer launches."""
    specs = []
    for i, (w, b, epi) in enumerate(layers):
        last = i == len(layers) - 1
        wl, _ = weight_limbs(w, "bf16x3")
        c = None
        if last:
            c = out if out is not None else torch.empty(a.rows, wl.rows, dtype=torch.float32, device=a.buf.device)
        specs.append({"w": wl, "bias": None if b is None else b.detach(), "epi": epi, "img": not last, "f32": c})
    return _mlp_chain_launch(a, specs)[-1][1]


def mlp_chain_train_ok(a, n_layers):
    """the training step's forward / input-gradient stacks can run as one-launch chains"""
    return (not _NO_MLP_CHAIN and isinstance(a, LimbImage) and a.fmt == "bf16x3" and a.rows >= MLP_CHAIN_MIN_ROWS
            and 2 <= n_layers <= 8)


def mlp_chain_failed():
    """True if a device-side wait of any hd_mlp_chain_limbs_f32 launch timed out since the last call (its output is
    NaN); synchronises the streams that own a workspace.  Tests and debugging."""
    bad = False
    for (_, stream, _, _), ws in _mlp_chain_ws.items():
        bad |= _lib.load().hd_mlp_chain_failed(_p(ws), ctypes.c_void_p(stream)) == 1
    return bad


def chain_small_failed():
    """True if a device-side wait of any hd_linear_chain_small_f32 launch timed out since the last call (the outputs
    of that launch are NaN; later launches are clean).  Synchronises the streams that own a workspace: tests and
    post-mortems only -- the hot path never reads the flag, a failed launch is visible as NaN outputs."""
    bad = False
    for (_, stream), ws in _chain_ws.items():
        bad |= _lib.load().hd_chain_failed(_p(ws), ctypes.c_void_p(stream)) == 1
    return bad


CHAIN_SMALL_MAX_ROWS, CHAIN_SMALL_MAX_WIDTH = 512, 512
_NO_CHAIN_SMALL = bool(os.environ.get("HD_NO_CHAIN_SMALL"))     # debugging knob: one GEMM launch per layer
_chain_ws = {}


def chain_image_cached(w):
    """hd_chain_pack_w_f32 image of a weight (d_out, d_in), rebuilt when the parameter changes."""
    key = ("chain", id(w))
    hit = _wt_cache.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()):
        _ready(hit[3])
        return hit[2]
    wd = w.detach()
    if wd.stride(1) != 1:
        wd = wd.contiguous()
    img = torch.empty(_lib.load().hd_chain_image_bytes(wd.shape[0], wd.shape[1]) // 4, dtype=torch.int32, device=wd.device)
    _lib.call("hd_chain_pack_w_f32", _p(wd), wd.stride(0), wd.shape[0], wd.shape[1], _p(img), _stream())
    if len(_wt_cache) > 256:
        _wt_cache.clear()
    _wt_cache[key] = (weakref.ref(w), (w._version, w.data_ptr()), img, _tag())
    return img


def chain_small_ok(x, layers):
    """The one-launch chain (hd_linear_chain_small_f32) applies: small fp32 batch, every layer <= 512 wide, inner
    widths multiples of 32, fp32-faithful limb arithmetic allowed (GEMM_MODE "exact" keeps the per-layer exact GEMMs)."""
    if _NO_CHAIN_SMALL or GEMM_MODE == "exact" or x.dtype != torch.float32 or x.dim() != 2:
        return False
    if not (0 < x.shape[0] <= CHAIN_SMALL_MAX_ROWS) or len(layers) > 31 or layers[0][2] == 2:
        return False
    for i, (w, b, epi) in enumerate(layers):
        n, k = w.shape
        if max(n, k) > CHAIN_SMALL_MAX_WIDTH or w.dtype != torch.float32 or (i + 1 < len(layers) and n % 32):
            return False
    return True


def linear_chain_small(x, layers, emb_after=None, emb=None, emb_idx=None):
    """layers = [(weight, bias, epi), ...] as linear_chain; after layer `emb_after` the rows of `emb` selected by
    `emb_idx` (int64) are added (ddpm.py:39).  One launch of 256 co-resident workgroups; the workspace is kept per device and stream."""
    _dev(x)
    x = x if x.stride(1) == 1 else x.contiguous()
    n = x.shape[0]
    arr = (_lib.HdChainLayer * len(layers))()
    keep = []
    for i, (w, b, epi) in enumerate(layers):
        img = chain_image_cached(w)
        bd = None if b is None else b.detach()
        keep += [img, bd]
        arr[i] = _lib.HdChainLayer(_p(img), _p(bd), w.shape[1], w.shape[0], epi, 1 if emb_after == i else 0)
    key = (x.device.index, raw_stream())
    ws = _chain_ws.get(key)
    if ws is None:
        # zero-filled once: the kernel's tagged words start from "never valid" (csrc/chain_small.hip)
        ws = _chain_ws[key] = torch.zeros(_lib.load().hd_chain_workspace_bytes() // 4, dtype=torch.int32, device=x.device)
    out = torch.empty(n, layers[-1][0].shape[0], dtype=torch.float32, device=x.device)
    tab = None if emb is None else emb.detach()
    _lib.call("hd_linear_chain_small_f32", _p(x), x.stride(0), n, arr, len(layers), _p(tab), _p(emb_idx),
              0 if tab is None else tab.shape[0], _p(out), out.stride(0), _p(ws), ws.numel() * 4, _stream())
    return out


# ---- adjacency -------------------------------------------------------------------
class CSR:
    """Device CSR (int32 rowptr/col, values in the COO's dtype and entry order)."""

    LONG_ROW = 512          # kLongRow of csrc/hd_spmm.cuh

    def __init__(self, rowptr, col, val, shape):
        self.rowptr, self.col, self.val, self.shape = rowptr, col, val, shape
        # rows with more than LONG_ROW entries (power-law hubs) get a workgroup each;
        # found once per adjacency (one small read-back), empty for the graph-set configs
        deg = rowptr[1:] - rowptr[:-1]
        self.long_rows = (deg > self.LONG_ROW).nonzero().flatten().to(torch.int32).contiguous()
        self.n_long = int(self.long_rows.numel())

    def values(self, dtype):
        return self.val if self.val.dtype == dtype else self.val.to(dtype)

    def long_rows_above(self, thr):
        """(rows with more than `thr` entries, their count) -- LONG_ROW's list, or one built once for a lower threshold"""
        if thr == self.LONG_ROW:
            return self.long_rows, self.n_long
        cache = self.__dict__.setdefault("_long_by_thr", {})
        if thr not in cache:
            deg = self.rowptr[1:] - self.rowptr[:-1]
            rows = (deg > thr).nonzero().flatten().to(torch.int32).contiguous()
            cache[thr] = (rows, int(rows.numel()))
        return cache[thr]

    def transposed(self):
        """CSR of adj^T (backward of torch.spmm w.r.t. its dense operand), built once."""
        t = getattr(self, "_t", None)
        if t is None:
            n = self.shape[0]
            deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(n, device=self.col.device), deg)
            t = csr_from_coo(torch.stack([self.col.to(torch.int64), rows]), self.val, (self.shape[1], self.shape[0]))
            t._t = self
            self._t = t
        return t


def csr_from_coo(indices, values, shape):
    """COO (int64 (2,nnz) indices) -> CSR on the GPU (hd_csr_from_coo).  The reference's
    adjacency is row-sorted (utils/data_utils.py:125-133); an unsorted COO is first
    ordered by row with a stable device sort (plumbing), then bucketed by the same kernel."""
    _dev(indices, values)
    n_rows, n_cols = int(shape[0]), int(shape[1])
    nnz = indices.shape[1]
    dev = indices.device
    row = indices[0].contiguous().to(torch.int64)
    colin = indices[1].contiguous().to(torch.int64)
    for attempt in range(2):
        rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("hd_csr_from_coo", _p(row), _p(colin), nnz, n_rows, n_cols, _p(rowptr), _p(col), _p(flags),
                  _stream())
        f = int(flags.item())
        if f & 2:
            raise IndexError("adjacency index out of range for shape %s" % (tuple(shape),))
        if not (f & 1):
            return CSR(rowptr, col[:nnz], values.contiguous(), (n_rows, n_cols))
        order = torch.sort(row, stable=True).indices
        row, colin, values = row[order], colin[order], values[order]
    raise RuntimeError("hd_csr_from_coo: COO still unsorted after a stable sort")


def csr_of(adj):
    """CSR of a torch sparse COO adjacency, built once and kept on the tensor object."""
    if isinstance(adj, CSR):
        return adj
    cached = getattr(adj, "_hd_csr", None)
    if cached is not None:
        return cached
    if not adj.is_sparse:
        raise TypeError("HypAgg expects a torch sparse COO adjacency (utils/data_utils.py:125-133)")
    csr = csr_from_coo(adj._indices(), adj._values(), adj.shape)
    try:
        adj._hd_csr = csr
    except Exception:
        pass
    return csr


def spmm(csr, x):
    _dev(x)
    x2 = x.contiguous()
    out = torch.empty(csr.shape[0], x2.shape[1], dtype=x2.dtype, device=x2.device)
    _lib.call("hd_spmm_csr" + _sfx(x2), _p(csr.rowptr), _p(csr.col), _p(csr.values(x2.dtype)), _p(x2), _p(out),
              csr.shape[0], x2.shape[1], _p(csr.long_rows) if csr.n_long else None, csr.n_long, _stream())
    return out


def spmm_into(csr, x, out, col0, long_thr=None):
    """out[:, col0:col0 + x.shape[1]] = adj @ x for a column slab x (contiguous) of wider rows
    (hd_spmm_csr_strided: same per-column summation order as spmm on whole rows).  long_thr: the long-row threshold
    of the aggregate this slab is a part of (agg_long_threshold of the whole graph); None = CSR.LONG_ROW."""
    _dev(x, out)
    assert x.is_contiguous() and out.is_contiguous() and out.dtype == x.dtype and out.shape[0] == csr.shape[0]
    w = x.shape[1]
    assert 0 <= col0 and col0 + w <= out.shape[1]
    dst = out[:, col0:]
    thr = CSR.LONG_ROW if long_thr is None else long_thr
    long_rows, n_long = csr.long_rows_above(thr)
    _lib.call("hd_spmm_csr_strided" + _sfx(x), _p(csr.rowptr), _p(csr.col), _p(csr.values(x.dtype)), _p(x), w,
              ctypes.c_void_p(dst.data_ptr()), out.shape[1], csr.shape[0], w,
              _p(long_rows) if n_long else None, n_long, thr, _stream())
    return out


# ---- sparse feature rows: the first HypLinear of a graph with identity / one-hot features ----------------------
class SparseRows:
    """The operand rows of a first HypLinear (ball: the lifted features with their row norms; hyperboloid: their
    logmap0) held as a CSR, because almost every entry is zero: utils/data_utils.py:171 gives Cora / Citeseer
    identity features of width N, whose lift is one scaled unit vector per row.  x W^T is then a gather of rows of W^T
    (hd_spmm_csr), not an N x N x d GEMM.  Built once per feature tensor (sparse_rows_cached)."""

    __slots__ = ("csr", "xnorm", "shape", "abi")

    def __init__(self, csr, xnorm, shape, abi):
        self.csr, self.xnorm, self.shape, self.abi = csr, xnorm, shape, abi


SPARSE_ROWS_MIN_WIDTH = 1024         # narrower features: the GEMM is cheap, the dense path stays
SPARSE_ROWS_MAX_PER_ROW = 8          # mean stored entries per row above which the GEMM wins
_sparse_rows_cache = {}


def sparse_rows_cached(x, manifold, c, lift):
    """SparseRows of lift(x) for `manifold`, or None when x is not sparse enough.  Inference only; the answer is cached
    per feature tensor (identity, version, storage) and curvature, so the dense N x K features are read once."""
    if (torch.is_grad_enabled() and x.requires_grad) or x.dim() != 2 or x.dtype != torch.float32 \
            or x.shape[1] < SPARSE_ROWS_MIN_WIDTH or x.shape[0] == 0:
        return None
    cv = curv(c)
    key = (id(x), manifold.abi, cv.c)
    hit = _sparse_rows_cache.get(key)
    if hit is not None and hit[0]() is x and hit[1] == (x._version, x.data_ptr()):
        _ready(hit[3])
        return hit[2]
    res = None
    if int(torch.count_nonzero(x)) <= SPARSE_ROWS_MAX_PER_ROW * x.shape[0]:
        with torch.no_grad():
            h = lift(x)                                              # proj(expmap0(proj_tan0(x))), dense, once
            if manifold.abi == "pb":
                u, xn = h, rownorm(h.contiguous())
            else:
                u, xn = map1("hd_hb_logmap0", h, c), None            # the operand of hyperboloid.py:122
            nz = u != 0
            counts = nz.sum(1)
            rowptr = torch.zeros(u.shape[0] + 1, dtype=torch.int32, device=u.device)
            rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            col = nz.nonzero()[:, 1].to(torch.int32).contiguous()    # row-major: a row's entries in column order
            val = u[nz].contiguous()
            res = SparseRows(CSR(rowptr, col, val, tuple(u.shape)), xn, tuple(u.shape), manifold.abi)
    if len(_sparse_rows_cache) > 16:
        _sparse_rows_cache.clear()
    _sparse_rows_cache[key] = (weakref.ref(x), (x._version, x.data_ptr()), res, _tag() if x.is_cuda else None)
    return res


def sparse_matvec(sr, m):
    """(sparse operand rows) @ m^T: every output row is the sum of its few stored entries times rows of m^T -- exact
    for one-hot rows (one product per output element, nothing to sum)."""
    k = sr.shape[1]
    mt = transposed_cached(m) if m.shape[1] == k else transposed_cached(m)[:k].contiguous()
    return spmm(sr.csr, mt)


def agg_tail(manifold, s, c_in, c_out=None, act_code=None):
    """proj(expmap0(s)) [+ HypAct] on a stored aggregate s = spmm(adj, xt) (hd_*_agg_tail)."""
    _dev(s)
    s2 = s.contiguous()
    out = torch.empty_like(s2)
    fuse = act_code is not None
    _lib.call(f"hd_{manifold}_agg_tail" + _sfx(s2), _p(s2), _p(out), s2.shape[0], s2.shape[1], curv(c_in),
              curv(c_out if fuse else c_in), act_code if fuse else 0, 1 if fuse else 0, _stream())
    return out


AGG_FOUR_ROW_MIN = 8192          # csrc/hd_spmm.cuh: from this many rows the fused tail runs four rows per wavefront
AGG_MID_ROW = 64                 # long-row threshold of the fused aggregate for graphs below AGG_FOUR_ROW_MIN rows


def agg_long_threshold(n_rows):
    """Rows with more entries than this are summed by a workgroup in 8 fixed slices instead of by one wavefront in CSR
    order (rounding-level difference).  A small graph's launch is as long as its longest one-wavefront row (config 4:
    42 us for 2708 rows, a row of several hundred neighbours), so there the threshold is 64; a property of the WHOLE
    graph, like agg_form."""
    return AGG_MID_ROW if n_rows < AGG_FOUR_ROW_MIN else CSR.LONG_ROW


def agg_form(manifold, n_rows, n_long):
    """The form of the fused aggregate tail that hd_*_agg picks by itself for an adjacency of n_rows rows with n_long
    rows above CSR.LONG_ROW entries: 2 = four rows per wavefront, 1 = one row per wavefront (they differ by rounding)."""
    return 2 if n_rows >= AGG_FOUR_ROW_MIN and (manifold != "hb" or n_long == 0) else 1


def agg(manifold, csr, xt, c_in, c_out=None, act_code=None):
    """proj(expmap0(spmm(adj, xt))) [+ HypAct when act_code is given], one kernel.  A CSR that is a row shard of a
    larger graph carries `whole` = (rows, long rows) of that graph and runs the form the whole graph takes."""
    _dev(xt)
    x2 = xt.contiguous()
    if csr.shape[1] != x2.shape[0]:
        raise ValueError(f"adjacency {csr.shape} does not match {tuple(x2.shape)} rows")
    out = torch.empty(csr.shape[0], x2.shape[1], dtype=x2.dtype, device=x2.device)
    fuse = act_code is not None
    whole = getattr(csr, "whole", None)
    form = agg_form(manifold, *whole) if whole is not None else 0
    thr = agg_long_threshold(whole[0] if whole is not None else csr.shape[0])
    long_rows, n_long = csr.long_rows_above(thr)
    _lib.call(f"hd_{manifold}_agg" + _sfx(x2), _p(csr.rowptr), _p(csr.col), _p(csr.values(x2.dtype)), _p(x2),
              _p(out), csr.shape[0], x2.shape[1], curv(c_in), curv(c_out if fuse else c_in),
              act_code if fuse else 0, (1 if fuse else 0) | (form << 8) | (thr << 10), _p(long_rows) if n_long else None,
              n_long, _stream())
    return out


# ---- labels / sign tables ----------------------------------------------------------
_label_cache = {}


def labels_dev(labels, device):
    """int32 device copy of the host label array (the reference indexes python lists
    with it row by row, models/direction_diffusion.py:159); cached by content."""
    if torch.is_tensor(labels):
        return labels.to(device=device, dtype=torch.int32).contiguous()
    arr = np.ascontiguousarray(np.asarray(labels).reshape(-1))
    key = (id(labels), str(device))
    hit = _label_cache.get(key)
    if hit is not None and hit[0].shape == arr.shape and np.array_equal(hit[0], arr):
        return hit[1]
    t = torch.from_numpy(arr.astype(np.int32)).to(device)
    if len(_label_cache) > 64:
        _label_cache.clear()
    _label_cache[key] = (arr.copy(), t)
    return t
