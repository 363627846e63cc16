"""Accuracy (vs fp64) and speed of the limb-image GEMM on the denoiser shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

torch.manual_seed(0)
x = torch.randn(300, 70, device="cuda")
rt = ops.limb_unpack(ops.limb_pack(x))
print("pack/unpack round trip max rel err", ((rt - x).abs() / x.abs().clamp_min(1e-30)).max().item(), flush=True)
shapes = [(333, 68, 70), (128, 32, 64), (15777, 516, 800), (15777, 800, 900), (15777, 900, 800), (15777, 800, 700), (15777, 700, 512),
          (15777, 512, 512), (15777, 24, 512), (200, 512, 512), (4096, 4096, 4096)]
for (M, K, N) in shapes:
    a = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, 1, device="cuda") * 2)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    ref = a.double() @ w.double().T + b.double()
    scale = (a.double().abs() @ w.double().abs().T + b.double().abs())
    al, wl = ops.limb_pack(a), ops.limb_pack(w)
    out, outl = ops.gemm_limbs(al, wl, b, epi=0, want_f32=True, want_limbs=True)
    torch.cuda.synchronize()
    err = ((out.double() - ref).abs() / scale).max().item()
    out2 = ops.limb_unpack(outl)
    err2 = ((out2.double() - out.double()).abs() / out.double().abs().clamp_min(1e-30)).max().item()
    ops.GEMM_MODE = "exact"
    ex = ops.gemm(a, w, b, epi=1)
    g, _ = ops.gemm_limbs(al, wl, b, epi=1)
    errg = ((g - ex).abs().max() / ex.abs().max()).item()
    t = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=True, want_limbs=False))
    t2 = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=False, want_limbs=True))
    tp = timeit(lambda: ops.limb_pack(a))
    print(f"M={M} K={K} N={N}: max|err|/sum|a||w| {err:.2e} limb-out vs f32-out {err2:.2e} gelu vs exact {errg:.2e} | f32 out {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.1f} TF | limb out {t2*1e3:.1f} us {2.0*M*N*K/t2/1e9:.1f} TF | pack A {tp*1e3:.1f} us", flush=True)
