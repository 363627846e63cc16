"""Accuracy (vs fp64) and speed of the two fp32 GEMM kernels on the denoiser shapes."""
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
shapes = [(15777, 516, 800), (15777, 800, 900), (15777, 900, 800), (15777, 800, 700), (15777, 700, 512),
          (15777, 512, 512), (15777, 24, 512), (200, 512, 512), (4096, 4096, 4096), (333, 68, 70)]
for (M, K, N) in shapes:
    a = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, 1, device="cuda") * 2)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    ref = a.double() @ w.double().T + b.double()
    scale = (a.double().abs() @ w.double().abs().T + b.double().abs())
    line = f"M={M} K={K} N={N}:"
    for mode in ("exact", "bf16x3"):
        ops.GEMM_MODE = mode
        out = ops.gemm(a, w, b, epi=0)
        err = ((out.double() - ref).abs() / scale).max().item()
        rel = ((out.double() - ref).norm() / ref.norm()).item()
        t = timeit(lambda: ops.gemm(a, w, b, epi=1))
        line += f"  [{mode}] max|err|/sum|a||w| {err:.2e} fro-rel {rel:.2e} {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.1f} TF"
    print(line, flush=True)
# integer data: both kernels exact
a = torch.randint(-64, 64, (777, 260), device="cuda").float(); w = torch.randint(-64, 64, (130, 260), device="cuda").float()
ops.GEMM_MODE = "bf16x3"
print("integer exact:", torch.equal(ops.gemm(a, w), (a.double() @ w.double().T).float()))
a = torch.randint(-(1 << 20), 1 << 20, (300, 64), device="cuda").float(); w = torch.randint(-8, 8, (96, 64), device="cuda").float()
print("integer 2^20 exact:", torch.equal(ops.gemm(a, w), (a.double() @ w.double().T).float()))
