"""Cost of the limb GEMM's write-out forms in training (z, fp32 copy, limb image): tools/bench_writeout.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops, autograd as hag
def timeit(fn, iters=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M=15777
for N,K in [(900,800),(800,900),(512,700)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    aux = torch.randn(M, (N + 15) // 16 * 16, device="cuda")[:, :N]
    al, wl = ops.limb_pack(a), ops.limb_pack(w)
    r = {}
    r["fwd CL"] = timeit(lambda: hag.gemm_limbs_train(al, wl, b, None, 1, want_z=False, want_f32=False, want_limbs=True))
    r["fwd C"] = timeit(lambda: hag.gemm_limbs_train(al, wl, b, None, 1, want_z=False, want_f32=True, want_limbs=False))
    r["fwd Z+CL"] = timeit(lambda: hag.gemm_limbs_train(al, wl, b, None, 1, want_z=True, want_f32=False, want_limbs=True))
    r["fwd Z+C+CL"] = timeit(lambda: hag.gemm_limbs_train(al, wl, b, None, 1, want_z=True, want_f32=True, want_limbs=True))
    r["bwd C+CL"] = timeit(lambda: hag.gemm_limbs_train(al, wl, None, aux, 3, want_f32=True, want_limbs=True))
    r["bwd CL"] = timeit(lambda: hag.gemm_limbs_train(al, wl, None, aux, 3, want_f32=False, want_limbs=True))
    print(N, K, {k: round(v,1) for k,v in r.items()}, flush=True)
