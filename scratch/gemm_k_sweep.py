import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import autograd as hag
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (15777, 16384, 32768):
  for N in (896, 900, 1024):
    for K in (400, 800, 1600, 3200):
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        r = {}
        for epi in (0, 1):
            us = timeit(lambda: hag.gemm_ex(x, w, b, None, None, out, None, epi))
            r[f"epi{epi}"] = (round(us, 1), round(2.0*M*N*K/us/1e6, 1))
        print(M, N, K, r, flush=True)
