import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
shapes = [(15777, 900, 800), (15777, 800, 516), (15777, 800, 900), (15777, 700, 800), (15777, 512, 700), (15777, 512, 512), (4096, 4096, 4096)]
torch.manual_seed(0)
for (M, N, K) in shapes:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    for epi in (1,):
        for _ in range(3): ops.gemm(a, w, b, epi=epi, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, b, epi=epi, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"M={M} N={N} K={K} epi={epi}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF")
