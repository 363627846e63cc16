import sys, os, torch, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
shapes = [(15777, 900, 800), (15777, 800, 516), (15777, 800, 900), (15777, 700, 800), (15777, 512, 700), (15777, 512, 512), (15777, 512, 24), (4096, 4096, 4096), (200, 512, 512)]
torch.manual_seed(0)
for (M, N, K) in shapes:
    a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda')
    res = []
    for tune in ("", "w1", "w2", "w3", "n1", "n2", "n3", "n4"):
        if tune: os.environ["HD_GEMM_TUNE"] = tune
        else: os.environ.pop("HD_GEMM_TUNE", None)
        for _ in range(3): ops.gemm(a, w, b, epi=1, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, w, b, epi=1, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append(f"{tune or 'auto'}:{ms*1e3:.0f}us/{2*M*N*K/ms/1e9:.0f}TF")
    print(f"M={M} N={N} K={K}  " + "  ".join(res), flush=True)
