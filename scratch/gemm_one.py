import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "exact"
torch.manual_seed(0)
a = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda')
if mode == "limbs":
    al, wl = ops.limb_pack(a), ops.limb_pack(w)
    for _ in range(40): ops.gemm_limbs(al, wl, b, epi=1, out=out)
else:
    ops.GEMM_MODE = mode
    for _ in range(40): ops.gemm(a, w, b, epi=1, out=out)
torch.cuda.synchronize()
