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
for (M, K, N) in [(4096, 4096, 4096), (15777, 800, 900)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    al, wl = ops.limb_pack(a), ops.limb_pack(w)
    t = timeit(lambda: ops.gemm_limbs(al, wl, None, epi=0))
    print(os.environ.get("HD_GEMM_LIMBS_DBG"), M, K, N, "%.1f us %.1f TF" % (t * 1e3, 2.0 * M * N * K / t / 1e9), flush=True)
