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
line = os.environ.get("HD_GEMM_LIMBS_TUNE", "-") + ":"
tot = 0.0
for (M, K, N) in [(15777, 516, 800), (15777, 800, 900), (15777, 900, 800), (15777, 800, 700), (15777, 700, 512), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    al, wl = ops.limb_pack(a), ops.limb_pack(w)
    ref = a.double() @ w.double().T + b.double()
    out, _ = ops.gemm_limbs(al, wl, b, epi=0)
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    t = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=False, want_limbs=True))
    if M != 4096: tot += t
    line += " %.0f us %.0f TF (%.0e)|" % (t * 1e3, 2.0 * M * N * K / t / 1e9, err)
print(line, "mlp total %.0f us" % (tot * 1e3), flush=True)
