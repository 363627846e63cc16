"""Accuracy (vs fp64) and speed of the f16x2 limb GEMM against the bf16x3 one on the denoiser shapes."""
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
x = torch.randn(300, 70, device="cuda") * torch.exp(torch.randn(300, 1, device="cuda") * 8)
x[5] = 0
x[7, 3] = 1e-30
rt = ops.limb_unpack(ops.limb_pack(x, "f16x2"))
rowmax = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-37)
print("f16x2 pack/unpack: max |err| / rowmax %.3e  max rel err %.3e" % (((rt - x).abs() / rowmax).max().item(),
      ((rt - x).abs() / x.abs().clamp_min(1e-37)).max().item()), flush=True)
# subnormal limbs: elements 2^-30 of the row maximum land on fp16's subnormal grid (2^-24 after
# scaling: ~8 bits); a matrix core that flushed them would lose the whole contribution
a = torch.full((128, 64), 2.0 ** -30, device="cuda"); a[:, 0] = 1.0
a = a * (1 + torch.rand_like(a))
w = torch.zeros(64, 64, device="cuda"); w[:, 1:] = torch.randn(64, 63, device="cuda")
ref = a.double() @ w.double().T
got, _ = ops.gemm_limbs(ops.limb_pack(a, "f16x2"), ops.limb_pack(w, "f16x2"))
print("subnormal-limb product: max err / sum|a||w| %.3e (flushed subnormals would give ~1)" %
      ((got.double() - ref).abs() / (a.double().abs() @ w.double().abs().T)).max().item(), flush=True)
shapes = [(333, 68, 70), (128, 32, 64), (15777, 516, 800), (15777, 800, 900), (15777, 900, 800), (15777, 800, 700),
          (15777, 700, 512), (15777, 512, 512), (4096, 4096, 4096)]
for (M, K, N) in shapes:
    a = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, 1, device="cuda") * 2)
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    ref = a.double() @ w.double().T + b.double()
    scale = (a.double().abs() @ w.double().abs().T + b.double().abs())
    res = {}
    for fmt in ("bf16x3", "f16x2"):
        al, wl = ops.limb_pack(a, fmt), ops.limb_pack(w, fmt)
        wn = float(w.double().norm(dim=1).max())
        out, outl = ops.gemm_limbs(al, wl, b, epi=0, want_f32=True, want_limbs=True, w_norm=wn)
        torch.cuda.synchronize()
        err = ((out.double() - ref).abs() / scale).max().item()
        out2 = ops.limb_unpack(outl)
        orow = out.abs().amax(dim=1, keepdim=True).clamp_min(1e-37)
        err2 = ((out2 - out).abs() / orow).max().item()
        amax_ok = True
        if fmt == "f16x2":
            amax_ok = bool(torch.equal(outl.amax.view(torch.float32), out.abs().amax(dim=1)))
            slack = (15 - torch.log2(out.abs().amax(dim=1).clamp_min(1e-37)) - outl.exp.float())
            res["slack"] = (slack.min().item(), slack.max().item())
        t = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=True, want_limbs=False))
        t2 = timeit(lambda: ops.gemm_limbs(al, wl, b, epi=1, want_f32=False, want_limbs=True, w_norm=wn))
        res[fmt] = (err, err2, t, t2, amax_ok)
    e3, e2 = res["bf16x3"], res["f16x2"]
    fl = 2.0 * M * N * K / 1e9
    print(f"M={M} K={K} N={N}: err/sum|a||w| bf16x3 {e3[0]:.2e} f16x2 {e2[0]:.2e} | limb-out vs f32-out / rowmax {e3[1]:.1e} {e2[1]:.1e}"
          f" amax exact {e2[4]} exp slack bits {res['slack'][0]:.1f}..{res['slack'][1]:.1f}"
          f" | f32 out {e3[2]*1e3:.1f} -> {e2[2]*1e3:.1f} us ({fl/e2[2]:.0f} TF) | limb out {e3[3]*1e3:.1f} -> {e2[3]*1e3:.1f} us ({fl/e2[3]:.0f} TF)", flush=True)
