import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
from hypdiff_amd.manifolds import Hyperboloid
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
man = Hyperboloid()
torch.manual_seed(0)
n = 15777
x = man.proj(man.expmap0(torch.cat([torch.zeros(n, 1), torch.randn(n, 23) * 0.3], 1).cuda(), 1.0), 1.0)
w = torch.nn.Parameter(torch.randn(512, 24, device="cuda") * 0.1)
with torch.no_grad():
    print("R=%s fused HypLinear 24->512: %.1f us" % (os.environ.get("HD_SMALLK_R", "4"), timeit(lambda: man.mobius_matvec(w, x, 1.0, _flags=3))), flush=True)
