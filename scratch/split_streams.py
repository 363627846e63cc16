"""Denoiser chain over all rows on one stream vs row halves / thirds on concurrent streams."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
torch.manual_seed(0)
M, D = 15777, 512
dims = [(513, 800), (800, 900), (900, 800), (800, 700), (700, 512)]
ws = [torch.nn.Parameter(torch.randn(n, k, device="cuda") / k ** 0.5) for k, n in dims]
bs = [torch.nn.Parameter(torch.randn(n, device="cuda") * 0.1) for k, n in dims]
layers = [(w, b, 1 if i < 4 else 0) for i, (w, b) in enumerate(zip(ws, bs))]
x = torch.randn(M, 513, device="cuda")
def chain(xx):
    return ops.linear_chain(xx, layers)
def run(parts, iters=50):
    bounds = [0]
    per = (M // parts + 127) // 128 * 128
    while bounds[-1] < M: bounds.append(min(M, bounds[-1] + per))
    xs = [x[bounds[i]:bounds[i + 1]].contiguous() for i in range(len(bounds) - 1)]
    streams = [torch.cuda.Stream() for _ in xs]
    main = torch.cuda.current_stream()
    def step():
        if len(xs) == 1:
            return [chain(xs[0])]
        outs = []
        for s, xx in zip(streams, xs):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(chain(xx))
        for s in streams: main.wait_stream(s)
        return outs
    with torch.no_grad():
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters): step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
for parts in (1, 2, 3, 4, 1, 2):
    print("row parts", parts, "chain ms %.4f" % run(parts), flush=True)
