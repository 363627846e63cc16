import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd import ops
M = 15777
dims = [516, 800, 900, 800, 700, 512]
torch.manual_seed(0)
Ws = [torch.randn(dims[i+1], dims[i], device="cuda") / dims[i]**0.5 for i in range(5)]
bs = [torch.randn(dims[i+1], device="cuda") for i in range(5)]
x = torch.randn(M, 516, device="cuda")
bufs = [torch.empty(M, dims[i+1], device="cuda") for i in range(5)]
def chain(lo, hi):
    h = x[lo:hi]
    for i in range(5):
        h = ops.gemm(h, Ws[i], bs[i], epi=1 if i < 4 else 0, out=bufs[i][lo:hi])
    return h
def single():
    chain(0, M)
def multi(ns, streams):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    step = (M + ns - 1) // ns
    step = (step + 127) // 128 * 128
    for s in range(ns):
        lo, hi = s * step, min(M, (s + 1) * step)
        if lo >= hi: break
        streams[s].wait_event(ev)
        with torch.cuda.stream(streams[s]):
            chain(lo, hi)
        e2 = torch.cuda.Event(); e2.record(streams[s]); main.wait_event(e2)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
fl = sum(2.0 * M * dims[i] * dims[i+1] for i in range(5))
t = timeit(single); print("single stream: %.1f us  %.1f TF" % (t*1e3, fl / t / 1e9))
ref = bufs[4].clone()
for ns in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    t = timeit(lambda: multi(ns, streams)); print("%d streams: %.1f us  %.1f TF  equal=%s" % (ns, t*1e3, fl / t / 1e9, torch.equal(ref, bufs[4])))
