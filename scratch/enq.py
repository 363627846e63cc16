import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_device(0)
w = bench.build_workload(512, 1000, 1432, "cuda:0")
with torch.no_grad():
    for i in range(20): bench.one_step(w, 999 - i)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(200): bench.one_step(w, 999 - i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("enqueue %.4f ms/step, total %.4f ms/step" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3), flush=True)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(100): bench.one_step(w, 999 - i)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
