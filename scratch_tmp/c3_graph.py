import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hypdiff_amd.ddpm import Model
from hypdiff_amd.dist import DataParallelTrainer
dev = "cuda"
D, Hd, L, T = 64, 512, 10, 1000
for B in (200, 4096):
    torch.manual_seed(0)
    model = Model(D, Hd, L, T, 1e-4, 0.02).to(dev).train()
    tr = DataParallelTrainer(model, lr=1e-3)
    x = torch.randn(B, D, device=dev)
    replay = tr.capture(lambda: model.loss_fn(x))
    for _ in range(5): replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): loss = replay()
    torch.cuda.synchronize()
    print("graph", B, round((time.perf_counter() - t0) / 50 * 1e3, 3), "ms", float(loss), flush=True)
