import torch, sys
sys.path.insert(0, "/root/repo")
from hypdiff_amd import ops
ops._ready = lambda tag: None
w = torch.randn(900, 802, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    torch.cuda._sleep(40_000_000)
    img_side, _ = ops.weight_limbs(w)
img_main, _ = ops.weight_limbs(w)
back = ops.limb_unpack(img_main)
torch.cuda.synchronize()
print("without the guard equal =", bool(torch.equal(back, w)))
