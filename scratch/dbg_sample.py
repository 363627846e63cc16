import sys, os, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from hypdiff_amd.models.direction_diffusion import diffusion
from oracle import diffusion as odiff
torch.manual_seed(0); rs = np.random.RandomState(0)
n, dim, T = 300, 16, 200
x = rs.standard_normal((n, dim)).astype(np.float32); x *= 0.5 / np.linalg.norm(x, axis=1, keepdims=True)
labels = rs.randint(0, 3, size=n); u0 = (rs.standard_normal((3, dim)) * 0.05).astype(np.float32)
d = diffusion(types.SimpleNamespace(r=2.0, t=1.0), dim, num_timesteps=T)
opt = torch.optim.Adam(d.parameters(), lr=1e-3)
xg, ug = torch.from_numpy(x).cuda(), torch.from_numpy(u0).cuda()
for ep in range(200):
    opt.zero_grad(); loss = d(xg, None, labels, ug, restrict=True, graphset=True); loss.backward(); opt.step()
print("loss", float(loss))
d.eval()
img0 = torch.randn(n, dim)
params = [(getattr(d.mlp, f"fc{i}").weight.detach().cpu(), getattr(d.mlp, f"fc{i}").bias.detach().cpu()) for i in range(1, 6)]
with torch.no_grad():
    direction = d.directions(ug)
    img = d.tran_direction(direction, img0.clone().cuda(), labels)
    h = d._logmap_centres(xg.clone(), labels, ug)
    m = d.moments(h); img = img + d._apply_moments(h, m, 0)
    # oracle
    signs = odiff.sign_table(torch.from_numpy(u0))
    oimg = odiff.tran_direction_(img0.clone(), signs, labels)
    oh = odiff.pb_logmap(torch.from_numpy(u0)[torch.from_numpy(labels)], torch.from_numpy(x), 1.0)
    oimg = oimg + odiff.normalization(oh)
    print("start diff", float((img.cpu() - oimg).abs().max()), "absmax", float(oimg.abs().max()))
    tables = odiff.schedule(T)
    for t in reversed(range(T)):
        img, x0 = d.p_sample(img, None, 1, direction, labels, ug, t, None, True)
        oimg, ox0 = odiff.p_sample(oimg, t, signs, labels, tables, params)
        if t % 10 == 9 or t < 3 or not bool(torch.isfinite(img).all()): print(t, "hip absmax %.3e finite %s | oracle absmax %.3e finite %s | x0 raw diff %.2e" % (
            float(img.abs().max()), bool(torch.isfinite(img).all()), float(oimg.abs().max()), bool(torch.isfinite(oimg).all()),
            float((x0.cpu() - ox0).abs().max())))
