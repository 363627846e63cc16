import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import manifolds as om, layers as ol
import hypdiff_amd.manifolds as hm
from hypdiff_amd import ops
g = np.load('tests/golden/manifold_PoincareBall.npz')
k = 'f32/c0.6/d16/s30/'
p, q = torch.from_numpy(g[k+'p']), torch.from_numpy(g[k+'q'])
ref = g[k+'mobius_add']
ex = om.pb_mobius_add(p.double(), q.double(), 0.6).numpy()
got = hm.PoincareBall().mobius_add(p.cuda(), q.cuda(), 0.6).cpu().numpy()
for i in range(5):
    print(i, ref[i,:3], ex[i,:3], got[i,:3])
g = np.load('tests/golden/layers.npz')
k0='ba100/'; k=k0+'Hyperboloid/bias0/'
h = torch.from_numpy(g[k+'lift'])
W = torch.from_numpy(g[k+'W0'])
man = hm.Hyperboloid()
u_ref = om.hb_logmap0(h, 1.0)
u = man.logmap0(h.cuda(), 1.0).cpu()
print('logmap0 err', (u-u_ref).abs().max().item())
mu_ref = u_ref @ W.T
mu = ops.gemm(u_ref.cuda(), W.cuda()).cpu()
print('gemm err', (mu-mu_ref).abs().max().item(), mu.shape)
bad = ((mu-mu_ref).abs() > 1e-4).nonzero()
print(bad[:10], len(bad))
out = man.mobius_matvec(W.cuda(), h.cuda(), 1.0, _flags=1).cpu()
print('lin err', (out - torch.from_numpy(g[k+'L0/linear'])).abs().max().item())
