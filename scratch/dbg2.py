import numpy as np, torch, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import manifolds as om
import hypdiff_amd.manifolds as hm
g = np.load('tests/golden/manifold_Hyperboloid.npz')
k = 'f32/c0.6/d16/s30/'
p, q = torch.from_numpy(g[k+'p']), torch.from_numpy(g[k+'q'])
ref = g[k+'mobius_add']
man = hm.Hyperboloid()
got = man.mobius_add(p.cuda(), q.cuda(), 0.6).cpu().numpy()
orc = om.hb_mobius_add(p, q, 0.6).numpy()
print(ref[:, :3]); print(got[:, :3]); print(orc[:, :3])
u_ref = om.hb_logmap0(q, 0.6); u = man.logmap0(q.cuda(), 0.6).cpu()
print('logmap0', (u-u_ref).abs().max().item(), u_ref.abs().max().item())
v_ref = om.hb_ptransp0(p, u_ref, 0.6); v = man.ptransp0(p.cuda(), u_ref.cuda(), 0.6).cpu()
print('ptransp0', (v-v_ref).abs().max().item(), v_ref.abs().max().item())
e_ref = om.hb_expmap(v_ref, p, 0.6); e = man.expmap(v_ref.cuda(), p.cuda(), 0.6).cpu()
print('expmap', (e-e_ref).abs().max().item(), e_ref.abs().max().item())
