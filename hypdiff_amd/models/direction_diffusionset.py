"""Graph-set direction diffusion on HIP kernels -- drop-in for the reference's
models/direction_diffusionset.py (`diffusionset`, `mlp`): tensors are (G, N, D), one
timestep per graph, per-graph centres u0 (G, K, D) and labels (G, N).

The reference walks G*N rows in Python double loops (:165-171, :317-320, :348-351,
:369-372).  Here the graph axis is folded into the row axis -- rows (G*N, D), t repeated
per node, labels offset by graph into one (G*K, D) centre / sign table -- and the same
row kernels as models/direction_diffusion.py run once over the whole set.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as hag
from .. import ops
from . import direction_diffusion as dd


class mlp(dd.mlp):
    """direction_diffusionset.py:47-72: t is broadcast over the nodes of its graph"""

    def __init__(self, input_dim, output_dim):
        nn.Module.__init__(self)
        self.fc1 = nn.Linear(input_dim + 1, 800)
        self.fc2 = nn.Linear(800, 900)
        self.fc3 = nn.Linear(900, 800)
        self.fc4 = nn.Linear(800, 700)
        self.fc5 = nn.Linear(700, output_dim)
        self._w1_cache = None

    def forward(self, x, t):
        if x.dim() == 2:            # already folded to (G*N, D) rows
            return dd.mlp.forward(self, x, t)
        g, n, d = x.shape
        if torch.is_tensor(t):
            t = t.repeat_interleave(n)
        out = dd.mlp.forward(self, x.reshape(g * n, d), t)
        return out.view(g, n, -1)


class diffusionset(dd.diffusion):
    def __init__(self, args, in_dim, out_dim, num_timesteps=200, prior_horizon=500, c=1.0, device=None):
        dd.diffusion.__init__(self, args, in_dim, num_timesteps, prior_horizon, c, device)
        self.mlp = mlp(in_dim, out_dim).to(self.device)

    # ---- folding helpers ---------------------------------------------------------------
    @staticmethod
    def _centres(u0):
        u = u0 if torch.is_tensor(u0) else torch.stack([x if torch.is_tensor(x) else torch.stack(list(x)) for x in u0])
        return u.reshape(-1, u.shape[-1]).contiguous(), u.shape[-2]

    @staticmethod
    def _flat_labels(labels, k):
        """(G, N) labels -> (G*N,) indices into the folded (G*K, D) centre / sign table.  Device
        tensors are folded on the device (no read-back, no re-upload); a 1-D int32 device tensor
        is taken as already folded (p_sample_loop folds once for its T steps)."""
        if torch.is_tensor(labels):
            if labels.dim() == 1 and labels.dtype == torch.int32 and labels.is_cuda:
                return labels
            if labels.is_cuda:
                off = torch.arange(labels.shape[0], device=labels.device, dtype=torch.int64)[:, None] * k
                return (labels.to(torch.int64) + off).reshape(-1).to(torch.int32)
            labels = labels.numpy()
        lab = np.asarray(labels)
        return (lab.astype(np.int64) + np.arange(lab.shape[0], dtype=np.int64)[:, None] * k).reshape(-1)

    def _signs(self, direction):
        """(G*K, D) fp32 sign table over all graphs' directions, cached per direction list"""
        key = tuple((id(d), d._version) for d in direction)
        if self._sign_cache is None or self._sign_cache[0] != key:
            tab = torch.cat([d.reshape(-1, d.shape[-1]) for d in direction], 0)
            self._sign_cache = (key, torch.sign(tab).to(torch.float32).contiguous(), list(direction), ops._tag())
        else:
            ops._ready(self._sign_cache[3])
        return self._sign_cache[1]

    def directions(self, u0):
        """[logmap0(u) for u in u0] (:307-309, :361-363): list over graphs of (K, D)"""
        cen, k = self._centres(u0)
        return list(self.manifold.logmap0(cen, c=1.0).view(-1, k, cen.shape[-1]).unbind(0))

    def tran_direction(self, direction_vector, gaussian_point, labels):
        """:161-171, in place"""
        g, n, d = gaussian_point.shape
        k = direction_vector[0].shape[0]
        flat = gaussian_point.view(g * n, d)
        dd.diffusion.tran_direction(self, direction_vector, flat, self._flat_labels(labels, k))
        return gaussian_point

    def getpri(self, t):
        return dd.diffusion.getpri(self, t)

    def q_sample(self, x_start, t, direction, noise, labels, u0, restrict=False):
        """:207-222 -> fp64 (G, N, D); noise sign-flipped in place when restrict"""
        g, n, d = x_start.shape
        noise = self.default(noise, lambda: torch.randn_like(x_start))
        k = direction[0].shape[0]
        out = dd.diffusion.q_sample(self, x_start.reshape(g * n, d), t.repeat_interleave(n), direction,
                                    noise.view(g * n, d), self._flat_labels(labels, k) if restrict else None, None,
                                    restrict)
        return out.view(g, n, d)

    def predict_noise_from_start(self, x_t, t, x0):
        """:237-243"""
        prio = self.getpri(t)
        return (self.extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t + prio.unsqueeze(1) * x_t - x0) / \
            self.extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape)

    def p_sample(self, x, h, adj, direction, labels, u0, t: int, x_self_cond=None, graphset=False):
        """:279-300 -> (pred_img_t fp64 (G,N,D), x_start fp32 clamped)"""
        g, n, d = x.shape
        k = direction[0].shape[0]
        pred, x0 = dd.diffusion.p_sample(self, x.reshape(g * n, d), None, adj, direction,
                                         self._flat_labels(labels, k), None, t, x_self_cond, graphset)
        return pred.view(g, n, d), x0.view(g, n, -1)

    def _logmap_centres(self, x, labels, u0):
        cen, k = self._centres(u0)
        g, n, d = x.shape
        lab = ops.labels_dev(self._flat_labels(labels, k), x.device)
        x.copy_(ops.map_at("hd_pb_logmap_at", cen, lab, x.reshape(g * n, d), 1.0).view(g, n, d))
        return x

    def p_sample_loop(self, x, adj, lables, u0, return_all_timesteps=False, graphset=True, img=None):
        """:302-352: no tran_direction on the initial noise here (commented out :310)"""
        if not graphset:
            raise NotImplementedError("gcndec denoiser not built; use graphset=True")
        g, n, d = x.shape
        img = torch.randn(x.shape, device=x.device) if img is None else img.clone()
        direction = self.directions(u0)
        # fold + upload the labels once for the whole loop (1-D int32 device tensor passes through)
        lables = ops.labels_dev(self._flat_labels(lables, direction[0].shape[0]), x.device)
        h = self._logmap_centres(x, lables, u0)            # mutates x like the reference
        m = self.moments(h)
        img = img + self._apply_moments(h, m, 0)
        for t in reversed(range(0, self.num_timesteps)):
            img, x_start = self.p_sample(img, None, 1, direction, lables, u0, t, None, True)
        ret = self._apply_moments(img, m, 1)
        cen, k = self._centres(u0)
        lab = ops.labels_dev(self._flat_labels(lables, k), x.device)
        return ops.map_at("hd_pb_expmap_at", cen, lab, ret.reshape(g * n, d), 1.0).view(g, n, d)

    def forward(self, x, adj, labels, u0, restrict=False, graphset=False, t=None, noise=None):
        """:354-384 training loss; differentiable w.r.t. the denoiser parameters (HIP backward
        kernels, hypdiff_amd/autograd.py) -- the loss diff.hyperdiff_graphset trains (diff.py:230-260)"""
        if not graphset:
            raise NotImplementedError("gcndec denoiser not built; use graphset=True")
        x = x.to(torch.float64)
        b = x.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long() if t is None else t
        noise = torch.randn_like(x) if noise is None else noise
        direction = self.directions(u0)
        if not restrict:
            x_start = self.manifold.logmap0(x, c=1.0)
        else:
            x_start = self._logmap_centres(x, labels, u0)
        x_start = self.normalization(x_start)
        xt = self.q_sample(x_start=x_start, t=t, direction=direction, noise=noise, labels=labels, u0=u0, restrict=True)
        out = self.mlp(xt, self._t_frac(t))
        target = x_start.to(torch.float32)
        return hag.mse_loss(out.reshape(-1, out.shape[-1]), target.reshape(-1, target.shape[-1]))
