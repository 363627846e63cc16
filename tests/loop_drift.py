"""The 200-step sampling loop (models/direction_diffusion.py:293-332) is 200 chained fp32 denoiser evaluations: rounding
differences between two correct fp32 implementations compound step by step.  Instead of a bare loose tolerance on the
end point, the tests measure that growth: the oracle's loop with the denoiser in fp32 (as the reference runs it) and
with the denoiser in fp64 (the exact yardstick) give, per step, the drift that fp32 arithmetic itself causes;
another implementation (the HIP path, or the reference's own golden end point) must stay within a small multiple of it."""
import numpy as np
import torch

from oracle import diffusion as odiff, manifolds as om


def mlp64(x, tt, p):
    h = torch.cat((x.double(), tt.unsqueeze(1).double()), dim=1)
    for i, (w, b) in enumerate(p):
        h = torch.nn.functional.linear(h, w.double(), b.double())
        if i + 1 < len(p):
            h = torch.nn.functional.gelu(h)
    return h


def oracle_loop_states(x, labels, u0, img, tables, params, exact, c=1.0):
    """oracle.diffusion.p_sample_loop step by step: returns ([img after every step, first entry = the start], end point).
    exact=True evaluates the denoiser in fp64 (everything else in the loop is fp64 already after the first step)."""
    T = tables["betas"].shape[0]
    labels = torch.as_tensor(labels, dtype=torch.long)
    signs = odiff.sign_table(u0, c)
    img = odiff.tran_direction_(img.clone(), signs, labels)
    h = om.pb_logmap(u0[labels], x, c).to(x.dtype)
    img = img + odiff.normalization(h)
    states = [img.clone()]
    old = odiff.mlp_denoiser
    if exact:
        odiff.mlp_denoiser = mlp64
    try:
        for t in reversed(range(T)):
            img, _ = odiff.p_sample(img, t, signs, labels, tables, params, c)
            states.append(img.clone())
    finally:
        odiff.mlp_denoiser = old
    ret = odiff.renormal(img, h)
    return states, om.pb_expmap(ret, u0[labels], c).to(ret.dtype)


def drift(states_a, states_b):
    """per step: max |a - b| / max |b|"""
    return np.array([float((a.double() - b.double()).abs().max() / b.double().abs().max()) for a, b in zip(states_a, states_b)])
