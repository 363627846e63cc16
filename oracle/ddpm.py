"""Oracle: graph-level latent DDPM (torch CPU) -- TEST INFRASTRUCTURE ONLY.
Restates ddpm.py:9-111 and ddpm_utils.py:32-53 of the reference functionally."""
import math

import torch
import torch.nn.functional as F


def mlp(x, layers):
    """ddpm.py:17-27; layers = [(W, b), ...]"""
    last = len(layers) - 1
    for i, (w, b) in enumerate(layers):
        if i == 0:
            x = F.gelu(F.linear(x, w, b))
        elif i == last:
            x = F.linear(x, w, b)
        else:
            x = x + F.gelu(F.linear(x, w, b))
    return x


def backbone(x, idx, m1, emb, m2):
    """ddpm.py:39-41"""
    return mlp(mlp(x, m1) + emb[idx], m2)


def schedule(T, beta_1, beta_T):
    beta = torch.linspace(start=beta_1, end=beta_T, steps=T)
    return beta, torch.cumprod(1 - beta, dim=0)


def noise_forward(x, idx, eps, alpha_bars, restrict=False, deta=0.01, c=1.0):
    """ddpm.py:97-104"""
    ab = alpha_bars[idx][:, None]
    if not restrict:
        return torch.sqrt(ab) * x + torch.sqrt(1 - ab) * eps
    pri = (deta * torch.tanh(math.sqrt(c) * idx / 2000)).reshape(-1, 1)
    return pri * x + torch.sqrt(ab) * x + torch.sign(x) * torch.sqrt(1 - ab) * eps


def sampler_step(x, idx, eps, noise, beta, alpha_bars):
    """ddpm_utils.py:37-45 (pred_noise, restrict False)"""
    alphas = 1 - beta
    prev = torch.cat([torch.ones(1), alpha_bars[:-1]])
    sqrt_tilde_beta = torch.sqrt((1 - prev[idx]) / (1 - alpha_bars[idx]) * beta[idx])
    mu = torch.sqrt(1 / alphas[idx]) * (x - beta[idx] / torch.sqrt(1 - alpha_bars[idx]) * eps)
    return mu + sqrt_tilde_beta * noise
