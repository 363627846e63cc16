"""Oracle: direction-diffusion step (torch CPU, vectorised over rows).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
models/direction_diffusion.py of the reference with the per-row Python loops
(:158-161, :305-306, :328-330, :350-352) replaced by row-vectorised tensor
ops.  dtype promotion is reproduced on purpose: the schedule tables are fp64,
`getpri` is fp32, the denoiser runs in fp32.
"""
import math

import torch
import torch.nn.functional as F

from .manifolds import pb_expmap, pb_logmap, pb_logmap0


def schedule(T):
    """models/direction_diffusion.py:71-81,94-107,173-184 -- linear beta schedule
    with scale 1000/T, everything fp64."""
    scale = 1000 / T
    betas = torch.linspace(scale * 0.0001, scale * 0.02, T, dtype=torch.float64)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    return dict(
        betas=betas, alphas=alphas, alphas_cumprod=ac,
        sqrt_alphas_cumprod=torch.sqrt(ac),
        sqrt_one_minus_alphas_cumprod=torch.sqrt(1.0 - ac),
        sqrt_recip_alphas_cumprod=torch.sqrt(1.0 / ac),
        sqrt_recipm1_alphas_cumprod=torch.sqrt(1.0 / ac - 1),
    )


def getpri(t, c=1.0, prior=1, horizon=500):
    """:191-196 -- prior * tanh(sqrt(c) * t / 500); int64 t times a python float is
    fp32 in torch, so the result is an fp32 (b, 1) column."""
    out = prior * torch.tanh(math.sqrt(c) * t / horizon)
    return out.reshape(t.shape[0], 1)


def sign_table(u0, c=1.0):
    """:298-300 / :342-344 + :159 -- sign(logmap0(u0[k])) as a (K, D) table."""
    return torch.sign(pb_logmap0(torch.as_tensor(u0), c))


def tran_direction_(noise, signs, labels):
    """:154-163, in place on `noise` like the reference."""
    labels = torch.as_tensor(labels, dtype=torch.long)
    noise.copy_(noise * signs[labels])
    return noise


def tran_direction_loop_(noise, direction, labels):
    """:154-163 the way the reference EXECUTES it: one Python iteration per row, each a sign(), a
    row multiply and a row store.  Same values as tran_direction_; used by bench.py's loop-faithful
    CPU baseline (`kind: "loop"`), where this loop is most of the reference's p_sample time."""
    for i in range(noise.shape[0]):
        noise[i] = noise[i] * torch.sign(direction[int(labels[i])])
    return noise


def q_sample(x_start, t, noise, signs, labels, tables, restrict=False, c=1.0, loop_direction=None):
    """:198-213.  Both branches of the reference return the same expression.  loop_direction: the
    list of direction vectors; when given, the sign flip runs as the reference's per-row loop."""
    if restrict and loop_direction is not None:
        tran_direction_loop_(noise, loop_direction, labels)
    elif restrict:
        tran_direction_(noise, signs, labels)
    a = tables["sqrt_alphas_cumprod"].gather(-1, t).reshape(-1, 1)
    b = tables["sqrt_one_minus_alphas_cumprod"].gather(-1, t).reshape(-1, 1)
    return a * x_start + b * noise + getpri(t, c) * x_start


def predict_noise_from_start(x_t, t, x0, tables, c=1.0):
    """:230-236."""
    ra = tables["sqrt_recip_alphas_cumprod"].gather(-1, t).reshape(-1, 1)
    rm1 = tables["sqrt_recipm1_alphas_cumprod"].gather(-1, t).reshape(-1, 1)
    return ((ra + getpri(t, c)) * x_t - x0) / rm1


def mlp_denoiser(x, t, params):
    """:56-69.  params = [(W1,b1), ..., (W5,b5)], exact-erf GELU between layers."""
    h = torch.cat((x.to(torch.float32), t.unsqueeze(1)), dim=1).float()
    for i, (w, b) in enumerate(params):
        h = F.linear(h, w, b)
        if i + 1 < len(params):
            h = F.gelu(h)
    return h


def p_sample(x, t, signs, labels, tables, params, c=1.0, loop_direction=None):
    """:272-291 with graphset=True (mlp denoiser).  Returns (x_{t-1}, x0_hat)."""
    b = x.shape[0]
    tb = torch.full((b,), t, dtype=torch.long)
    x0 = mlp_denoiser(x, tb, params)          # integer t fed to the denoiser (:276-278)
    # model_predictions (:248-260) derives the noise from the UNCLAMPED denoiser output;
    # p_mean_variance clamps x0 in place only afterwards (:266-267)
    eps = predict_noise_from_start(x, tb, x0, tables, c)
    x0.clamp_(-1.0, 1.0)
    t2 = torch.full((b,), t - 1 if t > 0 else 0, dtype=torch.long)
    return q_sample(x0, t2, eps, signs, labels, tables, restrict=True, c=c, loop_direction=loop_direction), x0


def normalization(x):
    """:145-149 -- global mean, unbiased std."""
    return (x - torch.mean(x)) / torch.std(x)


def renormal(x, h):
    """:150-153."""
    return x * torch.std(h) + torch.mean(h)


def forward_loss(x, labels, u0, t, noise, tables, params, c=1.0):
    """:335-363 with restrict=True, graphset=True; `t` and `noise` are injected
    instead of drawn.  Returns (loss, x_start, x_t)."""
    T = tables["betas"].shape[0]
    labels = torch.as_tensor(labels, dtype=torch.long)
    x = x.to(torch.float64)
    signs = sign_table(u0, c)
    x_start = pb_logmap(u0[labels], x, c).to(x.dtype)
    x_start = normalization(x_start)
    xt = q_sample(x_start, t, noise, signs, labels, tables, restrict=True, c=c)
    out = mlp_denoiser(xt, t / T, params)
    return F.mse_loss(out, x_start.to(torch.float32)), x_start, xt


def p_sample_loop(x, labels, u0, img, tables, params, c=1.0):
    """:293-332 with graphset=True; the initial Gaussian `img` is injected."""
    T = tables["betas"].shape[0]
    labels = torch.as_tensor(labels, dtype=torch.long)
    signs = sign_table(u0, c)
    img = tran_direction_(img.clone(), signs, labels)
    h = pb_logmap(u0[labels], x, c).to(x.dtype)
    img = img + normalization(h)
    for t in reversed(range(T)):
        img, _ = p_sample(img, t, signs, labels, tables, params, c)
    ret = renormal(img, h)
    return pb_expmap(ret, u0[labels], c).to(ret.dtype)
