"""DDPM ancestral sampler on HIP kernels -- drop-in for the reference's ddpm_utils.py
(`DDPMSampler`, restrict=False / target='pred_noise', the path diff_graphset.py:78-84
takes; the restrict=True branch of the reference reads an undefined attribute,
ddpm_utils.py:25, and is dead)."""
import torch

from . import ops


class DDPMSampler():
    def __init__(self, beta_1, beta_T, T, diffusion_fn, device, shape):
        """ddpm_utils.py:6-22"""
        self.betas = torch.linspace(start=beta_1, end=beta_T, steps=T)
        self.alphas = 1 - self.betas
        self.alpha_bars = torch.cumprod(1 - torch.linspace(start=beta_1, end=beta_T, steps=T), dim=0)
        self.alpha_prev_bars = torch.cat([torch.Tensor([1]), self.alpha_bars[:-1]])
        self.shape = shape
        self.deta = 0.01
        self.diffusion_fn = diffusion_fn
        self.device = device
        # per-step fp32 coefficients, same expressions as ddpm_utils.py:39-44
        self._c1 = torch.sqrt(1 / self.alphas)
        self._c2 = self.betas / torch.sqrt(1 - self.alpha_bars)
        self._c3 = torch.sqrt((1 - self.alpha_prev_bars) / (1 - self.alpha_bars) * self.betas)

    def one_step(self, x, idx, noise=None):
        """one iteration of _one_diffusion_step (ddpm_utils.py:37-45)"""
        eps = self.diffusion_fn(x, idx, get_target=False)
        out = torch.empty_like(x)
        ops._lib.call("hd_ddpm_step", ops._p(x), ops._p(eps), ops._p(noise), float(self._c1[idx]), float(self._c2[idx]),
                      float(self._c3[idx]), ops._p(out), x.numel(), ops._stream())
        return out

    def _one_diffusion_step(self, x, direction, restrict, target):
        if restrict or target != 'pred_noise':
            raise NotImplementedError("only restrict=False / target='pred_noise' (the live path of the reference)")
        x = x.to(torch.float32).contiguous()
        for idx in reversed(range(len(self.alpha_bars))):
            noise = None if idx == 0 else torch.randn_like(x)
            x = self.one_step(x, idx, noise)
            yield x

    @torch.no_grad()
    def sampling(self, sampling_number, x, only_final=False, restrict=False, target='pred_noise'):
        """ddpm_utils.py:80-96"""
        direction = torch.sign(x)
        sample = torch.randn([sampling_number, *self.shape]).to(device=self.device).squeeze()
        sampling_list = []
        final = None
        for sample in self._one_diffusion_step(sample, direction, restrict, target):
            final = sample
            if not only_final:
                sampling_list.append(final)
        return final if only_final else torch.stack(sampling_list)
