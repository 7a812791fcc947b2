"""EDM (Karras et al. 2022) parameterisation -- same surface as reference ``diff_params/edm.py`` (``EDM(type, sde_hp)``;
``cskip/cout/cin/cnoise``, ``Tweedie2score``, ``score2Tweedie``, ``_mean``, ``_std``, ``_ode_integrand``, ``denoiser``).
Training-only methods (``loss_fn``, ``sample_time_training``) are out of scope."""
from __future__ import annotations

import torch

from .shared import SDE


class EDM(SDE):
    def __init__(self, type, sde_hp):
        super().__init__(type, sde_hp)
        self.sigma_data = self.sde_hp.sigma_data
        self.sigma_min = self.sde_hp.sigma_min
        self.sigma_max = self.sde_hp.sigma_max
        self.rho = self.sde_hp.rho

    def sample_prior(self, shape):
        return torch.randn(shape)

    def cskip(self, sigma):       # reference edm.py:44-51
        return self.sigma_data ** 2 * (sigma ** 2 + self.sigma_data ** 2) ** -1

    def cout(self, sigma):        # edm.py:53-59
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cin(self, sigma):         # edm.py:61-67
        return (self.sigma_data ** 2 + sigma ** 2) ** (-0.5)

    def cnoise(self, sigma):      # edm.py:69-75
        if not torch.is_tensor(sigma):      # numpy float32 scalar (SDE.host_scalars)
            import numpy as np
            return np.float32(1 / 4) * np.log(np.float32(sigma))
        return (1 / 4) * torch.log(sigma)

    def lambda_w(self, sigma):    # edm.py:77-81
        return (sigma * self.sigma_data) ** (-2) * (self.sigma_data ** 2 + sigma ** 2)

    def Tweedie2score(self, tweedie, xt, t, *args, **kwargs):   # edm.py:83-84
        return (tweedie - self._mean(xt, t)) / self._std(t) ** 2

    def score2Tweedie(self, score, xt, t, *args, **kwargs):     # edm.py:86-87
        return self._std(t) ** 2 * score + self._mean(xt, t)

    def _mean(self, x, t):
        return x

    def _std(self, t):
        return t

    def _ode_integrand(self, x, t, score):                      # edm.py:95-96
        return -t * score
