"""Stochastic Euler-Heun sampler -- same surface as reference ``testing/EulerHeunSampler.py:6-107``."""
from __future__ import annotations

import torch

from .Sampler import Sampler


class EulerHeunSampler(Sampler):
    def __init__(self, model, diff_params, args):
        super().__init__(model, diff_params, args)
        sp = self.args.tester.sampling_params
        self.Schurn, self.Snoise, self.Stmin, self.Stmax = sp.Schurn, sp.Snoise, sp.Stmin, sp.Stmax
        self.order = sp.order

    def initialize_x(self, shape, device, schedule):
        return schedule[0] * self._randn(shape, device)

    def get_gamma(self, t):
        """gamma = min(Schurn/N, sqrt(2)-1) where Stmin < t < Stmax, N = len(t) = T+1 (reference :24-39)."""
        N = t.shape[0]
        gamma = torch.zeros(t.shape).to(t.device)
        indexes = torch.logical_and(t > self.Stmin, t < self.Stmax)
        gamma[indexes] = gamma[indexes] + torch.min(torch.Tensor([self.Schurn / N, 2 ** (1 / 2) - 1])).to(t.device)
        return gamma

    @staticmethod
    def _scalar(v):
        """schedule entries as host fp32 scalars: no device round trip inside the loop (a CUDA 0-dim tensor forces a host sync per use)"""
        import numpy as np
        return np.float32(float(v))                         # fp32 host arithmetic, like the reference's fp32 0-dim tensors

    def stochastic_timestep(self, x, t, gamma, Snoise=1):
        t_hat = t + gamma * t
        epsilon = self._randn(x.shape, x.device) * Snoise     # Snoise from the config never reaches here (reference :41,50)
        if x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and getattr(self, "use_hip_update", True):      # the fused kernels are fp32
            from . import _hipops
            return _hipops.perturb(x, epsilon, float((t_hat ** 2 - t ** 2) ** (1 / 2))), t_hat
        x_hat = x + ((t_hat ** 2 - t ** 2) ** (1 / 2)) * epsilon
        return x_hat, t_hat

    def step(self, x_i, t_i, t_iplus1, gamma_i, blind=False):
        t_i, t_iplus1, gamma_i = self._scalar(t_i), self._scalar(t_iplus1), self._scalar(gamma_i)
        with torch.no_grad():
            x_hat, t_hat = self.stochastic_timestep(x_i, t_i, gamma_i)
            x_den = self.get_Tweedie_estimate(x_hat, t_hat)
            score = self.Tweedie2score(x_den, x_hat, t_hat)
            ode_integrand = self.diff_params._ode_integrand(x_hat, t_hat, score)
            dt = t_iplus1 - t_hat
            if t_iplus1 != 0 and self.order == 2:
                t_prime = t_iplus1
                x_prime = x_hat + dt * ode_integrand
                x_den = self.get_Tweedie_estimate(x_prime, t_prime)
                score = self.Tweedie2score(x_den, x_prime, t_prime)
                ode_integrand_next = self.diff_params._ode_integrand(x_prime, t_prime, score)
                x_iplus1 = x_hat + dt * (.5 * (ode_integrand + ode_integrand_next))
            else:
                x_iplus1 = x_hat + dt * ode_integrand
            return x_iplus1, x_den

    def predict(self, shape, device, blind=False):
        t = self.create_schedule()                          # stays on the host: the loop below never waits for the device
        x = self.initialize_x(shape, device, t)
        tl, gl = t.tolist(), self.get_gamma(t).tolist()
        for i in range(0, self.T, 1):
            self.step_counter = i
            x, x_den = self.step(x, tl[i], tl[i + 1], gl[i], blind)
        return x.detach()

    def predict_unconditional(self, shape, device):
        self.y = None
        self.degradation = None
        return self.predict(shape, device)

    def predict_conditional(self, *args, **kwargs):
        raise NotImplementedError
