"""Oracle: EDM preconditioning + Euler-Heun / Euler-Heun-DPS samplers, PyTorch fp32 on CPU, B=1 semantics.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates reference ``diff_params/edm.py:44-96``,
``diff_params/shared.py:98-120``, ``testing/Sampler.py:39-72``, ``testing/EulerHeunSampler.py:24-104`` and
``testing/EulerHeunSamplerDPS.py:25-204``.  Gaussian draws come from an injected ``noise`` object
(``randn(shape)`` / ``rand(shape)``) in the reference's call order, so results are reproducible against
fixtures recorded from the reference with ``torch.randn`` patched to the same stream.
"""
from __future__ import annotations

import numpy as np
import torch

from .operators_ref import get_loss_ref


class NoiseStream:
    """k-th draw = RandomState(seed*100003 + k) of the requested shape: float32 values (cast to the default dtype, so the fp64 arbiter
    mode of ``oracle.precision`` sees exactly the same draws)."""

    def __init__(self, seed):
        self.seed, self.k = int(seed), 0

    def _rs(self):
        rs = np.random.RandomState((self.seed * 100003 + self.k) % (2 ** 32))
        self.k += 1
        return rs

    def randn(self, shape):
        return torch.from_numpy(self._rs().standard_normal(tuple(shape)).astype(np.float32)).to(device=torch.get_default_device(), dtype=torch.get_default_dtype())

    def rand(self, shape):
        return torch.from_numpy(self._rs().random_sample(tuple(shape)).astype(np.float32)).to(device=torch.get_default_device(), dtype=torch.get_default_dtype())


# --------------------------------------------------------------------------- EDM (reference diff_params/edm.py)
class EDMRef:
    def __init__(self, sde_hp):
        self.sigma_data = sde_hp.sigma_data

    def cskip(self, s):   # edm.py:44-51
        return self.sigma_data ** 2 * (s ** 2 + self.sigma_data ** 2) ** -1

    def cout(self, s):    # edm.py:53-59
        return s * self.sigma_data * (self.sigma_data ** 2 + s ** 2) ** (-0.5)

    def cin(self, s):     # edm.py:61-67
        return (self.sigma_data ** 2 + s ** 2) ** (-0.5)

    def cnoise(self, s):  # edm.py:69-75
        return (1 / 4) * torch.log(s)

    def denoiser(self, xn, net, t):
        """shared.py:98-120. xn (B,1,L); t 0-dim tensor."""
        sigma = t.reshape(1, 1, 1)
        cn = self.cnoise(sigma.squeeze()).repeat(xn.shape[0])
        return self.cskip(sigma) * xn + self.cout(sigma) * net(self.cin(sigma) * xn, cn)

    def tweedie2score(self, d, x, t):   # edm.py:83-84
        return (d - x) / t ** 2

    def ode_integrand(self, x, t, score):  # edm.py:95-96
        return -t * score


def create_schedule(sde_hp, T):
    """Sampler.py:39-56 (edm): note a/(T-1) with a = 0..T and t[-1] = 0."""
    a = torch.arange(0, T + 1)
    r = sde_hp.rho
    t = (sde_hp.sigma_max ** (1 / r) + a / (T - 1) * (sde_hp.sigma_min ** (1 / r) - sde_hp.sigma_max ** (1 / r))) ** r
    t[-1] = 0
    return t


def get_gamma(t, sp):
    """EulerHeunSampler.py:24-39: gamma = min(Schurn/N, sqrt2-1) where Stmin < t < Stmax, N = len(t) = T+1."""
    N = t.shape[0]
    g = torch.zeros(t.shape)
    idx = torch.logical_and(t > sp.Stmin, t < sp.Stmax)
    g[idx] = g[idx] + torch.min(torch.Tensor([sp.Schurn / N, 2 ** (1 / 2) - 1]))
    return g


class EulerHeunRef:
    """Plain (unconditional) Euler-Heun -- EulerHeunSampler.py:47-104.  Returns x (not x_den)."""

    def __init__(self, net, edm, args, noise):
        self.net, self.edm, self.args, self.noise = net, edm, args, noise
        sp = args.tester.sampling_params
        self.sp, self.T, self.order = sp, sp.T, sp.order
        self.sde_hp = sp.sde_hp

    def tweedie(self, x, t):
        return self.edm.denoiser(x.unsqueeze(1), self.net, t).squeeze(1)

    def stochastic_timestep(self, x, t, gamma):
        t_hat = t + gamma * t
        eps = self.noise.randn(x.shape)
        return x + ((t_hat ** 2 - t ** 2) ** (1 / 2)) * eps, t_hat

    def step(self, x, t, t_next, gamma):
        with torch.no_grad():
            x_hat, t_hat = self.stochastic_timestep(x, t, gamma)
            x_den = self.tweedie(x_hat, t_hat)
            d = self.edm.ode_integrand(x_hat, t_hat, self.edm.tweedie2score(x_den, x_hat, t_hat))
            dt = t_next - t_hat
            if t_next != 0 and self.order == 2:
                x_p = x_hat + dt * d
                x_den = self.tweedie(x_p, t_next)
                d2 = self.edm.ode_integrand(x_p, t_next, self.edm.tweedie2score(x_den, x_p, t_next))
                return x_hat + dt * (.5 * (d + d2)), x_den
            return x_hat + dt * d, x_den

    def predict_unconditional(self, shape):
        t = create_schedule(self.sde_hp, self.T)
        x = t[0] * self.noise.randn(shape)
        gamma = get_gamma(t, self.sp)
        for i in range(self.T):
            x, _ = self.step(x, t[i], t[i + 1], gamma[i])
        return x.detach()


class EulerHeunDPSRef(EulerHeunRef):
    """EulerHeunSamplerDPS.py:25-204 (warm init none / reverb_scaled / wpe_scaled)."""

    def __init__(self, net, edm, args, noise):
        super().__init__(net, edm, args, noise)
        self.ps = args.tester.posterior_sampling
        self.zeta = self.ps.zeta

    def initialize_x(self, shape, t):
        mode = self.ps.warm_initialization.mode
        if mode == "none":
            return t[0] * self.noise.randn(shape)
        if mode == "reverb_scaled":
            return self.ps.warm_initialization.scaling_factor * self.y.clone() / self.y.std() + t[0] * self.noise.randn(shape)
        if mode == "wpe_scaled":      # :32-54 through the restated nara_wpe (oracle/wpe_ref.py; that package's parity is unpinned)
            from .wpe_ref import wpe_warm_start_estimate
            w = self.ps.warm_initialization.wpe
            xp = wpe_warm_start_estimate(self.y.detach().cpu().numpy(), taps=w.taps, delay=w.delay, iterations=w.iterations)
            xp = torch.from_numpy(xp).to(device=self.y.device, dtype=self.y.dtype)
            xp = self.ps.warm_initialization.scaling_factor * xp / xp.std()
            return xp + t[0] * self.noise.randn(shape)
        raise NotImplementedError(mode)

    def likelihood_score(self, x_den, x):
        """:61-69 -- gradient through denoiser + network by autograd; normaliser uses args.exp.audio_len."""
        y_hat = self.operator.degradation(x_den)
        rec = self.rec_loss(self.y, y_hat)
        g = torch.autograd.grad(outputs=rec, inputs=x)[0]
        normguide = torch.norm(g) / (self.args.exp.audio_len ** 0.5)
        return self.zeta / (normguide + 1e-8) * g, rec

    def optimize_op(self, x_den, t):
        """:71-113."""
        op, hp, reg = self.operator, self.ps.blind_hp, self.ps.RIR_noise_regularization
        for _ in range(hp.op_updates_per_step):
            for p in op.params + op.params_phases:
                p.requires_grad = True
            op.update_H()
            loss = self.rec_loss_params(self.y, op.degradation(x_den))
            if self.rir_reg_loss is not None:
                rir = op.get_time_RIR()
                n = self.noise.randn(rir.shape)
                t_op = max(min(float(t), reg.crop_sigma_max), reg.crop_sigma_min)
                loss = loss + self.rir_reg_loss(rir, (rir + t_op * n).detach())
            self.optim.zero_grad()
            loss.backward()
            self.optim.step()
            for p in op.params:
                p.detach_()
            op.project_params()
            for p in op.params:
                p.requires_grad = True

    def _eval(self, x_in, t, blind, rescale=True):
        x_in.requires_grad = True
        x_den = self.tweedie(x_in, t)
        if blind:
            self.optimize_op(x_den.clone().detach(), t)
        lh, _ = self.likelihood_score(x_den, x_in)
        x_in.detach_()
        csm = self.ps.constraint_speech_magnitude
        if csm.use and rescale:                       # :127-129 -- the second-order evaluation (:139-149) does not rescale
            x_den = csm.speech_scaling / x_den.detach().std() * x_den
        score = self.edm.tweedie2score(x_den, x_in, t)
        return self.edm.ode_integrand(x_in, t, score) + lh, x_den

    def step(self, x, t, t_next, gamma, blind):
        """:115-157."""
        x_hat, t_hat = self.stochastic_timestep(x, t, gamma)
        d, x_den = self._eval(x_hat, t_hat, blind)
        dt = t_next - t_hat
        if t_next != 0 and self.order == 2:
            x_p = (x_hat + dt * d).detach()
            d2, x_den = self._eval(x_p, t_next, blind, rescale=False)
            x_new = x_hat + dt * (.5 * (d + d2))
        else:
            x_new = x_hat + dt * d
        return x_new.detach(), x_den.detach()

    def predict_conditional(self, y, operator, shape=None, blind=False, trace=None):
        """:183-204 + :159-178.  Returns x_den of the last step."""
        self.operator, self.y = operator, y
        self.rec_loss = get_loss_ref(self.ps.rec_loss, operator)
        if blind:
            hp = self.ps.blind_hp
            self.rec_loss_params = get_loss_ref(self.ps.rec_loss_params, operator)
            self.optim = torch.optim.Adam(operator.params + operator.params_phases, lr=hp.lr_op,
                                          weight_decay=hp.weight_decay, betas=(hp.beta1, hp.beta2))
            self.rir_reg_loss = get_loss_ref(self.ps.RIR_noise_regularization.loss, operator)
        shape = y.shape if shape is None else shape
        t = create_schedule(self.sde_hp, self.T)
        x = self.initialize_x(shape, t)
        gamma = get_gamma(t, self.sp)
        x_den = None
        for i in range(self.T):
            x, x_den = self.step(x, t[i], t[i + 1], gamma[i], blind)
            if trace is not None:
                trace.append((x.clone(), x_den.clone()))
        return x_den.detach()
