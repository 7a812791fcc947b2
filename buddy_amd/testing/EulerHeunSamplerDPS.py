"""Euler-Heun diffusion-posterior-sampling solver -- same surface as reference
``testing/EulerHeunSamplerDPS.py:15-204`` (``predict_conditional(y, operator, shape, blind)``; attribute ``operator``).

Batched: ``y`` may be ``(B, L)``; every reduction the reference takes over "the tensor" (``y.std()``,
``torch.norm(rec_grads)``, ``x_den.std()``) is taken per utterance, the operator holds per-utterance parameters, Adam
is elementwise -- so row b of a batched run equals the reference's B=1 run on utterance b.
The likelihood gradient flows by autograd through the operator/loss and then through the hand-written HIP network
VJP (``NCSNppTime`` is an autograd Function)."""
from __future__ import annotations

import torch

from ..utils.losses import get_loss
from .EulerHeunSampler import EulerHeunSampler


def _row_std(v):
    return v.std(dim=-1, keepdim=True)       # unbiased, like Tensor.std()


class EulerHeunSamplerDPS(EulerHeunSampler):
    def __init__(self, model, diff_params, args):
        super().__init__(model, diff_params, args)
        self.zeta = self.args.tester.posterior_sampling.zeta
        self._hip_op = False
        self._hip_loss = False
        self.use_hip_update = True      # fused elementwise tail on the GPU (False: torch expressions, as on the CPU)

    def initialize_x(self, shape, device, schedule):
        wi = self.args.tester.posterior_sampling.warm_initialization
        if wi.mode == "none":
            return schedule[0] * self._randn(shape, device)
        if wi.mode == "reverb_scaled":
            return wi.scaling_factor * self.y.clone() / _row_std(self.y) + schedule[0] * self._randn(shape, device)
        if wi.mode == "wpe_scaled":
            from ..utils.wpe import wpe_dereverb    # nara_wpe restated (third-party, parity unpinned)
            x_pred = wpe_dereverb(self.y, taps=wi.wpe.taps, delay=wi.wpe.delay, iterations=wi.wpe.iterations)
            if x_pred.shape[-1] < self.y.shape[-1]:
                x_pred = torch.nn.functional.pad(x_pred, (0, self.y.shape[-1] - x_pred.shape[-1]))
            x_pred = wi.scaling_factor * x_pred / _row_std(x_pred)
            return x_pred + schedule[0] * self._randn(shape, device)
        raise NotImplementedError

    def get_likelihood_score(self, x_den, x, t):
        if self._hip_op or self._hip_loss:
            rec = self.operator.hip_rec_loss(x_den)               # fused HIP loss + analytic d/dx_den; autograd continues into the net VJP
        else:
            y_hat = self.operator.degradation(x_den, mode="waveform")
            rec = self.rec_loss(self.y, y_hat)                    # sum over utterances: gradients decouple per row
        rec_grads = torch.autograd.grad(outputs=rec, inputs=x)[0]
        normguide = torch.linalg.vector_norm(rec_grads, dim=-1, keepdim=True) / (self.args.exp.audio_len ** 0.5)
        return self.zeta / (normguide + 1e-8) * rec_grads, rec

    def optimize_op(self, x_den, t):
        if self._hip_op:
            return self.operator.hip_optimize(x_den, t)           # the whole loop below as one library call (buddy_blindop_optimize)
        ps = self.args.tester.posterior_sampling
        for _ in range(ps.blind_hp.op_updates_per_step):
            for p in self.operator.params:
                p.requires_grad = True
            for p in self.operator.params_phases:
                p.requires_grad = True
            self.operator.update_H()
            y_hat = self.operator.degradation(x_den, mode="waveform")
            if self.rec_loss_params is not None:
                loss = self.rec_loss_params(self.y, y_hat)
                assert not torch.isnan(loss).any(), "rec_loss is Nan"
            else:
                loss = 0.
            if self.RIR_noise_regularization_loss is not None:
                rir_time = self.operator.get_time_RIR()
                if rir_time.dim() == 1:
                    rir_time = rir_time.unsqueeze(0)
                rir_noise = self.operator._randn(rir_time.shape[1:]) if hasattr(self.operator, "_randn") else torch.randn_like(rir_time)
                reg = ps.RIR_noise_regularization
                t_op = max(min(float(t), reg.crop_sigma_max), reg.crop_sigma_min)
                rir_noisy = rir_time + t_op * rir_noise
                loss = loss + self.RIR_noise_regularization_loss(rir_time, rir_noisy.detach())
            assert not torch.isnan(loss).any(), "loss is Nan"
            self.optimizer_operator.zero_grad()
            loss.backward()
            self.optimizer_operator.step()
            for p in self.operator.params:
                p.detach_()
            self.operator.project_params()
            for p in self.operator.params:
                p.requires_grad = True

    def _guided_eval(self, x_in, t, blind, rescale=True):
        """one guided evaluation; ``rescale=False`` is the Heun corrector, which the reference leaves un-rescaled (:139-149)"""
        x_in.requires_grad = True
        x_den = self.get_Tweedie_estimate(x_in, t)
        if blind:
            self.optimize_op(x_den.clone().detach(), t)
        lh_score, _ = self.get_likelihood_score(x_den, x_in, t)
        x_in.detach_()
        csm = self.args.tester.posterior_sampling.constraint_speech_magnitude
        if csm.use and rescale:
            x_den = csm.speech_scaling / _row_std(x_den.detach()) * x_den
        score = self.Tweedie2score(x_den, x_in, t)
        return self.diff_params._ode_integrand(x_in, t, score) + lh_score, x_den

    def _eval_parts(self, x_in, t, blind):
        """network + operator part of one guided evaluation: (likelihood score, raw Tweedie estimate), both detached."""
        x_in.requires_grad = True
        x_den = self.get_Tweedie_estimate(x_in, t)
        if blind:
            self.optimize_op(x_den.clone().detach(), t)
        lh_score, _ = self.get_likelihood_score(x_den, x_in, t)
        x_in.detach_()
        return lh_score.detach(), x_den.detach()

    def _step_hip(self, x_i, t_i, t_iplus1, gamma_i, blind):
        """same arithmetic as step() with the elementwise tail fused into HIP kernels (buddy_perturb / buddy_dps_update)."""
        from . import _hipops
        csm = self.args.tester.posterior_sampling.constraint_speech_magnitude
        x_hat, t_hat = self.stochastic_timestep(x_i, t_i, gamma_i)
        lh, x_den = self._eval_parts(x_hat, t_hat, blind)
        scale = (csm.speech_scaling / _hipops.row_std(x_den)).reshape(-1) if csm.use else None
        dt = float(t_iplus1 - t_hat)
        if t_iplus1 != 0 and self.order == 2:
            x_prime, d1, _ = _hipops.dps_update(x_hat, x_den, lh, scale, x_hat, None, float(t_hat), dt, 0.0, 1.0, want_d=True)
            lh2, x_den2 = self._eval_parts(x_prime, t_iplus1, blind)
            # the reference rescales x_den only in the first evaluation (:127-129); the corrector uses and returns the raw estimate (:139-149)
            x_next, _, x_den_out = _hipops.dps_update(x_prime, x_den2, lh2, None, x_hat, d1, float(t_iplus1), dt, 0.5, 0.5)
        else:
            x_next, _, x_den_out = _hipops.dps_update(x_hat, x_den, lh, scale, x_hat, None, float(t_hat), dt, 0.0, 1.0)
        return x_next, x_den_out

    def step(self, x_i, t_i, t_iplus1, gamma_i, blind=False):
        t_i, t_iplus1, gamma_i = self._scalar(t_i), self._scalar(t_iplus1), self._scalar(gamma_i)
        if x_i.is_cuda and x_i.dim() == 2 and self.use_hip_update:
            return self._step_hip(x_i, t_i, t_iplus1, gamma_i, blind)
        x_hat, t_hat = self.stochastic_timestep(x_i, t_i, gamma_i)
        ode_integrand, x_den = self._guided_eval(x_hat, t_hat, blind)
        dt = t_iplus1 - t_hat
        if t_iplus1 != 0 and self.order == 2:
            x_prime = (x_hat + dt * ode_integrand).detach()
            ode_integrand_next, x_den = self._guided_eval(x_prime, t_iplus1, blind, rescale=False)
            x_iplus1 = x_hat + dt * (.5 * (ode_integrand + ode_integrand_next))
        else:
            x_iplus1 = x_hat + dt * ode_integrand
        return x_iplus1.detach_(), x_den.detach()

    def predict(self, shape, device, blind=False):
        t = self.create_schedule()                          # host-side schedule: no device sync inside the loop
        x = self.initialize_x(shape, device, t)
        tl, gl = t.tolist(), self.get_gamma(t).tolist()
        x_den = None
        for i in range(0, self.T, 1):
            self.step_counter = i
            x, x_den = self.step(x, tl[i], tl[i + 1], gl[i], blind)
        return x_den.detach()       # DPS returns the last denoised estimate, not x (reference :178)

    def predict_unconditional(self, *args, **kwargs):
        raise ValueError("DPS not made for unconditional sampling")

    def predict_conditional(self, y, operator, shape=None, blind=False, **kwargs):
        ps = self.args.tester.posterior_sampling
        self.operator = operator
        self.y = y
        self.rec_loss = get_loss(ps.rec_loss, operator=self.operator)
        self._hip_op = bool(blind and hasattr(operator, "hip_optimize"))
        self._hip_loss = False
        if self._hip_op:
            operator.hip_bind(y, ps)
        elif not blind and hasattr(operator, "hip_rec_loss") and y.is_cuda:
            self._hip_loss = bool(operator.hip_bind(y, ps))       # informed: FIR + STFT loss + adjoints in the HIP library
        elif blind:
            self.rec_loss_params = get_loss(ps.rec_loss_params, operator=self.operator)
            self.optimizer_operator = torch.optim.Adam(self.operator.params + self.operator.params_phases, lr=ps.blind_hp.lr_op,
                                                       weight_decay=ps.blind_hp.weight_decay, betas=(ps.blind_hp.beta1, ps.blind_hp.beta2))
            self.RIR_noise_regularization_loss = get_loss(ps.RIR_noise_regularization.loss, operator=self.operator)
        if shape is None:
            shape = y.shape
        return self.predict(tuple(shape), y.device, blind)
