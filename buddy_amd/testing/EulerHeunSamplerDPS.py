"""Euler-Heun diffusion-posterior-sampling solver -- same surface as reference
``testing/EulerHeunSamplerDPS.py:15-204`` (``predict_conditional(y, operator, shape, blind)``; attribute ``operator``).

Batched: ``y`` may be ``(B, L)``; every reduction the reference takes over "the tensor" (``y.std()``,
``torch.norm(rec_grads)``, ``x_den.std()``) is taken per utterance, the operator holds per-utterance parameters, Adam
is elementwise -- so row b of a batched run equals the reference's B=1 run on utterance b.
The likelihood loss and its gradient w.r.t. the Tweedie estimate come from the operator's library handle (one call), autograd then
continues through the hand-written HIP network VJP (``NCSNppTime`` is an autograd Function).  The operator update (``optimize_op``) is one
library call.  No torch-op operator / Adam path exists in the product (oracle/batched/sampler.py holds that form for host-logic tests)."""
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
        self.use_hip_update = True      # fused elementwise tail on the GPU (False: torch expressions, as on the CPU)

    def initialize_x(self, shape, device, schedule):
        wi = self.args.tester.posterior_sampling.warm_initialization
        if wi.mode == "none":
            return schedule[0] * self._randn(shape, device)
        if wi.mode == "reverb_scaled":
            return wi.scaling_factor * self.y.clone() / _row_std(self.y) + schedule[0] * self._randn(shape, device)
        if wi.mode == "wpe_scaled":
            from ..utils.wpe import wpe_dereverb    # nara_wpe restated in HIP (third-party, parity unpinned): buddy_wpe_dereverb
            x_pred = wpe_dereverb(self.y, taps=wi.wpe.taps, delay=wi.wpe.delay, iterations=wi.wpe.iterations)
            if x_pred.shape[-1] < self.y.shape[-1]:
                x_pred = torch.nn.functional.pad(x_pred, (0, self.y.shape[-1] - x_pred.shape[-1]))
            x_pred = wi.scaling_factor * x_pred / _row_std(x_pred)
            return x_pred + schedule[0] * self._randn(shape, device)
        raise NotImplementedError

    def get_likelihood_score(self, x_den, x, t):
        # fused HIP loss + analytic d/dx_den (buddy_blindop_rec_loss_grad / _fir_loss_grad); autograd continues into the network VJP
        rec = self.operator.hip_rec_loss(x_den)
        rec_grads = torch.autograd.grad(outputs=rec, inputs=x)[0]
        normguide = torch.linalg.vector_norm(rec_grads, dim=-1, keepdim=True) / (self.args.exp.audio_len ** 0.5)
        return self.zeta / (normguide + 1e-8) * rec_grads, rec

    def optimize_op(self, x_den, t):
        """reference :71-113: ``op_updates_per_step`` Adam iterations on the operator parameters (update_H, degradation, the two losses,
        backward, Adam step, projection) -- ONE library call (``buddy_blindop_optimize``: 24 kernels per iteration in a captured hipGraph)."""
        return self.operator.hip_optimize(x_den, t)

    def _guided_eval(self, x_in, t, blind, rescale=True):
        """one guided evaluation; ``rescale=False`` is the Heun corrector, which the reference leaves un-rescaled (:139-149)"""
        x_in.requires_grad = True
        x_den = self.get_Tweedie_estimate(x_in, t)
        if blind:
            self.optimize_op(x_den.detach(), t)      # the library copies it into the captured graph's input buffer: no clone here
        lh_score, _ = self.get_likelihood_score(x_den, x_in, t)
        x_in.detach_()
        csm = self.args.tester.posterior_sampling.constraint_speech_magnitude
        if csm.use and rescale:
            x_den = csm.speech_scaling / _row_std(x_den.detach()) * x_den
        score = self.Tweedie2score(x_den, x_in, t)
        return self.diff_params._ode_integrand(x_in, t, score) + lh_score, x_den

    def _eval_parts(self, x_in, t, blind):
        """network + operator part of one guided evaluation -> (likelihood gradient, its per-utterance scale or None, raw Tweedie estimate), detached.
        Fast path (round 6): the network's saved forward and input-VJP and the operator's loss gradient are called directly -- no autograd graph, no
        helper kernels -- and the guidance normaliser zeta / (||g|| / sqrt(audio_len) + 1e-8) (:66-69) stays a per-utterance scalar that the fused update
        applies (``buddy_row_scale`` + ``buddy_dps_update``), instead of five tensor expressions and a scaled copy of the gradient."""
        from . import _hipops
        net, op = self.model, self.operator
        if hasattr(net, "denoise_saved") and hasattr(op, "hip_rec_loss_grad") and not torch.is_tensor(t):
            sc = self.diff_params.scalars_on_device(t, x_in.shape[0], x_in.device)
            x_den = net.denoise_saved(x_in, sc)
            if blind:
                self.optimize_op(x_den, t)
            g = net.input_vjp(op.hip_rec_loss_grad(x_den))
            return g, _hipops.row_scale(g, 1, self.zeta, self.args.exp.audio_len ** 0.5), x_den
        x_in.requires_grad = True
        x_den = self.get_Tweedie_estimate(x_in, t)
        if blind:
            self.optimize_op(x_den.detach(), t)
        lh_score, _ = self.get_likelihood_score(x_den, x_in, t)
        x_in.detach_()
        return lh_score.detach(), None, x_den.detach()

    def _step_hip(self, x_i, t_i, t_iplus1, gamma_i, blind):
        """same arithmetic as step() with the elementwise tail fused into HIP kernels (buddy_perturb / buddy_row_scale / buddy_dps_update)."""
        from . import _hipops
        csm = self.args.tester.posterior_sampling.constraint_speech_magnitude
        x_hat, t_hat = self.stochastic_timestep(x_i, t_i, gamma_i)
        lh, lhs, x_den = self._eval_parts(x_hat, t_hat, blind)
        scale = _hipops.row_scale(x_den, 0, csm.speech_scaling) if csm.use else None
        dt = float(t_iplus1 - t_hat)
        if t_iplus1 != 0 and self.order == 2:
            x_prime, d1, _ = _hipops.dps_update(x_hat, x_den, lh, scale, x_hat, None, float(t_hat), dt, 0.0, 1.0, want_d=True, lh_scale=lhs)
            lh2, lhs2, x_den2 = self._eval_parts(x_prime, t_iplus1, blind)
            # the reference rescales x_den only in the first evaluation (:127-129); the corrector uses and returns the raw estimate (:139-149)
            x_next, _, x_den_out = _hipops.dps_update(x_prime, x_den2, lh2, None, x_hat, d1, float(t_iplus1), dt, 0.5, 0.5, lh_scale=lhs2)
        else:
            x_next, _, x_den_out = _hipops.dps_update(x_hat, x_den, lh, scale, x_hat, None, float(t_hat), dt, 0.0, 1.0, lh_scale=lhs)
        return x_next, x_den_out

    def step(self, x_i, t_i, t_iplus1, gamma_i, blind=False):
        t_i, t_iplus1, gamma_i = self._scalar(t_i), self._scalar(t_iplus1), self._scalar(gamma_i)
        if x_i.is_cuda and x_i.dim() == 2 and x_i.dtype == torch.float32 and self.use_hip_update:
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

    def bind(self, y, operator, blind):
        """What the reference's predict_conditional sets up before its loop (:181-204): the operator, the observation, the losses and -- blind --
        a fresh Adam state.  Here all of it lives in the operator's library handle (``hip_bind``): compressed STFT of y cached, loss weights
        validated (utils/losses.get_loss), Adam moments zeroed.  An operator without the HIP entry points, an unsupported loss or a tensor
        that is not on the GPU raises: there is no torch-op path."""
        ps = self.args.tester.posterior_sampling
        self.operator, self.y = operator, y
        self.rec_loss = get_loss(ps.rec_loss, operator=operator)           # validated specification (the value comes from the library)
        if not (hasattr(operator, "hip_bind") and hasattr(operator, "hip_rec_loss")):
            raise NotImplementedError(f"{type(operator).__name__} has no HIP likelihood path (hip_bind / hip_rec_loss)")
        if blind and not hasattr(operator, "hip_optimize"):
            raise NotImplementedError(f"{type(operator).__name__} cannot be optimised blindly (no hip_optimize)")
        if blind:
            self.rec_loss_params = get_loss(ps.rec_loss_params, operator=operator)
            self.RIR_noise_regularization_loss = get_loss(ps.RIR_noise_regularization.loss, operator=operator)
        if operator.hip_bind(y, ps) is False:
            raise NotImplementedError("this observation / loss configuration is outside what the HIP likelihood kernels are built for "
                                      "(2-D GPU tensor of >= 1024 samples, l2_comp_stft_summean @ 0.667, operator STFT 1024/512/128 hann)")

    def predict_conditional(self, y, operator, shape=None, blind=False, **kwargs):
        self.bind(y, operator, blind)
        if shape is None:
            shape = y.shape
        return self.predict(tuple(shape), y.device, blind)
