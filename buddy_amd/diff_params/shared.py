"""Diffusion parameterisation base (inference part of reference ``diff_params/shared.py:8-120``)."""
from __future__ import annotations

import torch


class SDE:
    def __init__(self, type, sde_hp):
        self.type = type
        self.sde_hp = sde_hp

    # subclasses: cskip / cout / cin / cnoise / _mean / _std / _ode_integrand / Tweedie2score / score2Tweedie

    def denoiser(self, xn, net, t, *args, **kwargs):
        """Whole denoising step = network + preconditioning (reference shared.py:98-120).

        xn: (B,1,L) like the reference, or (B,L).  t: 0-dim tensor / float (reference: one sigma for the batch) or a
        (B,) tensor (per-utterance sigma).  If ``net`` exposes ``denoise_fused`` (the MI355X NCSNppTime) the scalars
        are folded into the STFT / overlap-add kernels; otherwise the generic expression is evaluated."""
        B = xn.shape[0]
        if not torch.is_tensor(t):
            t = torch.full((1,), float(t), dtype=xn.dtype, device=xn.device)      # a fill kernel, not a host-to-device copy
        t = torch.as_tensor(t, dtype=xn.dtype, device=xn.device)               # fp32 in the product; the float64 arbiter runs keep their sigma in float64
        sigma_b = self._std(t).reshape(-1).expand(B) if t.numel() in (1, B) else None
        if sigma_b is None:
            raise ValueError("t must be a scalar or have one entry per batch element")
        cskip, cout, cin, cnoise = self.cskip(sigma_b), self.cout(sigma_b), self.cin(sigma_b), self.cnoise(sigma_b)
        if hasattr(net, "denoise_fused"):
            x2 = xn[:, 0] if xn.dim() == 3 else xn
            y = net.denoise_fused(x2, cnoise, cin, cskip, cout)
            return y[:, None] if xn.dim() == 3 else y
        shape = (B,) + (1,) * (xn.dim() - 1)
        return cskip.view(shape) * xn + cout.view(shape) * net(cin.view(shape) * xn, cnoise)
