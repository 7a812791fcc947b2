"""Diffusion parameterisation base (inference part of reference ``diff_params/shared.py:8-120``)."""
from __future__ import annotations

import numpy as np
import torch


class SDE:
    def __init__(self, type, sde_hp):
        self.type = type
        self.sde_hp = sde_hp

    # subclasses: cskip / cout / cin / cnoise / _mean / _std / _ode_integrand / Tweedie2score / score2Tweedie

    def host_scalars(self, t):
        """(cnoise, cin, cskip, cout) of ONE sigma as fp32 host numbers: the subclass's formulas evaluated on a numpy float32 scalar (every operation
        rounds to fp32 like the tensor expressions), so the device only has to broadcast them (``buddy_fill_rows4``)."""
        s = np.float32(float(self._std(t)))
        return tuple(float(np.float32(f(s))) for f in (self.cnoise, self.cin, self.cskip, self.cout))

    def scalars_on_device(self, t, B, device):
        """(4, B) fp32 rows (cnoise, cin, cskip, cout) for a host-side sigma: one tiny launch instead of a dozen tensor expressions per evaluation"""
        from .. import _lib
        out = torch.empty(4, B, dtype=torch.float32, device=device)
        v = self.host_scalars(t)
        _lib.check(_lib.require_gpu().buddy_fill_rows4(_lib.ptr(out), B, v[0], v[1], v[2], v[3], _lib.stream_ptr()))
        return out

    def denoiser(self, xn, net, t, *args, **kwargs):
        """Whole denoising step = network + preconditioning (reference shared.py:98-120).

        xn: (B,1,L) like the reference, or (B,L).  t: 0-dim tensor / float (reference: one sigma for the batch) or a
        (B,) tensor (per-utterance sigma).  If ``net`` exposes ``denoise_fused`` (the MI355X NCSNppTime) the scalars
        are folded into the STFT / overlap-add kernels; otherwise the generic expression is evaluated."""
        B = xn.shape[0]
        if not torch.is_tensor(t) and hasattr(net, "denoise_fused") and xn.is_cuda and xn.dtype == torch.float32:
            sc = self.scalars_on_device(t, B, xn.device)
            x2 = xn[:, 0] if xn.dim() == 3 else xn
            y = net.denoise_fused(x2, sc[0], sc[1], sc[2], sc[3])
            return y[:, None] if xn.dim() == 3 else y
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
