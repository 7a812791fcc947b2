"""Thin wrappers over the fused sampler kernels of ``libbuddy_hip.so`` (per-utterance rows, device tensors, no autograd)."""
from __future__ import annotations

import torch

from .. import _lib


def perturb(x, eps, scale):
    """x + scale * eps (reference EulerHeunSampler.py:41-45)."""
    x, eps = x.contiguous(), eps.contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.load().buddy_perturb(_lib.ptr(x), _lib.ptr(eps), float(scale), _lib.ptr(out), x.numel(), _lib.stream_ptr()))
    return out


def row_std(x):
    """unbiased per-row standard deviation (Tensor.std() per utterance), accumulated in fp64 on the device -> (B,1) fp32."""
    x = x.contiguous()
    B, L = x.shape
    mom = torch.empty(B, 2, dtype=torch.float64, device=x.device)
    _lib.check(_lib.load().buddy_row_moments(_lib.ptr(x), _lib.ptr(mom), B, L, _lib.stream_ptr()))
    var = (mom[:, 1] - mom[:, 0] ** 2 / L) / (L - 1)
    return var.clamp_min(0).sqrt().to(torch.float32).unsqueeze(1)


def row_scale(x, mode, p0, p1=1.0):
    """(B,) fp32, one launch: mode 0 -> p0 / std(x_b) (unbiased, Tensor.std() per utterance); mode 1 -> p0 / (||x_b||_2 / p1 + 1e-8)."""
    x = x.contiguous()
    B, L = x.shape
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().buddy_row_scale(_lib.ptr(x), _lib.ptr(out), B, L, int(mode), float(p0), float(p1), _lib.stream_ptr()))
    return out


def dps_update(x_hat, x_den, lh, den_scale, base, d_prev, t, dt, w_prev, w_cur, want_d=False, lh_scale=None):
    """fused score -> ODE integrand (+ lh_scale[b] * likelihood gradient) -> Euler/Heun update; returns (x_next, d or None, x_den')."""
    B, L = x_hat.shape
    c = lambda v: None if v is None else v.contiguous()
    x_hat, x_den, lh, den_scale, base, d_prev, lh_scale = c(x_hat), c(x_den), c(lh), c(den_scale), c(base), c(d_prev), c(lh_scale)
    out = torch.empty_like(x_hat)
    d_out = torch.empty_like(x_hat) if want_d else None
    xd_out = torch.empty_like(x_hat)
    _lib.check(_lib.load().buddy_dps_update(_lib.ptr(x_hat), _lib.ptr(x_den), _lib.ptr(lh), _lib.ptr(lh_scale), _lib.ptr(den_scale), _lib.ptr(base), _lib.ptr(d_prev),
                                            float(t), float(dt), float(w_prev), float(w_cur), _lib.ptr(out), _lib.ptr(d_out), _lib.ptr(xd_out),
                                            B, L, _lib.stream_ptr()))
    return out, d_out, xd_out
