"""Reverb DSP utilities (reference ``utils/reverb_utils.py``), device-resident.

``fast_apply_RIR`` keeps the reference signature but runs the hand-written time-domain FIR kernel
(``buddy_fir``): the same linear convolution the reference evaluates through a 2^17-point complex FFT, exact to
fp32 round-off, with its transpose as the autograd backward.  ``hilbert`` / ``minimum_phase_version`` follow the
reference formulas on torch's device FFT (used by the blind operator's filter projection only)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _lib


class _FirFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h):
        lib = _lib.require_gpu()
        x = x.contiguous().float()
        h = h.contiguous().float()
        B, L = x.shape
        M = h.shape[-1]
        stride = M if (h.dim() == 2 and h.shape[0] == B and B > 1) else 0
        y = torch.empty_like(x)
        _lib.check(lib.buddy_fir(_lib.ptr(x), _lib.ptr(h), stride, _lib.ptr(y), B, L, M, 0, _lib.stream_ptr()))
        ctx.save_for_backward(h)
        ctx.stride = stride
        return y

    @staticmethod
    def backward(ctx, g):
        h, = ctx.saved_tensors
        g = g.contiguous()
        B, L = g.shape
        gx = torch.empty_like(g)
        _lib.check(_lib.load().buddy_fir(_lib.ptr(g), _lib.ptr(h), ctx.stride, _lib.ptr(gx), B, L, h.shape[-1], 1, _lib.stream_ptr()))
        return gx, None


def fast_apply_RIR(y, filter, rm_delay=False, zero_pad=False):
    """reference reverb_utils.py:25-61.  y: (B,N); filter: (M,) shared or (B,M) one RIR per utterance (zero padded)."""
    if rm_delay:
        filter = filter[..., int(torch.argmax(filter)):]
    return _FirFn.apply(y, filter.to(y.device))


def hilbert(h):
    """reference reverb_utils.py:3-7: window [2]*ceil(N/2) ++ [0]*floor(N/2), DC/Nyquist not special-cased."""
    n = h.size(-1)
    window = 2 * torch.heaviside(torch.linspace(-1, 1, steps=n), values=torch.ones(1)).to(h.device)
    window = torch.flip(window, dims=(-1,))
    return torch.fft.ifft(window * torch.fft.fft(h))


def minimum_phase_version(h):
    """reference reverb_utils.py:9-23 (batched over leading dims)."""
    T = h.size(-1)
    h = F.pad(h, (0, T))
    H = torch.fft.fft(h)
    log_abs = torch.log(torch.abs(H) + 1e-8)
    phase = -torch.imag(hilbert(log_abs))
    e = torch.exp(1j * phase)
    out = torch.real(torch.fft.ifft(torch.abs(H).type(e.dtype) * e))
    return out[..., :-T]
