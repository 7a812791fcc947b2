"""Reverb DSP utilities (reference ``utils/reverb_utils.py``), device-resident.

``fast_apply_RIR`` keeps the reference signature but runs the hand-written time-domain FIR kernel
(``buddy_fir``): the same linear convolution the reference evaluates through a 2^17-point complex FFT, exact to
fp32 round-off, with its transpose as the autograd backward.  ``hilbert`` / ``minimum_phase_version`` (reference :3-23) run inside the
blind operator's library handle (25 856-point FFT kernels, ``BlindSubbandFiltering.minimum_phase``); their torch restatement is test
infrastructure (``oracle/batched/operators.py``)."""
from __future__ import annotations

import torch

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
