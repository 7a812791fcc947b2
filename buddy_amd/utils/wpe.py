"""WPE (weighted prediction error) dereverberation for the ``wpe_scaled`` warm start of the blind sampler
(reference ``testing/EulerHeunSamplerDPS.py:32-54``: ``nara_wpe.utils.stft/istft`` with size 512 / shift 128 and
``nara_wpe.wpe.wpe(Y, taps=50, delay=2, iterations=5, statistics_mode='full')``).

``nara_wpe`` is a third-party package that is neither in the reference's ``requirements.txt`` nor installable offline, so this
is a restatement of its published algorithm (Nakatani et al. 2010; Drude et al. 2018, batch "full statistics" variant) and of
its STFT conventions (periodic Blackman analysis window, "fading" zero padding of size-shift samples on both sides,
bi-orthogonal synthesis window).  PARITY UNPINNED: there is no reference output to compare with.
Runs once per utterance, outside the sampling loop.  On the GPU the whole estimate is ONE library call (``buddy_wpe_dereverb``,
``csrc/wpe.hip``): hand-written complex128 STFT, the iterations (inverse power, correlation matrix, Cholesky solve, prediction filter;
one workgroup per (utterance, bin) row) and the overlap-add iSTFT.  Checked on the GPU against ``oracle/wpe_ref.py`` (numpy, written
independently).  The torch functions below are the ``backend="torch"`` form for CPU host-logic tests and never run on a CUDA tensor."""
from __future__ import annotations

import math

import torch


def _blackman_periodic(n, device):
    k = torch.arange(n, dtype=torch.float64, device=device)
    return 0.42 - 0.5 * torch.cos(2 * math.pi * k / n) + 0.08 * torch.cos(4 * math.pi * k / n)


def stft(x, size=512, shift=128):
    """x (..., samples) -> (..., frames, size//2+1); fading + end padding like nara_wpe.utils.stft."""
    x = x.to(torch.float64)
    x = torch.nn.functional.pad(x, (size - shift, size - shift))
    n = x.shape[-1]
    if n < size or (n - size) % shift:
        x = torch.nn.functional.pad(x, (0, (shift - (n - size) % shift) % shift if n >= size else size - n))
    frames = x.unfold(-1, size, shift)
    return torch.fft.rfft(frames * _blackman_periodic(size, x.device), n=size, dim=-1)


def _biorthogonal(window, shift):
    size = window.shape[0]
    den = torch.zeros_like(window)
    for k in range(-(size // shift) + 1, size // shift):
        lo, hi = max(0, k * shift), min(size, size + k * shift)
        if lo < hi:
            den[lo:hi] += window[lo - k * shift:hi - k * shift] ** 2
    return window / den


def istft(X, size=512, shift=128):
    """(..., frames, size//2+1) -> (..., samples), overlap-add with the bi-orthogonal window, fading removed."""
    w = _biorthogonal(_blackman_periodic(size, X.device), shift)
    seg = torch.fft.irfft(X, n=size, dim=-1) * w
    T = X.shape[-2]
    out = torch.zeros(X.shape[:-2] + (T * shift + size - shift,), dtype=seg.dtype, device=X.device)
    for j in range(T):
        out[..., j * shift:j * shift + size] += seg[..., j, :]
    return out[..., size - shift:out.shape[-1] - (size - shift)]


def wpe(Y, taps=10, delay=3, iterations=3):
    """Y (F, D, T) complex -> dereverberated (F, D, T); statistics_mode='full', psd_context=0."""
    F, D, T = Y.shape
    Yt = torch.zeros(F, taps * D, T, dtype=Y.dtype, device=Y.device)
    for tau in range(taps):
        s = delay + tau
        if s < T:
            Yt[:, tau * D:(tau + 1) * D, s:] = Y[:, :, :T - s]
    X = Y
    for _ in range(iterations):
        power = (X.real ** 2 + X.imag ** 2).mean(dim=1)                        # (F, T)
        inv = 1.0 / torch.maximum(power, 1e-10 * power.amax(dim=-1, keepdim=True))
        Yti = Yt * inv[:, None, :]
        R = Yti @ Yt.conj().transpose(1, 2)
        P = Yti @ Y.conj().transpose(1, 2)
        G = torch.linalg.solve(R, P)
        X = Y - G.conj().transpose(1, 2) @ Yt
    return X


def wpe_hip(Y, taps=10, delay=3, iterations=3):
    """Same as :func:`wpe` for D = 1 on a CUDA tensor through the hand-written kernel (``buddy_wpe``): Y (F, 1, T) complex128."""
    from .. import _lib
    lib = _lib.require_gpu()
    F, D, T = Y.shape
    assert D == 1 and Y.is_cuda and Y.dtype == torch.complex128
    Yr = torch.view_as_real(Y.contiguous()).contiguous()                    # (F, 1, T, 2) float64
    Xr = torch.empty_like(Yr)
    scratch = torch.empty(F * T, dtype=torch.float64, device=Y.device)
    _lib.check(lib.buddy_wpe(_lib.ptr(Yr), _lib.ptr(Xr), _lib.ptr(scratch), F, T, int(taps), int(delay), int(iterations), _lib.stream_ptr()))
    return torch.view_as_complex(Xr)


def wpe_dereverb(y, taps=50, delay=2, iterations=5, size=512, shift=128):
    """y (B, L) float -> (B, L) float32: stft -> per-utterance single-channel WPE -> istft (reference :36-51).
    A CUDA tensor goes through the HIP library (one call, no torch ops, no fallback); the torch form serves CPU tensors."""
    if y.is_cuda:
        from .. import _lib
        lib = _lib.require_gpu()
        assert (size, shift) == (512, 128), "the HIP warm start is built for the reference's stft_options (size 512, shift 128)"
        B, L = y.shape
        yc = y.contiguous().float()
        out = torch.empty_like(yc)
        work = torch.empty(int(lib.buddy_wpe_workspace_bytes(B, L)) // 8, dtype=torch.float64, device=y.device)
        _lib.check(lib.buddy_wpe_dereverb(_lib.ptr(yc), _lib.ptr(out), _lib.ptr(work), B, L, int(taps), int(delay), int(iterations), _lib.stream_ptr()))
        return out
    out = []
    for b in range(y.shape[0]):
        Y = stft(y[b:b + 1], size, shift)                   # (1, T, F)
        Z = wpe(Y.permute(2, 0, 1).contiguous(), taps=taps, delay=delay, iterations=iterations).permute(1, 2, 0)
        out.append(istft(Z, size, shift))
    x = torch.cat(out, dim=0).to(torch.float32)
    return x[..., :y.shape[-1]]
