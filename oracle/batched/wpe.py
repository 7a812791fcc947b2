"""TEST INFRASTRUCTURE, not product: torch (float64, CPU) restatement of nara_wpe's published STFT conventions and of the WPE iterations
(Nakatani et al. 2010; Drude et al. 2018, "full statistics"), written separately from ``oracle/wpe_ref.py`` (numpy).  Two restatements of one
published algorithm agreeing to 1e-6 is what ``tests/test_host_logic.py`` checks; the product warm start is ``buddy_wpe_dereverb``
(``buddy_amd/csrc/wpe.hip``), checked on the GPU against ``oracle/wpe_ref.py``.  Reference call site: ``testing/EulerHeunSamplerDPS.py:32-54``."""
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


def wpe_dereverb(y, taps=50, delay=2, iterations=5, size=512, shift=128):
    """y (B, L) float -> (B, L) float32: stft -> per-utterance single-channel WPE -> istft (reference :36-51)."""
    out = []
    for b in range(y.shape[0]):
        Y = stft(y[b:b + 1], size, shift)                   # (1, T, F)
        Z = wpe(Y.permute(2, 0, 1).contiguous(), taps=taps, delay=delay, iterations=iterations).permute(1, 2, 0)
        out.append(istft(Z, size, shift))
    x = torch.cat(out, dim=0).to(torch.float32)
    return x[..., :y.shape[-1]]
