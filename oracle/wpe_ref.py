"""Oracle: the WPE warm start of the blind sampler (``wpe_scaled``), numpy complex128 on the CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Independent of ``buddy_amd/utils/wpe.py`` and ``buddy_amd/csrc/wpe.hip`` (imports
neither); written against the third-party package the reference calls, not against the product.

What the reference does (``testing/EulerHeunSamplerDPS.py:32-54``)::

    Y = nara_wpe.utils.stft(y.cpu().numpy(), size=512, shift=128)          # (1, T, 257)
    Z = nara_wpe.wpe.wpe(Y.transpose(2, 0, 1), taps=50, delay=2, iterations=5, statistics_mode='full').transpose(1, 2, 0)
    x_pred = nara_wpe.utils.istft(Z, size=512, shift=128)[..., :L]
    x = 0.05 * x_pred / x_pred.std() + t_0 * randn

``nara_wpe`` (fgnt/nara_wpe, MIT) is NOT in ``/root/reference``, not in its ``requirements.txt`` and not installed here, so this is a
restatement of the package's published algorithm and conventions from its documentation / the papers, function by function:

* ``nara_wpe.utils.stft`` (defaults ``window=scipy.signal.windows.blackman``, ``fading=True``, ``pad=True``, ``symmetric_window=False``):
  zero-pad ``size - shift`` samples on both sides ("fading"), cut into frames of ``size`` every ``shift`` with the tail zero-padded to a
  whole frame (``segment_axis(..., end='pad')``), periodic window ``blackman(size + 1)[:-1]``, ``numpy.fft.rfft``.
* ``nara_wpe.utils.istft``: synthesis window = the analysis window divided by the sum of its squares over the ``size / shift`` overlapping
  positions (``_biorthogonal_window``), overlap-add of ``irfft`` frames, fading samples removed.
* ``nara_wpe.wpe.wpe`` (= the per-frequency loop ``wpe_v8`` over ``wpe_v6``; T. Nakatani et al., "Speech dereverberation based on
  variance-normalized delayed linear prediction", IEEE TASLP 18(7), 2010, eqs. (12)-(15) / Drude et al., "NARA-WPE", ITG 2018,
  Algorithm 1, batch form).  Per frequency bin, Y: (D, T), here D = 1::

      Y~_t   = [y_{t-delay-taps+1}; ...; y_{t-delay}]                      (``build_y_tilde``: OLDEST frame first, zeros before t = 0)
      X <- Y
      repeat ``iterations``:
          lambda_t = max(mean_d |x_{d,t}|^2, 1e-10 * max_t(.))             (``get_power_inverse``, psd_context = 0)
          R = sum_t Y~_t Y~_t^H / lambda_t,   P = sum_t Y~_t y_t^H / lambda_t     (statistics_mode='full': all frames)
          G = R^{-1} P                                                     (``_stable_solve`` -> numpy.linalg.solve)
          X = Y - G^H Y~

PARITY UNPINNED against nara_wpe itself (package absent: no output of it exists to compare with); what this file pins is the PRODUCT
(HIP kernels, ``buddy_wpe_dereverb``) against an independently written statement of the same published algorithm.
"""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------- nara_wpe.utils
def _window(size):
    """scipy.signal.windows.blackman(size + 1)[:-1]: the periodic Blackman window (symmetric_window=False)"""
    n = np.arange(size + 1, dtype=np.float64)
    w = 0.42 - 0.5 * np.cos(2.0 * np.pi * n / size) + 0.08 * np.cos(4.0 * np.pi * n / size)
    return w[:-1]


def _segment_pad(x, length, shift):
    """segment_axis(x, length, shift, axis=-1, end='pad'): (..., n) -> (..., frames, length), the tail zero-padded to a whole frame"""
    n = x.shape[-1]
    if n < length:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (length - n,), x.dtype)], -1)
    elif (n + shift - length) % shift != 0:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (shift - (n + shift - length) % shift,), x.dtype)], -1)
    frames = (x.shape[-1] - length) // shift + 1
    idx = np.arange(length)[None, :] + shift * np.arange(frames)[:, None]
    return x[..., idx]


def stft(time_signal, size=512, shift=128):
    """(..., samples) -> (..., frames, size // 2 + 1) complex128"""
    x = np.asarray(time_signal)
    pad = np.zeros(x.shape[:-1] + (size - shift,), x.dtype)
    x = np.concatenate([pad, x, pad], -1)                       # fading=True
    seg = _segment_pad(x, size, shift)
    return np.fft.rfft(seg * _window(size), n=size, axis=-1)


def _biorthogonal_window(analysis_window, shift):
    size = len(analysis_window)
    assert size % shift == 0
    k = size // shift
    s = np.sum(analysis_window.reshape(k, shift) ** 2, axis=0)
    return analysis_window / np.tile(s, k)


def istft(stft_signal, size=512, shift=128):
    """(..., frames, size // 2 + 1) -> (..., frames * shift - (size - shift)) float64"""
    assert stft_signal.shape[-1] == size // 2 + 1
    w = _biorthogonal_window(_window(size), shift)
    frames = stft_signal.shape[-2]
    out = np.zeros(stft_signal.shape[:-2] + (frames * shift + size - shift,), np.float64)
    seg = w * np.real(np.fft.irfft(stft_signal, n=size, axis=-1))
    for t in range(frames):
        out[..., t * shift:t * shift + size] += seg[..., t, :]
    return out[..., size - shift:out.shape[-1] - (size - shift)]


# --------------------------------------------------------------------------- nara_wpe.wpe
def build_y_tilde(Y, taps, delay):
    """(D, T) -> (taps * D, T): block k holds Y delayed by delay + taps - 1 - k frames (the oldest frame first), zero before the start"""
    D, T = Y.shape
    Yt = np.zeros((taps * D, T), Y.dtype)
    for k in range(taps):
        s = delay + taps - 1 - k
        if s < T:
            Yt[k * D:(k + 1) * D, s:] = Y[:, :T - s]
    return Yt


def get_power_inverse(signal):
    """(D, T) -> (T,): 1 / max(mean_d |x|^2, 1e-10 max_t)   (psd_context = 0)"""
    power = np.mean(signal.real ** 2 + signal.imag ** 2, axis=-2)
    eps = 1e-10 * np.max(power)
    return 1.0 / np.maximum(power, eps)


def wpe_one_frequency(Y, taps, delay, iterations):
    """(D, T) -> (D, T), statistics_mode='full'"""
    X = np.copy(Y)
    Yt = build_y_tilde(Y, taps, delay)
    for _ in range(iterations):
        inv = get_power_inverse(X)
        Yti = Yt * inv[None, :]
        R = Yti @ Yt.conj().T
        P = Yti @ Y.conj().T
        G = np.linalg.solve(R, P)
        X = Y - G.conj().T @ Yt
    return X


def wpe(Y, taps=10, delay=3, iterations=3):
    """(F, D, T) -> (F, D, T): independent problems per frequency bin"""
    out = np.empty_like(Y)
    for f in range(Y.shape[0]):
        out[f] = wpe_one_frequency(Y[f], taps, delay, iterations)
    return out


# --------------------------------------------------------------------------- the call site, EulerHeunSamplerDPS.py:32-54
def wpe_warm_start_estimate(y, taps=50, delay=2, iterations=5, size=512, shift=128):
    """y (1, L) float32 numpy (ONE utterance: the reference samples with batch size 1, its ``.std()`` is that utterance's) ->
    x_pred (1, L) float64 BEFORE the rescaling to ``scaling_factor`` (:51)."""
    y = np.asarray(y)
    assert y.ndim == 2 and y.shape[0] == 1
    Y = stft(y, size=size, shift=shift).transpose(2, 0, 1)              # (F, 1, T)
    Z = wpe(Y, taps=taps, delay=delay, iterations=iterations).transpose(1, 2, 0)
    x = istft(Z, size=size, shift=shift)
    if x.shape[-1] > y.shape[-1]:
        x = x[..., :y.shape[-1]]
    return x
