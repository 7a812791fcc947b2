"""WPE (weighted prediction error) dereverberation for the ``wpe_scaled`` warm start of the blind sampler
(reference ``testing/EulerHeunSamplerDPS.py:32-54``: ``nara_wpe.utils.stft/istft`` with size 512 / shift 128 and
``nara_wpe.wpe.wpe(Y, taps=50, delay=2, iterations=5, statistics_mode='full')``).

``nara_wpe`` is a third-party package that is neither in the reference's ``requirements.txt`` nor installable offline, so the library
restates its published algorithm (Nakatani et al. 2010; Drude et al. 2018, batch "full statistics" variant) and its STFT conventions
(periodic Blackman analysis window, "fading" zero padding of size-shift samples on both sides, bi-orthogonal synthesis window).  PARITY
UNPINNED against the package itself; checked on the GPU against the independent numpy oracle ``oracle/wpe_ref.py``.
The whole estimate is ONE library call (``buddy_wpe_dereverb``, ``csrc/wpe.hip``): hand-written complex128 STFT, the iterations (inverse
power, correlation matrix, Cholesky solve, prediction filter; one workgroup per (utterance, bin) row) and the overlap-add iSTFT.  There is
no CPU form in the product: a CPU tensor raises ``BuddyHipError`` (the torch restatement used by host-logic tests lives in ``oracle/batched``)."""
from __future__ import annotations

import torch

from .. import _lib


def wpe_hip(Y, taps=10, delay=3, iterations=3):
    """The WPE iterations alone for D = 1 through the hand-written kernel (``buddy_wpe``): Y (F, 1, T) complex128 on the GPU."""
    lib = _lib.require_gpu()
    F, D, T = Y.shape
    assert D == 1 and Y.is_cuda and Y.dtype == torch.complex128
    Yr = torch.view_as_real(Y.contiguous()).contiguous()                    # (F, 1, T, 2) float64
    Xr = torch.empty_like(Yr)
    scratch = torch.empty(F * T, dtype=torch.float64, device=Y.device)
    _lib.check(lib.buddy_wpe(_lib.ptr(Yr), _lib.ptr(Xr), _lib.ptr(scratch), F, T, int(taps), int(delay), int(iterations), _lib.stream_ptr()))
    return torch.view_as_complex(Xr)


def wpe_dereverb(y, taps=50, delay=2, iterations=5, size=512, shift=128):
    """y (B, L) float32 on the GPU -> (B, L) float32: stft -> per-utterance single-channel WPE -> istft (reference :36-51), one library call."""
    lib = _lib.require_gpu()
    if not y.is_cuda:
        raise _lib.BuddyHipError("wpe_dereverb: the warm start runs on the MI355X only (got a CPU tensor)")
    assert (size, shift) == (512, 128), "the HIP warm start is built for the reference's stft_options (size 512, shift 128)"
    B, L = y.shape
    yc = y.contiguous().float()
    out = torch.empty_like(yc)
    work = torch.empty(int(lib.buddy_wpe_workspace_bytes(B, L)) // 8, dtype=torch.float64, device=y.device)
    _lib.check(lib.buddy_wpe_dereverb(_lib.ptr(yc), _lib.ptr(out), _lib.ptr(work), B, L, int(taps), int(delay), int(iterations), _lib.stream_ptr()))
    return out
