"""Subband (STFT-domain) reverberation operators, same surface as reference
``testing/operators/subband_filtering.py`` (``SubbandFiltering`` :8-136, ``BlindSubbandFiltering`` :142-351), batched
per utterance: every parameter / filter tensor carries a leading utterance axis ``U`` (U = 1 reproduces the
reference exactly; U > 1 is the per-utterance vmap of it -- no cross-utterance coupling anywhere).

Two implementations of ``BlindSubbandFiltering`` behind one constructor:

* ``BlindSubbandFilteringHIP`` -- the product path on the GPU: parameters, Adam state and every intermediate live inside a
  ``buddy_blindop_*`` handle of ``libbuddy_hip.so`` (hand-written forward + analytic backward kernels, one library call per
  ``optimize_op``); selected whenever the signal ``length`` is given and the device is a GPU (what ``Tester.prepare_batch`` does).
* the torch-op class body below (rocFFT / grouped conv1d / autograd): host-logic tests on CPU and an on-GPU cross-check
  (``backend="torch"``); it is not what the bench or the sampler run.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from ... import _lib
from ...utils import reverb_utils
from .shared import Operator
from ._stft import OperatorSTFT


def linear_interp(knots, values, query):
    """Piecewise-linear interpolation over ``knots`` evaluated at ``query`` -- what the reference obtains from
    ``torchcde.LinearInterpolation(torchcde.linear_interpolation_coeffs(v), t=knots).evaluate(query)``
    (subband_filtering.py:233-235; third-party, restated from API semantics).  values (..., K), returns (..., Q)."""
    K = knots.shape[0]
    idx = (torch.bucketize(query, knots) - 1).clamp(0, K - 2)
    t0, t1 = knots[idx], knots[idx + 1]
    frac = (query - t0) / (t1 - t0)
    v0, v1 = values[..., idx], values[..., idx + 1]
    return v0 + frac * (v1 - v0)


class SubbandFiltering(Operator, OperatorSTFT):
    def __init__(self, op_hp, sample_rate, device=None):
        super().__init__()
        self.H = None
        self.op_hp = op_hp
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self._init_stft(op_hp, sample_rate, dev)
        assert self.n_fft >= self.win_length, "n_fft must be greater than 2*win_length to avoid temporal aliasing"
        self.Nf = self.op_hp.Nf
        self.length_rir = self.hop_length * self.Nf
        self.time = torch.arange(self.Nf, dtype=torch.float32) / (self.sample_rate / self.hop_length)

    def subband_filtering(self, X, H):
        """Per-band causal FIR along frames (reference :67-74).  X (U,F,T) complex, H (F,Nf) or (U,F,Nf)."""
        pre = int((self.win_length // self.hop_length) / 2) - 1
        U, Fb, T = X.shape
        if H.dim() == 2:
            H = H.unsqueeze(0).expand(U, -1, -1)
        Hf = torch.flip(H, dims=[-1]).reshape(U * Fb, 1, -1)
        Xp = F.pad(X, (Hf.shape[-1] - 1 - pre, pre)).reshape(1, U * Fb, -1)
        return F.conv1d(Xp, Hf, groups=U * Fb).reshape(U, Fb, -1)

    def degradation(self, x, mode="waveform", H=None, detach_operator=False):
        init_shape = x.shape
        X = self.apply_stft(x)
        if H is None:
            assert self.H is not None, "filter is not initialized"
            H = self.H
        if detach_operator:
            H = H.detach()
        Y = self.subband_filtering(X, H)
        if mode == "waveform":
            y = self.apply_istft(Y, length=init_shape[-1])
            return y.squeeze(0) if len(init_shape) == 1 else y
        elif mode == "STFT":
            return Y

    def get_time_RIR(self, excitation=None, H=None):
        """(U, length_rir+1024) estimated time-domain RIR(s) (reference :103-113; U=1 squeezes like the reference)."""
        if excitation is None:
            x = torch.zeros(int(self.length_rir + 1024), dtype=torch.float32, device=self.device)
            x[0] = 1
        else:
            x = torch.as_tensor(excitation, dtype=torch.float32, device=self.device)
        Hh = self.H if H is None else H
        U = Hh.shape[0] if Hh.dim() == 3 else 1
        r = self.degradation(x.unsqueeze(0).expand(U, -1), H=Hh)
        return r.squeeze(0) if U == 1 else r

    def update_H(self, rir=None, H=None):
        if rir is not None:
            H = self.stft(rir)
            H = H * (8) / (self.win_length / (self.hop_length))
            H = H[..., 1:]
            if self.op_hp.Nf > H.shape[-1]:
                H = torch.cat((H, torch.zeros(H.shape[:-1] + (self.op_hp.Nf - H.shape[-1],), device=H.device)), -1)
            else:
                H = H[..., 0:self.op_hp.Nf]
            self.H = H
        elif H is not None:
            self.H = H
        else:
            raise ValueError("Either rir or H must be specified. This is the informed scenario, so we need to know the filter")
        assert self.H.shape[-2] == self.n_fft // 2 + 1 and self.H.shape[-1] == self.Nf


class _HipRecLoss(torch.autograd.Function):
    """sum_u weight * l2_comp_stft_summean(y_u, degrade(x_den_u)) with the analytic gradient from the HIP operator."""

    @staticmethod
    def forward(ctx, x_den, op, weight):
        lib = _lib.require_gpu()
        x = x_den.contiguous().float()
        loss = torch.empty(op.U, device=x.device)
        g = torch.empty_like(x)
        _lib.check(lib.buddy_blindop_rec_loss_grad(op._h, _lib.ptr(x), float(weight), _lib.ptr(loss), _lib.ptr(g), _lib.stream_ptr()))
        ctx.save_for_backward(g)
        op.last_rec_per_utt = loss
        return loss.sum()

    @staticmethod
    def backward(ctx, gout):
        g, = ctx.saved_tensors
        return gout * g, None, None


class BlindSubbandFiltering(SubbandFiltering):
    def __new__(cls, op_hp, sample_rate, *a, backend=None, length=None, device=None, **k):
        if cls is BlindSubbandFiltering:
            dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
            use_hip = (backend == "hip") or (backend is None and length is not None and str(dev).startswith("cuda"))
            if use_hip:
                return super().__new__(BlindSubbandFilteringHIP)
        return super().__new__(cls)

    def __init__(self, op_hp, sample_rate, magnitude_distance=True, H_cplx=False, num_utts=1, noise=None, device=None, backend=None,
                 length=None):
        """``num_utts``: utterances handled by this operator object (per-utterance parameters).  ``noise``: optional list
        of per-utterance noise sources with ``rand(shape)`` / ``randn(shape)`` (parity runs); default torch RNG.
        ``length`` (signal length in samples) + a CUDA device selects the hand-written HIP backend
        (``BlindSubbandFilteringHIP``); ``backend="torch"`` forces this torch-op implementation."""
        super().__init__(op_hp, sample_rate, device=device)
        self.U = int(num_utts)
        self.noise = noise
        self.Amin, self.Amax = self.op_hp.Amin, self.op_hp.Amax
        self.EQ_freqs = torch.tensor([float(f) for f in self.op_hp.EQ_freqs], device=self.device)
        self.fix_EQ_extremes = self.op_hp.fix_EQ_extremes
        self.num_bands = len(self.EQ_freqs) - 2 if self.fix_EQ_extremes else len(self.EQ_freqs)
        if self.op_hp.init_single_value:
            t60 = [self.num_bands * [float(t)] for t in op_hp.init_params.T60_breakpoints]
            wts = [self.num_bands * [float(w)] for w in op_hp.init_params.multiexp_weighting]
        else:
            t60, wts = op_hp.init_params.T60_breakpoints, op_hp.init_params.multiexp_weighting
        t60 = torch.tensor(t60, dtype=torch.float32, device=self.device)
        wts = torch.tensor(wts, dtype=torch.float32, device=self.device)
        frame_rate = self.sample_rate / op_hp.hop
        decay = 6.908 / (t60 * frame_rate)
        self.num_exponentials = decay.shape[0]
        assert len(wts) == self.num_exponentials, "multiexp_weighting must have the same length as T60_breakpoints"
        assert t60.shape[-1] == self.num_bands and wts.shape[1] == self.num_bands
        # parameters: (U, E, bands), (U, E, bands), (U, F, Nf)
        self.params_decay = torch.nn.Parameter(decay.unsqueeze(0).repeat(self.U, 1, 1))
        self.params_decay_weighting = torch.nn.Parameter(wts.unsqueeze(0).repeat(self.U, 1, 1))
        self.max_decay = 6.908 / (op_hp.T60min * frame_rate)
        self.min_decay = 6.908 / (op_hp.T60max * frame_rate)
        with torch.no_grad():
            ph = self._rand((self.n_fft // 2 + 1, self.Nf)) * 2 * np.pi - np.pi
        self.phases = torch.nn.Parameter(ph, requires_grad=True)
        self.params = [self.params_decay, self.params_decay_weighting]
        self.params_phases = [self.phases]
        self.fix_direct_path = self.op_hp.fix_direct_path
        self.compute_direct_path_mag_correction()
        if self.op_hp.init_phases == "random_coherent":
            self.update_H(use_noise=True)
        elif self.op_hp.init_phases == "random":
            self.update_H()
        else:
            raise NotImplementedError("This is not implemented yet")

    # -- noise plumbing (reference draws with torch.rand / torch.randn on the fly) --------------------
    def _rand(self, shape):
        if self.noise is None:
            return torch.rand((self.U,) + tuple(shape)).to(self.device)
        return torch.stack([n.rand(shape) for n in self.noise]).to(self.device)

    def _randn(self, shape):
        if self.noise is None:
            return torch.randn((self.U,) + tuple(shape)).to(self.device)
        return torch.stack([n.randn(shape) for n in self.noise]).to(self.device)

    def compute_direct_path_mag_correction(self):
        h = torch.zeros((self.length_rir,), device=self.device)
        h[0] = 1 * (self.win_length / (self.hop_length * 2))
        self.direct_path_mag_correction = self.stft(h)[:, 1:].abs()

    def correct_OLA(self, A, inverse=False):
        K = int(self.win_length / (self.hop_length) - 1)
        win_sum = torch.sum(self.window)
        corr = torch.ones(A.shape[-1], device=A.device)
        for k in range(0, K):
            corr[k] = win_sum / torch.sum(self.window[int((K - k) * self.hop_length):])
        return A * corr if inverse else A / corr

    def design_subband_filter(self):
        """reference :224-239 -> (U, F, Nf) magnitudes."""
        Nf = len(self.time)
        decay_bp = torch.exp(self.params[0])                                   # (U,E,bands)
        weights = self.params[1]
        n = torch.arange(0, Nf, device=self.device).float()
        inner = (weights.unsqueeze(-1) * decay_bp.unsqueeze(-1) ** (-n)).sum(1)  # (U,bands,Nf)
        if self.fix_EQ_extremes:
            z = torch.zeros(inner.shape[0], 1, Nf, device=self.device)
            dm = torch.cat([z, inner, z], dim=1)                                   # rows 0 and -1 stay zero
        else:
            dm = inner
        dm = torch.log(dm.transpose(1, 2) + 1e-6)                                # (U,Nf,knots)
        H2 = linear_interp(self.EQ_freqs.to(torch.float32), dm, self.freqs)       # (U,Nf,F)
        H2 = torch.exp(H2.transpose(1, 2))
        assert not torch.isnan(H2).any(), "decay is Nan"
        return H2

    def design_filter(self, correct_OLA=True):
        A = self.design_subband_filter() + 1e-6
        if correct_OLA:
            A = self.correct_OLA(A)
        if self.fix_direct_path:
            A = A + self.direct_path_mag_correction
        assert A.shape[-2] == self.n_fft // 2 + 1 and A.shape[-1] == self.op_hp.Nf
        return A

    def get_noise(self, noise=None):
        if noise is None:
            noise = self._randn((self.length_rir,))
        N = self.stft(noise) / self._norm
        return N[..., 1:]

    def update_H(self, rir=None, H=None, use_noise=False, noise=None, phases=None):
        if rir is not None:
            super().update_H(rir=rir)
        elif H is not None:
            super().update_H(H=H)
        else:
            A = self.design_filter()
            if use_noise:
                N = self.get_noise(noise)
                self.H = self.cons(A * torch.exp(1j * N.angle()), length=self.length_rir)
                self.params_phases[0] = torch.angle(self.H).detach()
            elif phases is not None:
                self.params_phases[0] = phases
                self.H = self.cons(A * torch.exp(1j * phases), length=self.length_rir)
            else:
                self.H = self.cons(A * torch.exp(1j * self.params_phases[0]), length=self.length_rir)
        assert self.H.shape[-2] == self.n_fft // 2 + 1 and self.H.shape[-1] == self.Nf

    def update_params(self, params_dict):
        T60s = torch.tensor(params_dict.T60_breakpoints, dtype=torch.float32, device=self.device)
        w = torch.tensor(params_dict.multiexp_weighting, dtype=torch.float32, device=self.device)
        decays = 6.908 / (T60s * (self.sample_rate / self.hop_length))
        assert len(w) == len(T60s)
        self.num_exponentials = len(T60s)
        self.params[0] = torch.nn.Parameter(decays.unsqueeze(0).repeat(self.U, 1, 1), requires_grad=True)
        self.params[1] = torch.nn.Parameter(w.unsqueeze(0).repeat(self.U, 1, 1), requires_grad=True)

    def project_params(self):
        """reference :298-331, vectorised over utterances and bands (same clamps, same order over exponentials)."""
        for i in range(len(self.params)):
            self.params[i].detach_()
        d, w = self.params[0], self.params[1]
        with torch.no_grad():
            if self.op_hp.clamp_decay:
                for i in range(d.shape[1]):
                    for k in range(d.shape[2]) if self.op_hp.strictly_decreasing_decay else [None]:
                        sl = slice(None) if k is None else k
                        lo = self.min_decay if (k is None or k == 0) else d[:, i, k - 1]
                        hi = self.max_decay
                        if i > 0 and self.op_hp.enforce_long_decay_in_second_exponential:
                            hi = torch.clamp(d[:, 0, sl] / 1.01, max=self.max_decay)
                        cur = d[:, i, sl]
                        lo_t = torch.as_tensor(lo, dtype=cur.dtype, device=cur.device)
                        hi_t = torch.as_tensor(hi, dtype=cur.dtype, device=cur.device)
                        d[:, i, sl] = torch.minimum(torch.maximum(cur, lo_t), hi_t)     # torch.clamp(min,max): max wins
            lo, hi = 10 ** (self.Amin / 20), 10 ** (self.Amax / 20)
            w[:, 0] = torch.clamp(w[:, 0], min=lo, max=hi)
            for i in range(1, w.shape[1]):
                w[:, i] = torch.minimum(torch.clamp(w[:, i], min=lo), w[:, 0])
        assert not torch.isnan(d).any(), "decay is Nan"
        assert not torch.isnan(w).any(), "weights is Nan"

    def cons(self, X, length=None):
        """Consistency + minimum-phase projection (reference :333-351), batched over utterances."""
        L = X.shape[-1]
        X = F.pad(X, (1, 1))
        h = self.istft(X, length=length)
        h = F.pad(h, (0, self.hop_length))
        if self.op_hp.minimum_phase:
            h = reverb_utils.minimum_phase_version(h)
        if self.fix_direct_path:
            first = torch.full(h.shape[:-1] + (1,), 1 * (self.win_length / (self.hop_length * 2)), dtype=h.dtype, device=h.device)
            h = torch.cat([first, h[..., 1:]], dim=-1)
        X_rec = self.stft(h)[..., 1:-1]
        return X_rec[..., :L]


def create_blindop_handle(op_hp, sample_rate, num_utts, length):
    """``buddy_blindop_create`` from the reference op_hp block; returns (handle, dict of derived constants)."""
    import ctypes as C
    knots = [float(f) for f in op_hp.EQ_freqs]
    num_bands = len(knots) - 2
    if op_hp.init_single_value:
        t60 = [num_bands * [float(t)] for t in op_hp.init_params.T60_breakpoints]
        wts = [num_bands * [float(w)] for w in op_hp.init_params.multiexp_weighting]
    else:
        t60, wts = op_hp.init_params.T60_breakpoints, op_hp.init_params.multiexp_weighting
    frame_rate = sample_rate / op_hp.hop
    decay = 6.908 / (torch.tensor(t60, dtype=torch.float32) * frame_rate)
    max_decay = 6.908 / (op_hp.T60min * frame_rate)
    min_decay = 6.908 / (op_hp.T60max * frame_rate)
    lib = _lib.require_gpu()
    h = C.c_void_p()
    kn = (C.c_float * len(knots))(*knots)
    _lib.check(lib.buddy_blindop_create(int(num_utts), int(length), int(op_hp.Nf), int(decay.shape[0]), len(knots), kn, int(sample_rate),
                                        float(0.667), float(min_decay), float(max_decay), float(10 ** (op_hp.Amin / 20)),
                                        float(10 ** (op_hp.Amax / 20)), int(bool(op_hp.clamp_decay)),
                                        int(bool(op_hp.enforce_long_decay_in_second_exponential)), C.byref(h)))
    return h, dict(num_bands=num_bands, decay=decay, wts=wts, max_decay=max_decay, min_decay=min_decay, comp=0.667)


def create_stft_loss_handle(sample_rate, num_utts, length):
    """Library handle used only for its STFT-1024/512/128 + compressed-spectrum-loss machinery (informed operator): the blind filter
    parameters of the handle are placeholders."""
    import ctypes as C
    lib = _lib.require_gpu()
    h = C.c_void_p()
    kn = (C.c_float * 3)(0.0, sample_rate / 4.0, sample_rate / 2.0)
    _lib.check(lib.buddy_blindop_create(int(num_utts), int(length), 100, 1, 3, kn, int(sample_rate), float(0.667), 0.01, 1.0, 1.0, 100.0, 1, 0,
                                        C.byref(h)))
    return h


class _HipFirRecLoss(torch.autograd.Function):
    """sum_u weight * l2_comp_stft_summean(y_u, x_den_u * rir_u) for the informed operator, analytic gradient from the library."""

    @staticmethod
    def forward(ctx, x_den, op, weight):
        lib = _lib.require_gpu()
        x = x_den.contiguous().float()
        rir = op.params.detach().contiguous().float()
        M = rir.shape[-1]
        loss = torch.empty(x.shape[0], device=x.device)
        g = torch.empty_like(x)
        _lib.check(lib.buddy_blindop_fir_loss_grad(op._hip_h, _lib.ptr(x), _lib.ptr(rir), 0 if rir.dim() == 1 else M, M, float(weight),
                                                   _lib.ptr(loss), _lib.ptr(g), _lib.stream_ptr()))
        ctx.save_for_backward(g)
        op.last_rec_per_utt = loss
        return loss.sum()

    @staticmethod
    def backward(ctx, gout):
        g, = ctx.saved_tensors
        return gout * g, None, None


class BlindSubbandFilteringHIP(BlindSubbandFiltering):
    """Same interface, hand-written HIP backend (``buddy_blindop_*`` in ``include/buddy_hip.h``): parameters, Adam state, the
    filter H and every intermediate live on the device inside the library handle; forward and analytic backward of
    design_filter -> cons (iSTFT, minimum phase, STFT) -> subband FIR -> iSTFT -> STFT -> compressed-spectrum loss run as fused
    kernels, a whole ``optimize_op`` (reference EulerHeunSamplerDPS.py:71-113) is ONE library call."""

    def __init__(self, op_hp, sample_rate, magnitude_distance=True, H_cplx=False, num_utts=1, noise=None, device=None, backend=None,
                 length=None):
        SubbandFiltering.__init__(self, op_hp, sample_rate, device=device)
        import ctypes as C
        assert length is not None, "the HIP backend needs the signal length"
        assert op_hp.fix_EQ_extremes and op_hp.minimum_phase and op_hp.fix_direct_path and not op_hp.strictly_decreasing_decay
        self.U, self.noise, self.length = int(num_utts), noise, int(length)
        self.Amin, self.Amax = op_hp.Amin, op_hp.Amax
        knots = [float(f) for f in op_hp.EQ_freqs]
        self.num_bands = len(knots) - 2
        if op_hp.init_single_value:
            t60 = [self.num_bands * [float(t)] for t in op_hp.init_params.T60_breakpoints]
            wts = [self.num_bands * [float(w)] for w in op_hp.init_params.multiexp_weighting]
        else:
            t60, wts = op_hp.init_params.T60_breakpoints, op_hp.init_params.multiexp_weighting
        frame_rate = self.sample_rate / op_hp.hop
        decay = 6.908 / (torch.tensor(t60, dtype=torch.float32) * frame_rate)
        self.num_exponentials = decay.shape[0]
        self.max_decay = 6.908 / (op_hp.T60min * frame_rate)
        self.min_decay = 6.908 / (op_hp.T60max * frame_rate)
        self.comp = None              # compression exponent fixed at hip_bind (losses) -- default of the shipped configs
        lib = _lib.require_gpu()
        h = C.c_void_p()
        kn = (C.c_float * len(knots))(*knots)
        _lib.check(lib.buddy_blindop_create(self.U, self.length, int(op_hp.Nf), int(self.num_exponentials), len(knots), kn, int(sample_rate),
                                            float(0.667), float(self.min_decay), float(self.max_decay), float(10 ** (self.Amin / 20)),
                                            float(10 ** (self.Amax / 20)), int(bool(op_hp.clamp_decay)),
                                            int(bool(op_hp.enforce_long_decay_in_second_exponential)), C.byref(h)))
        self._h = h
        self._comp_created = 0.667
        d0 = decay.unsqueeze(0).repeat(self.U, 1, 1).to(self.device).contiguous()
        w0 = torch.tensor(wts, dtype=torch.float32).unsqueeze(0).repeat(self.U, 1, 1).to(self.device).contiguous()
        with torch.no_grad():
            ph = (self._rand((self.n_fft // 2 + 1, self.Nf)) * 2 * np.pi - np.pi).contiguous()
        _lib.check(lib.buddy_blindop_set_params(self._h, _lib.ptr(d0), _lib.ptr(w0), _lib.ptr(ph), 1, _lib.stream_ptr()))
        self.last_rec_per_utt = None
        if op_hp.init_phases == "random_coherent":
            self.update_H(use_noise=True)
        elif op_hp.init_phases == "random":
            self.update_H()
        else:
            raise NotImplementedError("This is not implemented yet")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib.load().buddy_blindop_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- state views (fresh device copies in the reference layout) ----
    def _get(self):
        E, NB, F = self.num_exponentials, self.num_bands, self.n_fft // 2 + 1
        d = torch.empty(self.U, E, NB, device=self.device); w = torch.empty_like(d)
        p = torch.empty(self.U, F, self.Nf, device=self.device)
        _lib.check(_lib.load().buddy_blindop_get_params(self._h, _lib.ptr(d), _lib.ptr(w), _lib.ptr(p), _lib.stream_ptr()))
        return d, w, p

    @property
    def params(self):
        d, w, _ = self._get()
        return [d, w]

    @property
    def params_phases(self):
        return [self._get()[2]]

    @property
    def H(self):
        out = torch.empty(self.U, self.n_fft // 2 + 1, self.Nf, 2, device=self.device)
        _lib.check(_lib.load().buddy_blindop_get_H(self._h, _lib.ptr(out), _lib.stream_ptr()))
        return torch.view_as_complex(out)

    @H.setter
    def H(self, v):
        if v is not None:
            raise NotImplementedError("H is owned by the HIP operator")

    def set_params(self, decay=None, weights=None, phases=None, reset_adam=False):
        c = lambda t: None if t is None else t.to(self.device).float().contiguous()
        d, w, p = c(decay), c(weights), c(phases)
        _lib.check(_lib.load().buddy_blindop_set_params(self._h, _lib.ptr(d), _lib.ptr(w), _lib.ptr(p), int(reset_adam), _lib.stream_ptr()))

    def update_H(self, rir=None, H=None, use_noise=False, noise=None, phases=None):
        if rir is not None or H is not None:
            raise NotImplementedError("informed H is the torch SubbandFiltering operator")
        if phases is not None:
            self.set_params(phases=phases)
        n = None
        if use_noise:
            n = (noise if noise is not None else self._randn((self.length_rir,))).to(self.device).float().contiguous()
        _lib.check(_lib.load().buddy_blindop_update_H(self._h, _lib.ptr(n), _lib.stream_ptr()))

    def project_params(self):
        """reference :298-331 on the device-resident parameters (also applied inside buddy_blindop_optimize after every Adam step)"""
        _lib.check(_lib.load().buddy_blindop_project(self._h, _lib.stream_ptr()))

    def design_filter(self, correct_OLA=True):
        """reference :241-251 -> (U, F, Nf) magnitudes from the current decay / weights"""
        assert correct_OLA
        A = torch.empty(self.U, self.n_fft // 2 + 1, self.Nf, device=self.device)
        _lib.check(_lib.load().buddy_blindop_design_filter(self._h, _lib.ptr(A), _lib.stream_ptr()))
        return A

    def apply_stft(self, x):
        """reference :41-52 for signals of the bound length: (U, F, T) complex64"""
        xx = (x.unsqueeze(0) if x.dim() == 1 else x).contiguous().float()
        if tuple(xx.shape) != (self.U, self.length):
            raise NotImplementedError(f"the HIP operator transforms (U={self.U}, L={self.length}) signals, got {tuple(xx.shape)}")
        T = 1 + (self.length + self.win_length) // self.hop_length
        X = torch.empty(self.U, self.n_fft // 2 + 1, T, 2, device=self.device)
        _lib.check(_lib.load().buddy_blindop_apply_stft(self._h, _lib.ptr(xx), _lib.ptr(X), _lib.stream_ptr()))
        return torch.view_as_complex(X)

    def minimum_phase(self, h):
        """utils/reverb_utils.py:9-23 at the size cons() uses: h (U, hop * (Nf + 1))"""
        hh = h.to(self.device).float().contiguous()
        assert tuple(hh.shape) == (self.U, self.length_rir + self.hop_length)
        out = torch.empty_like(hh)
        _lib.check(_lib.load().buddy_blindop_minphase(self._h, _lib.ptr(hh), _lib.ptr(out), _lib.stream_ptr()))
        return out

    def adam_state(self):
        """torch.optim.Adam state of [decay, weights, phases] in the reference layouts: dict of (exp_avg, exp_avg_sq), and the step count"""
        import ctypes as C
        E, NB, F = self.num_exponentials, self.num_bands, self.n_fft // 2 + 1
        mk = lambda *sh: torch.empty(*sh, device=self.device)
        md, vd, mw, vw = mk(self.U, E, NB), mk(self.U, E, NB), mk(self.U, E, NB), mk(self.U, E, NB)
        mp, vp = mk(self.U, F, self.Nf), mk(self.U, F, self.Nf)
        step = C.c_int(0)
        _lib.check(_lib.load().buddy_blindop_get_adam(self._h, _lib.ptr(md), _lib.ptr(vd), _lib.ptr(mw), _lib.ptr(vw), _lib.ptr(mp), _lib.ptr(vp),
                                                      C.byref(step), _lib.stream_ptr()))
        return dict(decay=(md, vd), weights=(mw, vw), phases=(mp, vp)), int(step.value)

    def degradation(self, x, mode="waveform", H=None, detach_operator=False):
        assert mode == "waveform" and H is None
        squeeze = x.dim() == 1
        xx = (x.unsqueeze(0) if squeeze else x).contiguous().float()
        y = torch.empty_like(xx)
        _lib.check(_lib.load().buddy_blindop_degrade(self._h, _lib.ptr(xx), _lib.ptr(y), _lib.stream_ptr()))
        return y.squeeze(0) if squeeze else y

    def get_time_RIR(self, excitation=None, H=None):
        assert excitation is None and H is None
        out = torch.empty(self.U, self.length_rir + 1024, device=self.device)
        _lib.check(_lib.load().buddy_blindop_time_rir(self._h, _lib.ptr(out), _lib.stream_ptr()))
        return out.squeeze(0) if self.U == 1 else out

    # ---- sampler fast paths ----
    def hip_bind(self, y, ps):
        """cache comp(STFT(y)); read loss weights / compression from the posterior_sampling config"""
        # the regulariser is gated like the reference gates it (EulerHeunSamplerDPS.py:94,200): only loss.name == "none" turns it off;
        # RIR_noise_regularization.use is never read there
        reg_loss = ps.RIR_noise_regularization.loss
        for l in (ps.rec_loss, ps.rec_loss_params) + (() if reg_loss.name == "none" else (reg_loss,)):
            assert l.name == "l2_comp_stft_summean" and abs(l.compression_factor - self._comp_created) < 1e-9, \
                "HIP operator supports l2_comp_stft_summean with compression_factor 0.667"
        self.w_rec = float(ps.rec_loss.get("weight", 1.0))
        self.w_rec_params = float(ps.rec_loss_params.get("weight", 1.0))
        self.w_reg = None if reg_loss.name == "none" else float(reg_loss.get("weight", 1.0))
        self.reg = ps.RIR_noise_regularization
        self.hp = ps.blind_hp
        yy = y.contiguous().float()
        _lib.check(_lib.load().buddy_blindop_set_y(self._h, _lib.ptr(yy), _lib.stream_ptr()))
        self.set_params(reset_adam=True)          # fresh Adam state, like constructing torch.optim.Adam in predict_conditional

    def hip_rec_loss(self, x_den):
        return _HipRecLoss.apply(x_den, self, self.w_rec)

    def hip_optimize(self, x_den, t):
        n_it = int(self.hp.op_updates_per_step)
        noise = None
        if self.w_reg is not None:
            Lr = self.length_rir + 1024
            if self.noise is None:
                noise = torch.randn(n_it, self.U, Lr, device=self.device)      # the reference draws this one on the device too (randn_like(rir_time))
            else:
                noise = torch.stack([torch.stack([n.randn((Lr,)) for n in self.noise]) for _ in range(n_it)]).to(self.device)
            noise = noise.contiguous()
        t_op = max(min(float(t), self.reg.crop_sigma_max), self.reg.crop_sigma_min)
        xd = x_den.contiguous().float()
        _lib.check(_lib.load().buddy_blindop_optimize(self._h, _lib.ptr(xd), _lib.ptr(noise), float(t_op), n_it, self.w_rec_params,
                                                      float(self.w_reg or 0.0), float(self.hp.lr_op), float(self.hp.beta1), float(self.hp.beta2),
                                                      float(self.hp.weight_decay), _lib.stream_ptr()))
