"""TEST INFRASTRUCTURE, not product: torch-op (ATen / rocFFT / autograd) form of the reverberation operators, batched per utterance -- the
same arithmetic as ``oracle/operators_ref.py`` (B = 1, reference-faithful) with a leading utterance axis.  Lived inside ``buddy_amd/`` as
``backend="torch"`` through round 3; the product package now contains only the HIP operator.  Used by
  * the CPU host-logic tests (sampler control flow, batching, noise-stream order, rank sharding) -- ``tests/test_host_logic.py``,
    ``tests/test_distributed_cpu.py``;
  * the on-GPU cross-checks of the HIP operator against autograd -- ``tests/test_hip_operator.py``.
Reference lines: ``testing/operators/subband_filtering.py`` (``SubbandFiltering`` :8-136, ``BlindSubbandFiltering`` :142-351),
``utils/reverb_utils.py:3-23`` (hilbert / minimum phase), STFT conventions ``reverb.py:54-84``."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class Operator(torch.nn.Module):
    """reference testing/operators/shared.py"""

    def degradation(self, *a, **k):
        raise NotImplementedError

    def update_params(self, *a, **k):
        raise NotImplementedError


class OperatorSTFT:
    def _init_stft(self, op_hp, sample_rate, device):
        self.sample_rate = sample_rate
        self.op_hp = op_hp
        self.device = device
        self.n_fft = op_hp.NFFT
        self.win_length = op_hp.win_length
        self.hop_length = op_hp.hop
        w = op_hp.window
        if w == "hann":
            self.window = torch.hann_window(self.win_length, device=device)
            assert self.hop_length <= self.win_length / 4, "hop length must be less than 1/4 of win_length to avoid temporal aliasing"
        else:
            raise NotImplementedError("window type {} not implemented".format(w))
        self.window_padded = F.pad(self.window, (0, self.n_fft - self.win_length), mode="constant", value=0)
        self.freqs = torch.fft.rfftfreq(self.n_fft, d=1 / sample_rate).to(device)
        self._norm = torch.sqrt(torch.sum(self.window_padded ** 2))

    def stft(self, x):
        return torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window_padded, center=True,
                          onesided=True, return_complex=True, normalized=False, pad_mode="constant")

    def istft(self, X, length=None):
        return torch.istft(X, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window_padded, onesided=True,
                           center=True, normalized=False, return_complex=False, length=length)

    def apply_stft(self, x):
        if x.dim() == 1:
            x = x.unsqueeze(0)
        elif x.dim() != 2:
            raise ValueError("x must have shape (batch, samples) or (samples)")
        return self.stft(F.pad(x, (0, self.win_length))) / self._norm

    def apply_istft(self, X, length=None):
        if length is None:
            print("Warning: length is None, istft may crash")
            length_param = None
        else:
            length_param = length + self.win_length // 2
        X = X * self._norm       # the reference scales its argument in place (subband_filtering.py:61); callers never reuse it
        x = self.istft(X, length=length_param)
        return x[..., self.win_length // 2:]


class StftOnly(OperatorSTFT):
    """just the operator STFT (apply_stft / apply_istft) for an op_hp block: lets the torch loss formulas run on signals produced by the
    product's HIP operators"""

    def __init__(self, op_hp, sample_rate=16000, device="cpu"):
        self._init_stft(op_hp, sample_rate, device)


from oracle.operators_ref import hilbert_ref as hilbert, minimum_phase_ref as minimum_phase_version, linear_interp as _linear_interp_kc   # noqa: E402


def linear_interp(knots, values, query):
    """values (..., K) -> (..., Q): the oracle's piecewise-linear interpolation (restated from torchcde's API semantics, parity unpinned) on one channel."""
    return _linear_interp_kc(knots, values.unsqueeze(-1), query).squeeze(-1)


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
        self.time = torch.arange(self.Nf, dtype=torch.get_default_dtype()) / (self.sample_rate / self.hop_length)

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
            x = torch.zeros(int(self.length_rir + 1024), dtype=torch.get_default_dtype(), device=self.device)
            x[0] = 1
        else:
            x = torch.as_tensor(excitation, dtype=torch.get_default_dtype(), device=self.device)
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


class BlindSubbandFiltering(SubbandFiltering):
    def __init__(self, op_hp, sample_rate, magnitude_distance=True, H_cplx=False, num_utts=1, noise=None, device=None, backend=None,
                 length=None):
        """``num_utts``: utterances handled by this operator object (per-utterance parameters).  ``noise``: optional list
        of per-utterance noise sources with ``rand(shape)`` / ``randn(shape)`` (parity runs); default torch RNG.  ``backend`` / ``length``
        are accepted and ignored (the product class's signature)."""
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
        t60 = torch.tensor(t60, dtype=torch.get_default_dtype(), device=self.device)
        wts = torch.tensor(wts, dtype=torch.get_default_dtype(), device=self.device)
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
        n = torch.arange(0, Nf, device=self.device).to(torch.get_default_dtype())
        inner = (weights.unsqueeze(-1) * decay_bp.unsqueeze(-1) ** (-n)).sum(1)  # (U,bands,Nf)
        if self.fix_EQ_extremes:
            z = torch.zeros(inner.shape[0], 1, Nf, device=self.device)
            dm = torch.cat([z, inner, z], dim=1)                                   # rows 0 and -1 stay zero
        else:
            dm = inner
        dm = torch.log(dm.transpose(1, 2) + 1e-6)                                # (U,Nf,knots)
        H2 = linear_interp(self.EQ_freqs.to(torch.get_default_dtype()), dm, self.freqs)       # (U,Nf,F)
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
        T60s = torch.tensor(params_dict.T60_breakpoints, dtype=torch.get_default_dtype(), device=self.device)
        w = torch.tensor(params_dict.multiexp_weighting, dtype=torch.get_default_dtype(), device=self.device)
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
            h = minimum_phase_version(h)
        if self.fix_direct_path:
            first = torch.full(h.shape[:-1] + (1,), 1 * (self.win_length / (self.hop_length * 2)), dtype=h.dtype, device=h.device)
            h = torch.cat([first, h[..., 1:]], dim=-1)
        X_rec = self.stft(h)[..., 1:-1]
        return X_rec[..., :L]
