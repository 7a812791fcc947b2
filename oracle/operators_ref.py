"""Oracle: reverb operators, DSP utils and the reconstruction loss, PyTorch fp32/complex64 on CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates reference ``utils/reverb_utils.py``,
``testing/operators/{reverb,subband_filtering}.py`` and the live branch of ``utils/losses.py:59-64``.
Differentiable through torch autograd, exactly like the reference (which is how DPS gets its gradients).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- DSP utils
def hilbert_ref(h):
    """reference utils/reverb_utils.py:3-7 -- window is [2]*N/2 ++ [0]*N/2 (DC/Nyquist NOT special-cased;
    torch.heaviside(linspace(-1,1,N), 1) flipped)."""
    n = h.shape[-1]
    window = 2 * torch.heaviside(torch.linspace(-1, 1, steps=n), values=torch.ones(1)).to(h.device)
    window = torch.flip(window, dims=(-1,))
    return torch.fft.ifft(window * torch.fft.fft(h))


def minimum_phase_ref(h):
    """reference utils/reverb_utils.py:9-23 (batched over leading dims)."""
    T = h.shape[-1]
    h = F.pad(h, (0, T))
    H = torch.fft.fft(h)
    log_abs = torch.log(torch.abs(H) + 1e-8)
    phase = -torch.imag(hilbert_ref(log_abs))
    e = torch.exp(1j * phase)
    out = torch.real(torch.fft.ifft(torch.abs(H).type(e.dtype) * e))
    return out[..., :-T]


def fast_apply_rir_ref(y, filt):
    """reference utils/reverb_utils.py:25-61 (rm_delay=False, zero_pad=False): linear convolution through a
    full complex FFT of size 2^ceil(log2(N+M-1)); keep first N samples.  y (B,N), filt (M,)."""
    N, M = y.shape[-1], filt.shape[-1]
    n = int(2 ** math.ceil(math.log2(N + M - 1)))
    Y = torch.fft.fft(y, n, dim=-1)
    H = torch.fft.fft(filt[None], n, dim=-1)
    return torch.fft.ifft(Y * H, n, dim=-1)[..., :N].real


def linear_interp(knots_t, values, query):
    """torchcde.LinearInterpolation(coeffs, t).evaluate(query) restated from API semantics (piecewise linear
    over knots; call site reference subband_filtering.py:233-235).  values (..., K, C) at knots_t (K,),
    query (Q,) -> (..., Q, C).  PARITY UNPINNED (torchcde absent, see oracle/__init__.py)."""
    K = knots_t.shape[0]
    idx = torch.bucketize(query.detach(), knots_t.detach()) - 1
    idx = idx.clamp(0, K - 2)
    t0, t1 = knots_t[idx], knots_t[idx + 1]
    frac = ((query - t0) / (t1 - t0)).unsqueeze(-1)
    v0 = values[..., idx, :]
    v1 = values[..., idx + 1, :]
    return v0 + frac * (v1 - v0)


# --------------------------------------------------------------------------- operator STFT (both operators share it)
class _OpSTFT:
    """STFT helpers shared by RIROperator / SubbandFiltering -- reference reverb.py:54-84 == subband_filtering.py:41-80."""

    def _init_stft(self, op_hp, sample_rate):
        self.sample_rate = sample_rate
        self.n_fft = op_hp.NFFT
        self.win_length = op_hp.win_length
        self.hop_length = op_hp.hop
        assert op_hp.window == "hann"
        assert self.hop_length <= self.win_length / 4
        self.window = torch.hann_window(self.win_length)
        self.window_padded = F.pad(self.window, (0, self.n_fft - self.win_length))
        self.freqs = torch.fft.rfftfreq(self.n_fft, d=1 / sample_rate)
        self.norm = torch.sqrt(torch.sum(self.window_padded ** 2))

    def stft(self, x):
        return torch.stft(x, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window_padded,
                          center=True, onesided=True, return_complex=True, normalized=False, pad_mode="constant")

    def istft(self, X, length=None):
        return torch.istft(X, self.n_fft, hop_length=self.hop_length, win_length=self.n_fft, window=self.window_padded,
                           onesided=True, center=True, normalized=False, return_complex=False, length=length)

    def apply_stft(self, x):
        if x.dim() == 1:
            x = x[None]
        return self.stft(F.pad(x, (0, self.win_length))) / self.norm

    def apply_istft(self, X, length):
        X = X * self.norm  # reference scales in place (subband_filtering.py:61); value-identical
        x = self.istft(X, length=length + self.win_length // 2)
        return x[..., self.win_length // 2:]


class RIROperatorRef(_OpSTFT):
    """reference testing/operators/reverb.py:8-88."""

    def __init__(self, op_hp, sample_rate=16000):
        self._init_stft(op_hp, sample_rate)
        self.params = None

    def update_params(self, k):
        self.params = k

    def degradation(self, x, **_):
        return fast_apply_rir_ref(x, self.params)

    def get_time_RIR(self):
        return self.params


class BlindSubbandFilteringRef(_OpSTFT):
    """reference testing/operators/subband_filtering.py:8-351 (SubbandFiltering + BlindSubbandFiltering) for the
    shipped op_hp (fix_EQ_extremes, init_single_value, random_coherent, minimum_phase, fix_direct_path)."""

    def __init__(self, op_hp, sample_rate, noise):
        self._init_stft(op_hp, sample_rate)
        self.op_hp = op_hp
        self.Nf = op_hp.Nf
        self.length_rir = self.hop_length * self.Nf
        self.EQ_freqs = torch.tensor([float(f) for f in op_hp.EQ_freqs])
        assert op_hp.fix_EQ_extremes and op_hp.init_single_value
        self.num_bands = len(self.EQ_freqs) - 2
        t60 = torch.tensor([self.num_bands * [float(t)] for t in op_hp.init_params.T60_breakpoints])
        wts = torch.tensor([self.num_bands * [float(w)] for w in op_hp.init_params.multiexp_weighting])
        frame_rate = self.sample_rate / self.hop_length
        self.decay = (6.908 / (t60 * frame_rate)).requires_grad_(True)            # :167, :181
        self.weights = wts.clone().requires_grad_(True)                           # :182
        self.max_decay = 6.908 / (op_hp.T60min * frame_rate)                      # :184
        self.min_decay = 6.908 / (op_hp.T60max * frame_rate)                      # :185
        self.phases = (noise.rand((self.n_fft // 2 + 1, self.Nf)) * 2 * np.pi - np.pi).requires_grad_(True)  # :188
        self.params = [self.decay, self.weights]
        self.params_phases = [self.phases]
        h = torch.zeros(self.length_rir)
        h[0] = self.win_length / (self.hop_length * 2)
        self.direct_path_mag = self.stft(h)[:, 1:].abs()                          # :201-205
        self.H = None
        assert op_hp.init_phases == "random_coherent"
        self.update_H(use_noise=True, noise=noise)                                # :197-198

    # -- filter design ------------------------------------------------------
    def design_filter(self):
        """design_subband_filter + correct_OLA + direct path -- reference :212-251."""
        Nf = self.Nf
        decay_bp = torch.exp(self.params[0])
        n = torch.arange(0, Nf).to(torch.get_default_dtype())[None, None, :]
        inner = (self.params[1].unsqueeze(-1) * decay_bp.unsqueeze(-1) ** (-n)).sum(0)   # (25, Nf)
        zero = torch.zeros(1, Nf)
        dm = torch.cat([zero, inner, zero], dim=0)                                      # rows 0 and 26 stay 0
        dm = torch.log(dm.transpose(0, 1) + 1e-6)                                       # (Nf, 27)
        H2 = linear_interp(self.EQ_freqs, dm.unsqueeze(-1), self.freqs)                 # (Nf, 513, 1)
        A = torch.exp(H2.squeeze(-1).transpose(0, 1)) + 1e-6                            # (513, Nf)
        K = int(self.win_length / self.hop_length - 1)
        win_sum = torch.sum(self.window)
        cols = []
        for k in range(Nf):
            if k < K:
                cols.append(A[:, k] / (win_sum / torch.sum(self.window[int((K - k) * self.hop_length):])))
            else:
                cols.append(A[:, k])
        A = torch.stack(cols, dim=1)
        return A + self.direct_path_mag

    def cons(self, X, length):
        """reference :333-351."""
        L = X.shape[-1]
        X = F.pad(X, (1, 1))
        h = self.istft(X, length=length)
        h = F.pad(h, (0, self.hop_length))
        h = minimum_phase_ref(h)
        first = torch.full((1,), self.win_length / (self.hop_length * 2), dtype=h.dtype)
        h = torch.cat([first, h[1:]])                                                   # h[0] = 2.0 (:346)
        return self.stft(h)[:, 1:-1][..., :L]

    def update_H(self, use_noise=False, noise=None):
        """reference :261-285."""
        A = self.design_filter()
        if use_noise:
            n = noise.randn((self.length_rir,))
            N = (self.stft(n) / self.norm)[:, 1:]
            self.H = self.cons(A * torch.exp(1j * N.angle()), self.length_rir)
            self.phases = torch.angle(self.H).detach()
            self.params_phases[0] = self.phases
        else:
            self.H = self.cons(A * torch.exp(1j * self.params_phases[0]), self.length_rir)

    # -- degradation -------------------------------------------------------
    def subband_filtering(self, X, H):
        """reference :67-74: per-band causal FIR along frames, complex, one pre-impulse frame."""
        pre = int((self.win_length // self.hop_length) / 2) - 1
        Hf = torch.flip(H, dims=[-1]).unsqueeze(1)
        Xp = F.pad(X, (Hf.shape[-1] - 1 - pre, pre))
        return F.conv1d(Xp, Hf, groups=Hf.shape[0])

    def degradation(self, x, H=None, **_):
        shape = x.shape
        X = self.apply_stft(x)
        Y = self.subband_filtering(X, self.H if H is None else H)
        y = self.apply_istft(Y, shape[-1])
        return y.squeeze(0) if len(shape) == 1 else y

    def get_time_RIR(self):
        x = torch.zeros(int(self.length_rir + 1024))
        x[0] = 1
        return self.degradation(x)

    def project_params(self):
        """reference :298-331 (clamp_decay, not strictly_decreasing, enforce_long_decay_in_second_exponential)."""
        with torch.no_grad():
            d, w = self.params[0], self.params[1]
            for i in range(d.shape[0]):
                for k in range(d.shape[1]):
                    mx = self.max_decay
                    if i > 0:
                        mx = min(float(d[0][k]) / 1.01, mx)
                    d[i][k] = torch.clamp(d[i][k], min=self.min_decay, max=mx)
            lo, hi = 10 ** (self.op_hp.Amin / 20), 10 ** (self.op_hp.Amax / 20)
            for k in range(w.shape[1]):
                w[0][k] = torch.clamp(w[0][k], min=lo, max=hi)
                for i in range(1, w.shape[0]):
                    w[i][k] = torch.clamp(w[i][k], min=lo, max=float(w[0][k]))


# --------------------------------------------------------------------------- loss
def l2_comp_stft_summean(op, x, x_hat, weight, c):
    """reference utils/losses.py:59-64 with frequency weighting == ones (key-name mismatch)."""
    X, Xh = op.apply_stft(x), op.apply_stft(x_hat)
    Xc = (X.abs() + 1e-8) ** c * torch.exp(1j * X.angle())
    Xhc = (Xh.abs() + 1e-8) ** c * torch.exp(1j * Xh.angle())
    return weight * torch.mean(torch.sum((Xc - Xhc).abs() ** 2, dim=-2))


def get_loss_ref(loss_args, op):
    if loss_args.name == "none":
        return None
    assert loss_args.name == "l2_comp_stft_summean"
    w, c = loss_args.get("weight", 1.0), loss_args.compression_factor
    return lambda x, x_hat: l2_comp_stft_summean(op, x, x_hat, w, c)
