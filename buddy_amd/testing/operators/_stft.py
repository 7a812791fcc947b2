"""STFT helpers shared by both operators (reference ``reverb.py:54-84`` == ``subband_filtering.py:41-80``):
n_fft 1024, hann(512) zero-padded to 1024, hop 128, centre with constant padding, / sqrt(sum w^2)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


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
