"""STFT parameters shared by both operators (reference ``reverb.py:54-84`` == ``subband_filtering.py:41-80``): n_fft 1024, hann(512)
zero-padded to 1024, hop 128, centre with constant padding, / sqrt(sum w^2).  The transforms themselves run inside ``libbuddy_hip.so``
(``fft1024_r2c`` / ``c2r`` kernels behind ``buddy_blindop_apply_stft`` and the fused loss / operator calls); this mixin only carries the
numbers and the window the library is built for."""
from __future__ import annotations

import math


class OperatorSTFT:
    def _init_stft(self, op_hp, sample_rate, device):
        self.sample_rate = sample_rate
        self.op_hp = op_hp
        self.device = device
        self.n_fft = op_hp.NFFT
        self.win_length = op_hp.win_length
        self.hop_length = op_hp.hop
        if op_hp.window != "hann":
            raise NotImplementedError("window type {} not implemented".format(op_hp.window))
        assert self.hop_length <= self.win_length / 4, "hop length must be less than 1/4 of win_length to avoid temporal aliasing"
        # sqrt(sum_k hann(k)^2) of the periodic Hann window of win_length samples = sqrt(3 * win_length / 8)
        self._norm = math.sqrt(3.0 * self.win_length / 8.0)
