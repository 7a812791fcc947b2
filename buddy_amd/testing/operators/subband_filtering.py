"""Subband (STFT-domain) blind reverberation operator, same surface as reference ``testing/operators/subband_filtering.py``
(``BlindSubbandFiltering`` :142-351 and the methods it inherits from ``SubbandFiltering`` :8-136), batched per utterance: every parameter /
filter tensor carries a leading utterance axis ``U`` (U = 1 reproduces the reference exactly; U > 1 is the per-utterance vmap of it -- no
cross-utterance coupling anywhere).

HIP only.  Parameters, Adam state, the filter H and every intermediate live inside a ``buddy_blindop_*`` handle of ``libbuddy_hip.so``
(hand-written forward + analytic backward kernels, one library call per ``optimize_op``).  There is no torch-op implementation in the product:
constructing the operator without a GPU raises ``BuddyHipError``.  The torch-op restatement used by the CPU host-logic tests and by the
on-GPU autograd cross-checks lives in ``oracle/batched/operators.py``.
"""
from __future__ import annotations

import numpy as np
import torch

from ... import _lib
from .shared import Operator
from ._stft import OperatorSTFT


class SubbandFiltering(Operator, OperatorSTFT):
    """Constants of the STFT-domain filter model (reference :8-33): Nf frames of filter per band, hop / window of the operator STFT."""

    def __init__(self, op_hp, sample_rate, device=None):
        super().__init__()
        self.op_hp = op_hp
        dev = device if device is not None else "cuda"
        self._init_stft(op_hp, sample_rate, dev)
        assert self.n_fft >= self.win_length, "n_fft must be greater than 2*win_length to avoid temporal aliasing"
        self.Nf = self.op_hp.Nf
        self.length_rir = self.hop_length * self.Nf
        self.time = torch.arange(self.Nf, dtype=torch.float32) / (self.sample_rate / self.hop_length)


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


class BlindSubbandFiltering(SubbandFiltering):
    """Reference interface, hand-written HIP backend (``buddy_blindop_*`` in ``include/buddy_hip.h``): parameters, Adam state, the
    filter H and every intermediate live on the device inside the library handle; forward and analytic backward of
    design_filter -> cons (iSTFT, minimum phase, STFT) -> subband FIR -> iSTFT -> STFT -> compressed-spectrum loss run as fused
    kernels, a whole ``optimize_op`` (reference EulerHeunSamplerDPS.py:71-113) is ONE library call."""

    def __init__(self, op_hp, sample_rate, magnitude_distance=True, H_cplx=False, num_utts=1, noise=None, device=None, backend=None,
                 length=None):
        import ctypes as C
        lib = _lib.require_gpu()              # no GPU / no library: BuddyHipError, never a CPU path
        if backend not in (None, "hip"):
            raise NotImplementedError(f"BlindSubbandFiltering(backend={backend!r}): the product has the HIP operator only")
        if length is None:
            raise ValueError("BlindSubbandFiltering needs the signal length in samples (length=...): the library handle is built for (U, L)")
        SubbandFiltering.__init__(self, op_hp, sample_rate, device=device)
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

    # -- noise plumbing (the reference draws with torch.rand / torch.randn on the fly) --------------------
    def _rand(self, shape):
        if self.noise is None:
            return torch.rand((self.U,) + tuple(shape)).to(self.device)
        return torch.stack([n.rand(shape) for n in self.noise]).to(self.device)

    def _randn(self, shape):
        if self.noise is None:
            return torch.randn((self.U,) + tuple(shape)).to(self.device)
        return torch.stack([n.randn(shape) for n in self.noise]).to(self.device)

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
            raise NotImplementedError("an externally given H / RIR is the informed scenario (RIROperator); the blind operator designs H from its parameters")
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
        """Adam state (exp_avg, exp_avg_sq as the reference's optimizer keeps them) of [decay, weights, phases] in the reference layouts: dict of (exp_avg, exp_avg_sq), and the step count"""
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
        self.set_params(reset_adam=True)          # fresh Adam state, like constructing the optimizer in the reference's predict_conditional (:193)

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
                # injected per-utterance streams (tests, the float64 arbiter runs): drawn on the host in reference call order, ONE pinned buffer and
                # ONE asynchronous copy per step -- a pageable .to(device) makes the host wait for the whole queue, i.e. drains the GPU every step
                host = torch.stack([torch.stack([n.randn((Lr,)) for n in self.noise]) for _ in range(n_it)]).contiguous()
                cuda = torch.device(self.device).type == "cuda"
                noise = (host.pin_memory() if cuda else host).to(self.device, non_blocking=cuda)
            noise = noise.contiguous()
        t_op = max(min(float(t), self.reg.crop_sigma_max), self.reg.crop_sigma_min)
        xd = x_den.contiguous().float()
        _lib.check(_lib.load().buddy_blindop_optimize(self._h, _lib.ptr(xd), _lib.ptr(noise), float(t_op), n_it, self.w_rec_params,
                                                      float(self.w_reg or 0.0), float(self.hp.lr_op), float(self.hp.beta1), float(self.hp.beta2),
                                                      float(self.hp.weight_decay), _lib.stream_ptr()))


