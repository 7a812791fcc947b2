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


# ---- the differentiable surface (round 6): the pieces the REFERENCE's own sampler autograds through, each a torch.autograd.Function whose forward and
# backward are library calls (buddy_blindop_* and their _vjp entries).  H travels between them as its real view (U, 513, Nf, 2): no complex-gradient
# conventions inside the Functions; ``operator.H`` hands out the complex view.
class _UpdateHFn(torch.autograd.Function):
    """H = cons(design_filter(decay, weights) * exp(j phases)) (reference subband_filtering.py:253-285) from the operator's persistent parameter tensors."""

    @staticmethod
    def forward(ctx, decay, weights, phases, op):
        lib = _lib.require_gpu()
        c = lambda t: t.detach().to(op.device).float().contiguous()
        d, w, p = c(decay), c(weights), c(phases)
        _lib.check(lib.buddy_blindop_set_params(op._h, _lib.ptr(d), _lib.ptr(w), _lib.ptr(p), 0, _lib.stream_ptr()))
        _lib.check(lib.buddy_blindop_update_H(op._h, None, _lib.stream_ptr()))
        H = torch.empty(op.U, op.n_fft // 2 + 1, op.Nf, 2, device=op.device)
        _lib.check(lib.buddy_blindop_get_H(op._h, _lib.ptr(H), _lib.stream_ptr()))
        op._h_epoch += 1
        ctx.op, ctx.epoch = op, op._h_epoch
        return H

    @staticmethod
    def backward(ctx, gH):
        op = ctx.op
        if ctx.epoch != op._h_epoch:
            raise _lib.BuddyHipError("backward through a stale update_H: the handle keeps the state of its LAST update_H (call backward before the next one)")
        E, NB, F = op.num_exponentials, op.num_bands, op.n_fft // 2 + 1
        gd = torch.empty(op.U, E, NB, device=op.device); gw = torch.empty_like(gd); gp = torch.empty(op.U, F, op.Nf, device=op.device)
        _lib.check(_lib.load().buddy_blindop_update_H_vjp(op._h, _lib.ptr(gH.contiguous().float()), _lib.ptr(gd), _lib.ptr(gw), _lib.ptr(gp), _lib.stream_ptr()))
        return gd, gw, gp, None


class _DegradeFn(torch.autograd.Function):
    """y = degradation(x) with the operator's current H (reference :82-101); differentiable w.r.t. x and -- when H carries a graph -- w.r.t. H."""

    @staticmethod
    def forward(ctx, x, Hr, op):
        xx = x.contiguous().float()
        y = torch.empty_like(xx)
        _lib.check(_lib.require_gpu().buddy_blindop_degrade(op._h, _lib.ptr(xx), _lib.ptr(y), _lib.stream_ptr()))
        ctx.op, ctx.epoch = op, op._h_epoch
        ctx.save_for_backward(xx)
        return y

    @staticmethod
    def backward(ctx, gy):
        op = ctx.op
        if ctx.epoch != op._h_epoch:
            raise _lib.BuddyHipError("backward through a degradation whose H has been rebuilt since (update_H): differentiate before the next update_H")
        x, = ctx.saved_tensors
        want_x, want_H = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = torch.empty_like(x) if want_x else None
        gH = torch.empty(op.U, op.n_fft // 2 + 1, op.Nf, 2, device=x.device) if want_H else None
        if want_x or want_H:
            _lib.check(_lib.load().buddy_blindop_degrade_vjp(op._h, _lib.ptr(x), _lib.ptr(gy.contiguous().float()), _lib.ptr(gx), _lib.ptr(gH), _lib.stream_ptr()))
        return gx, gH, None


class _TimeRirFn(torch.autograd.Function):
    """get_time_RIR (reference :103-113) = degradation of the unit impulse; differentiable w.r.t. H."""

    @staticmethod
    def forward(ctx, Hr, op):
        out = torch.empty(op.U, op.length_rir + 1024, device=op.device)
        _lib.check(_lib.require_gpu().buddy_blindop_time_rir(op._h, _lib.ptr(out), _lib.stream_ptr()))
        ctx.op, ctx.epoch = op, op._h_epoch
        return out

    @staticmethod
    def backward(ctx, g):
        op = ctx.op
        if ctx.epoch != op._h_epoch:
            raise _lib.BuddyHipError("backward through a time RIR whose H has been rebuilt since (update_H)")
        gH = torch.empty(op.U, op.n_fft // 2 + 1, op.Nf, 2, device=op.device)
        _lib.check(_lib.load().buddy_blindop_time_rir_vjp(op._h, _lib.ptr(g.contiguous().float()), _lib.ptr(gH), _lib.stream_ptr()))
        return gH, None


class _StftFn(torch.autograd.Function):
    """apply_stft (reference :41-52) -> real view (U, 513, frames, 2); backward = the adjoint transform (library handle ``h`` of the right (U, L))."""

    @staticmethod
    def forward(ctx, x, h, frames):
        xx = x.contiguous().float()
        U, n = xx.shape
        X = torch.empty(U, 513, frames, 2, device=xx.device)
        _lib.check(_lib.require_gpu().buddy_blindop_stft(h, _lib.ptr(xx), n, _lib.ptr(X), _lib.stream_ptr()))
        ctx.h, ctx.n = h, n
        return X

    @staticmethod
    def backward(ctx, G):
        gx = torch.empty(G.shape[0], ctx.n, device=G.device)
        _lib.check(_lib.load().buddy_blindop_stft_adjoint(ctx.h, _lib.ptr(G.contiguous().float()), ctx.n, _lib.ptr(gx), _lib.stream_ptr()))
        return gx, None, None


class _StftLossFn(torch.autograd.Function):
    """sum_u weight * l2_comp_stft_summean(a_u, b_u) (reference utils/losses.py:59-64) in ONE library call, gradient w.r.t. whichever side needs it."""

    @staticmethod
    def forward(ctx, a, b, h, weight, sink):
        aa, bb = a.contiguous().float(), b.contiguous().float()
        U, n = aa.shape
        loss = torch.empty(U, device=aa.device)
        ga = torch.empty_like(aa) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(bb) if ctx.needs_input_grad[1] else None
        _lib.check(_lib.require_gpu().buddy_blindop_stft_loss(h, _lib.ptr(aa), _lib.ptr(bb), n, float(weight), _lib.ptr(loss), _lib.ptr(ga), _lib.ptr(gb),
                                                              _lib.stream_ptr()))
        ctx.save_for_backward(*[t for t in (ga, gb) if t is not None])
        ctx.have = (ga is not None, gb is not None)
        if sink is not None:
            sink.last_loss_per_utt = loss
        return loss.sum()

    @staticmethod
    def backward(ctx, gout):
        saved = list(ctx.saved_tensors)
        ga = gout * saved.pop(0) if ctx.have[0] else None
        gb = gout * saved.pop(0) if ctx.have[1] else None
        return ga, gb, None, None, None


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
        self._loss_norm = 0           # member of the l2_comp_stft family the fused calls evaluate (hip_bind): 0 summean
        self._h_epoch = 0             # bumped by every update_H: a saved autograd node of an older H refuses to run its backward
        self._Hr = None               # H of the last autograd-mode update_H (real view, carries the graph to the parameters)
        self._pt = None               # persistent parameter tensors (decay, weights, phases) once somebody asked for ``params`` (see there)
        self._pt_seen = None
        self._lib_newer = True        # the library's copy changed since the persistent tensors were last filled
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

    # The reference keeps its parameters as torch tensors the sampler's Adam holds on to (``Adam(operator.params + operator.params_phases)``,
    # EulerHeunSamplerDPS.py:193) and flips ``requires_grad`` on (:78-81).  Here the parameters live in the library handle; ``params`` /
    # ``params_phases`` hand out PERSISTENT tensors that mirror them: refreshed from the handle when a library call changed it (hip_optimize,
    # project_params, set_params), pushed into the handle by update_H when torch changed them in place (an optimizer step: tensor._version moved).
    def _persistent(self):
        if self._pt is None:
            self._pt = list(self._get())
            self._lib_newer = False
            self._pt_seen = [t._version for t in self._pt]
        elif self._lib_newer:
            with torch.no_grad():
                for t, v in zip(self._pt, self._get()):
                    t.copy_(v)
            self._lib_newer = False
            self._pt_seen = [t._version for t in self._pt]
        return self._pt

    def _torch_side_changed(self):
        return self._pt is not None and not self._lib_newer and [t._version for t in self._pt] != self._pt_seen

    def _push_persistent(self):
        d, w, p = (t.detach().float().contiguous() for t in self._pt)
        _lib.check(_lib.load().buddy_blindop_set_params(self._h, _lib.ptr(d), _lib.ptr(w), _lib.ptr(p), 0, _lib.stream_ptr()))
        self._pt_seen = [t._version for t in self._pt]

    @property
    def params(self):
        return self._persistent()[:2]

    @property
    def params_phases(self):
        return self._persistent()[2:]

    @property
    def H(self):
        if self._Hr is not None:          # autograd mode: the tensor the graph runs through
            return torch.view_as_complex(self._Hr)
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
        if self._torch_side_changed():
            self._push_persistent()       # an in-place torch update that has not reached the handle yet must not be lost under a partial set
        _lib.check(_lib.load().buddy_blindop_set_params(self._h, _lib.ptr(d), _lib.ptr(w), _lib.ptr(p), int(reset_adam), _lib.stream_ptr()))
        if d is not None or w is not None or p is not None:
            self._lib_newer = True

    def update_H(self, rir=None, H=None, use_noise=False, noise=None, phases=None):
        if rir is not None or H is not None:
            raise NotImplementedError("an externally given H / RIR is the informed scenario (RIROperator); the blind operator designs H from its parameters")
        if phases is not None:
            self.set_params(phases=phases)
        if not use_noise and self._pt is not None and any(t.requires_grad for t in self._pt) and torch.is_grad_enabled():
            # the reference's optimize_op (:78-83): parameters flagged requires_grad, then update_H -- H comes out attached to them
            d, w, p = self._persistent()
            self._Hr = _UpdateHFn.apply(d, w, p, self)
            self._pt_seen = [t._version for t in self._pt]
            return
        if self._torch_side_changed():
            self._push_persistent()
        n = None
        if use_noise:
            n = (noise if noise is not None else self._randn((self.length_rir,))).to(self.device).float().contiguous()
        _lib.check(_lib.load().buddy_blindop_update_H(self._h, _lib.ptr(n), _lib.stream_ptr()))
        self._h_epoch += 1
        self._Hr = None
        if use_noise:
            self._lib_newer = True        # phases := angle(H)

    def project_params(self):
        """reference :298-331 on the device-resident parameters (also applied inside buddy_blindop_optimize after every Adam step)"""
        if self._torch_side_changed():
            self._push_persistent()
        _lib.check(_lib.load().buddy_blindop_project(self._h, _lib.stream_ptr()))
        if self._pt is not None:          # the reference projects its tensors in place (:298-331): the persistent mirrors follow
            self._lib_newer = True
            self._persistent()

    def design_filter(self, correct_OLA=True):
        """reference :241-251 -> (U, F, Nf) magnitudes from the current decay / weights"""
        assert correct_OLA
        A = torch.empty(self.U, self.n_fft // 2 + 1, self.Nf, device=self.device)
        _lib.check(_lib.load().buddy_blindop_design_filter(self._h, _lib.ptr(A), _lib.stream_ptr()))
        return A

    def apply_stft(self, x):
        """reference :41-52 for signals of the bound length: (U, F, T) complex64"""
        xx = x.unsqueeze(0) if x.dim() == 1 else x
        n = int(xx.shape[-1])
        if xx.shape[0] != self.U or n not in (self.length, self.length_rir + 1024):
            raise NotImplementedError(f"the HIP operator transforms (U={self.U}, L={self.length}) signals and its own time RIRs "
                                      f"(U, {self.length_rir + 1024}), got {tuple(xx.shape)}")
        T = 1 + (n + self.win_length) // self.hop_length
        return torch.view_as_complex(_StftFn.apply(xx, self._h, T))      # differentiable: the reference's get_loss(...) autograds through apply_stft

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
        xx = x.unsqueeze(0) if squeeze else x
        Hr = self._Hr if (self._Hr is not None and not detach_operator) else None
        y = _DegradeFn.apply(xx, Hr, self)           # a graph w.r.t. x (likelihood score, :61-69) and, in autograd mode, w.r.t. H (optimize_op, :86)
        return y.squeeze(0) if squeeze else y

    def get_time_RIR(self, excitation=None, H=None):
        assert excitation is None and H is None
        out = _TimeRirFn.apply(self._Hr, self)
        return out.squeeze(0) if self.U == 1 else out

    # ---- sampler fast paths ----
    def hip_bind(self, y, ps):
        """cache comp(STFT(y)); read loss weights / compression from the posterior_sampling config"""
        # the regulariser is gated like the reference gates it (EulerHeunSamplerDPS.py:94,200): only loss.name == "none" turns it off;
        # RIR_noise_regularization.use is never read there
        reg_loss = ps.RIR_noise_regularization.loss
        from ...utils.losses import NORM_MODE
        used = (ps.rec_loss, ps.rec_loss_params) + (() if reg_loss.name == "none" else (reg_loss,))
        if any(hasattr(l, "loss_1") or l.name not in NORM_MODE for l in used):
            raise NotImplementedError(f"the HIP operator's fused calls evaluate one member of {sorted(NORM_MODE)} (hybrids: through get_loss(...)(x, x_hat))")
        comps, names = {float(l.compression_factor) for l in used}, {l.name for l in used}
        if len(comps) != 1 or len(names) != 1 or not (0.0 < min(comps) <= 1.0):
            raise NotImplementedError("the HIP operator's fused calls take ONE loss name and ONE compression factor in (0, 1] for the reconstruction, "
                                      "parameter and regulariser terms (the shipped configs: l2_comp_stft_summean @ 0.667 for all three)")
        self._loss_norm = NORM_MODE[names.pop()]
        _lib.check(_lib.load().buddy_blindop_set_loss_norm(self._h, self._loss_norm))
        self.set_compression(comps.pop())
        self.w_rec = float(ps.rec_loss.get("weight", 1.0))
        self.w_rec_params = float(ps.rec_loss_params.get("weight", 1.0))
        self.w_reg = None if reg_loss.name == "none" else float(reg_loss.get("weight", 1.0))
        self.reg = ps.RIR_noise_regularization
        self.hp = ps.blind_hp
        yy = y.contiguous().float()
        _lib.check(_lib.load().buddy_blindop_set_y(self._h, _lib.ptr(yy), _lib.stream_ptr()))
        self._y_bound = yy
        self.set_params(reset_adam=True)          # fresh Adam state, like constructing the optimizer in the reference's predict_conditional (:193)

    def _loss_handle(self, U, n):
        """library handle whose STFT / loss kernels take (U, n) signals (utils.losses.LossSpec.__call__)"""
        if U != self.U or n not in (self.length, self.length_rir + 1024):
            raise NotImplementedError(f"the HIP operator evaluates losses of (U={self.U}, {self.length}) signals and of its time RIRs, got ({U}, {n})")
        return self._h

    def set_compression(self, c):
        """compression exponent of the spectral losses (reference utils/losses.py:60-62: any value in (0, 1])"""
        if abs(float(c) - self._comp_created) > 1e-12:
            _lib.check(_lib.load().buddy_blindop_set_compression(self._h, float(c)))
            self._comp_created = float(c)
            if getattr(self, "_y_bound", None) is not None:       # the cached compressed observation depends on the exponent
                _lib.check(_lib.load().buddy_blindop_set_y(self._h, _lib.ptr(self._y_bound), _lib.stream_ptr()))

    def hip_rec_loss(self, x_den):
        return _HipRecLoss.apply(x_den, self, self.w_rec)

    def hip_rec_loss_grad(self, x_den):
        """d (sum_u w_rec * rec_loss_u) / d x_den straight from the library (no autograd graph); the per-utterance losses land in ``last_rec_per_utt``"""
        x = x_den.contiguous().float()
        loss = torch.empty(self.U, device=x.device)
        g = torch.empty_like(x)
        _lib.check(_lib.load().buddy_blindop_rec_loss_grad(self._h, _lib.ptr(x), float(self.w_rec), _lib.ptr(loss), _lib.ptr(g), _lib.stream_ptr()))
        self.last_rec_per_utt = loss
        return g

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
        if self._torch_side_changed():
            self._push_persistent()
        _lib.check(_lib.load().buddy_blindop_optimize(self._h, _lib.ptr(xd), _lib.ptr(noise), float(t_op), n_it, self.w_rec_params,
                                                      float(self.w_reg or 0.0), float(self.hp.lr_op), float(self.hp.beta1), float(self.hp.beta2),
                                                      float(self.hp.weight_decay), _lib.stream_ptr()))
        self._h_epoch += 1; self._Hr = None; self._lib_newer = True      # parameters and H moved inside the library


