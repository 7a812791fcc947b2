"""``RIROperator`` -- informed time-domain reverberation operator, same surface as reference
``testing/operators/reverb.py:8-88``; ``degradation`` runs the hand-written FIR kernel (``buddy_fir``).
Batched: ``update_params`` accepts one RIR ``(M,)`` (shared) or a list / ``(B,M)`` stack of per-utterance RIRs."""
from __future__ import annotations

import torch

from ...utils import reverb_utils
from .shared import Operator
from ._stft import OperatorSTFT


class RIROperator(Operator, OperatorSTFT):
    def __init__(self, op_hp, time_kernel_size=10, sample_rate=16000, device=None):
        super().__init__()
        self.time_kernel_size = time_kernel_size
        self.params = None
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self._init_stft(op_hp, sample_rate, dev)

    def degradation(self, x, rm_delay=False, **ignored_kwargs):
        assert self.params is not None, "filter is None"
        return reverb_utils.fast_apply_RIR(x, self.params, rm_delay=rm_delay)

    def update_params(self, k, **ignored_kwargs):
        if isinstance(k, (list, tuple)):
            M = max(r.shape[-1] for r in k)
            k = torch.stack([torch.nn.functional.pad(torch.as_tensor(r, dtype=torch.float32), (0, M - r.shape[-1])) for r in k])
        k = torch.as_tensor(k, dtype=torch.float32).to(self.device)
        if self.params is None:
            self.params = torch.nn.Parameter(k, requires_grad=False)
        else:
            self.params.data = k

    # ---- sampler fast path: likelihood loss + analytic gradient in the HIP library (buddy_blindop_fir_loss_grad) ----
    def hip_bind(self, y, ps):
        """True if the HIP likelihood path is usable for this run (CUDA, shipped loss); caches comp(STFT(y)) in a library handle."""
        from .subband_filtering import create_stft_loss_handle
        from ... import _lib
        l, hp = ps.rec_loss, self.op_hp
        from ...utils.losses import NORM_MODE
        if not (y.is_cuda and y.dim() == 2 and not hasattr(l, "loss_1") and l.name in NORM_MODE and 0.0 < float(l.compression_factor) <= 1.0
                and (hp.NFFT, hp.win_length, hp.hop, hp.window) == (1024, 512, 128, "hann") and y.shape[1] >= 1024):
            return False
        h = self._loss_handle(int(y.shape[0]), int(y.shape[1]))
        self._hip_w = float(l.get("weight", 1.0))
        self._comp_created, self._loss_norm = float(l.compression_factor), NORM_MODE[l.name]
        _lib.check(_lib.load().buddy_blindop_set_compression(h, self._comp_created))
        _lib.check(_lib.load().buddy_blindop_set_loss_norm(h, self._loss_norm))
        _lib.check(_lib.load().buddy_blindop_set_y(h, _lib.ptr(y.contiguous().float()), _lib.stream_ptr()))
        return True

    def _loss_handle(self, U, n):
        """library handle (STFT-1024/512/128 + compressed-spectrum loss machinery) for (U, n) signals; one is kept, rebuilt when the shape changes"""
        from .subband_filtering import create_stft_loss_handle
        hp = self.op_hp
        if (hp.NFFT, hp.win_length, hp.hop, hp.window) != (1024, 512, 128, "hann") or n < 1024:
            raise NotImplementedError("the HIP STFT / loss kernels are built for the operator STFT 1024 / 512 / 128 (hann) and signals of >= 1024 samples")
        key = (int(U), int(n))
        if getattr(self, "_hip_key", None) != key:
            self._hip_release()
            self._hip_h = create_stft_loss_handle(self.sample_rate, key[0], key[1])
            self._hip_key = key
        return self._hip_h

    def apply_stft(self, x):
        """reference reverb.py:54-72 -> (U, 513, frames) complex64, differentiable (its adjoint runs in the library): get_loss(...)(y, y_hat) of the
        reference's own utils/losses.py autograds through it"""
        from .subband_filtering import _StftFn
        xx = x.unsqueeze(0) if x.dim() == 1 else x
        h = self._loss_handle(int(xx.shape[0]), int(xx.shape[1]))
        T = 1 + (int(xx.shape[1]) + self.win_length) // self.hop_length
        return torch.view_as_complex(_StftFn.apply(xx, h, T))

    def hip_rec_loss(self, x_den):
        from .subband_filtering import _HipFirRecLoss
        return _HipFirRecLoss.apply(x_den, self, self._hip_w)

    def hip_rec_loss_grad(self, x_den):
        """d (sum_u weight * rec_loss_u) / d x_den straight from the library (no autograd graph)"""
        from ... import _lib
        x = x_den.contiguous().float()
        rir = self.params.detach().contiguous().float()
        M = rir.shape[-1]
        loss = torch.empty(x.shape[0], device=x.device)
        g = torch.empty_like(x)
        _lib.check(_lib.load().buddy_blindop_fir_loss_grad(self._hip_h, _lib.ptr(x), _lib.ptr(rir), 0 if rir.dim() == 1 else M, M, float(self._hip_w),
                                                           _lib.ptr(loss), _lib.ptr(g), _lib.stream_ptr()))
        self.last_rec_per_utt = loss
        return g

    def _hip_release(self):
        try:
            if getattr(self, "_hip_h", None) is not None:
                from ... import _lib
                _lib.load().buddy_blindop_destroy(self._hip_h)
        except Exception:
            pass
        self.__dict__["_hip_h"] = None; self.__dict__["_hip_key"] = None     # not via nn.Module.__setattr__: also runs at interpreter shutdown

    def __del__(self):
        try:
            self._hip_release()
        except Exception:
            pass

    def optim_fwd(self, Xden, Y):
        return torch.sum((self.degradation(Xden) - Y) ** 2)

    def get_time_RIR(self):
        return self.params
