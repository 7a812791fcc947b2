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

    def optim_fwd(self, Xden, Y):
        return torch.sum((self.degradation(Xden) - Y) ** 2)

    def get_time_RIR(self):
        return self.params
