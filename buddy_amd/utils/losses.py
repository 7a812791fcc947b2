"""Reconstruction losses -- same factory surface as reference ``utils/losses.py:17-95`` (``get_loss(loss_args, operator)`` -> ``loss(x, x_hat)``).

The product evaluates the loss INSIDE ``libbuddy_hip.so``: the sampler's fast path fuses it with the operator (``buddy_blindop_rec_loss_grad`` /
``buddy_blindop_fir_loss_grad`` / ``buddy_blindop_optimize``: STFT-1024/512/128, compressed-spectrum difference, frame mean, analytic adjoints), and --
round 6 -- the object ``get_loss`` returns is CALLABLE like the reference's: ``loss(x, x_hat)`` is one library call (``buddy_blindop_stft_loss``) behind a
``torch.autograd.Function``, differentiable w.r.t. either argument, so the reference's own ``get_likelihood_score`` / ``optimize_op``
(``testing/EulerHeunSamplerDPS.py:61-113``) run on it unmodified.  Supported: the shipped ``l2_comp_stft_summean`` (reference ``losses.py:59-64``) with any
compression factor in (0, 1]; other names raise ``NotImplementedError`` -- there is no torch-op evaluation path in the product (the formulas as torch
expressions live in ``oracle/batched/losses.py``).  Per-utterance semantics: the library returns one loss per utterance (``operator.last_loss_per_utt``)
and the call returns their sum, so gradients decouple per utterance."""
from __future__ import annotations

SUPPORTED = ("l2_comp_stft_summean",)
COMPRESSION = 0.667          # the shipped configs' factor (conf/tester/*.yaml)


class LossSpec:
    """A validated loss block: name, weight, compression factor -- and, bound to an operator, the loss itself: ``spec(x, x_hat)``."""

    def __init__(self, name, weight, compression_factor, operator=None):
        self.name, self.weight, self.compression_factor, self.operator = name, float(weight), float(compression_factor), operator

    def __call__(self, x, x_hat):
        op = self.operator
        if op is None or not hasattr(op, "_loss_handle"):
            raise NotImplementedError(f"loss '{self.name}' needs an operator with a HIP loss handle (get_loss(loss_args, operator=...))")
        from .. import _lib
        from ..testing.operators.subband_filtering import _StftLossFn
        a = x.unsqueeze(0) if x.dim() == 1 else x
        b = x_hat.unsqueeze(0) if x_hat.dim() == 1 else x_hat
        if a.shape != b.shape or a.dim() != 2 or not a.is_cuda:
            raise NotImplementedError(f"loss '{self.name}': two (U, L) GPU tensors of one shape expected, got {tuple(x.shape)} and {tuple(x_hat.shape)}")
        h = op._loss_handle(int(a.shape[0]), int(a.shape[1]))
        if hasattr(op, "set_compression"):
            op.set_compression(self.compression_factor)
        else:
            _lib.check(_lib.load().buddy_blindop_set_compression(h, self.compression_factor))
        return _StftLossFn.apply(a, b, h, self.weight, op)

    def __repr__(self):
        return f"LossSpec({self.name!r}, weight={self.weight}, compression_factor={self.compression_factor})"


def get_loss(loss_args, operator=None):
    if loss_args.name == "none":
        return None
    if hasattr(loss_args, "loss_1"):
        raise NotImplementedError("hybrid losses (loss_1, loss_2, ...) are not built into the HIP operator")
    name = loss_args.name
    if name not in SUPPORTED:
        raise NotImplementedError(f"rec_loss {name} not implemented in the HIP operator (supported: {SUPPORTED})")
    c = loss_args.get("compression_factor", None)
    if c is None or not (0.0 < float(c) <= 1.0):
        raise NotImplementedError(f"compression_factor {c}: the reference asserts 0 < factor <= 1 (utils/losses.py:60)")
    if loss_args.get("freq_weighting", None) is not None:
        raise NotImplementedError("freq_weighting is not built into the HIP loss (the shipped configs never set this key, appendix B.8)")
    return LossSpec(name, loss_args.get("weight", 1.0), c, operator)
