"""Reconstruction losses -- same factory surface as reference ``utils/losses.py:17-95`` (``get_loss(loss_args, operator)`` -> ``loss(x, x_hat)``).

The product evaluates the loss INSIDE ``libbuddy_hip.so``: the sampler's fast path fuses it with the operator (``buddy_blindop_rec_loss_grad`` /
``buddy_blindop_fir_loss_grad`` / ``buddy_blindop_optimize``: STFT-1024/512/128, compressed-spectrum difference, frame mean, analytic adjoints), and --
round 6 -- the object ``get_loss`` returns is CALLABLE like the reference's: ``loss(x, x_hat)`` is one library call (``buddy_blindop_stft_loss``) behind a
``torch.autograd.Function``, differentiable w.r.t. either argument, so the reference's own ``get_likelihood_score`` / ``optimize_op``
(``testing/EulerHeunSamplerDPS.py:61-113``) run on it unmodified.  Supported: the compressed-spectrum family ``l2_comp_stft_summean`` (the shipped one,
reference ``losses.py:59-64``), ``l2_comp_stft_sum`` (``:46-50``) and ``l2_comp_stft_mean`` (``:52-57``) with any compression factor in (0, 1], and hybrids of
them (``loss_1``, ``loss_2``, ...: ``:22-23``) through the callable; other names raise ``NotImplementedError`` -- there is no torch-op evaluation path in the product (the formulas as torch
expressions live in ``oracle/batched/losses.py``).  Per-utterance semantics: the library returns one loss per utterance (``operator.last_loss_per_utt``)
and the call returns their sum, so gradients decouple per utterance."""
from __future__ import annotations

SUPPORTED = ("l2_comp_stft_summean", "l2_comp_stft_sum", "l2_comp_stft_mean")
NORM_MODE = {"l2_comp_stft_summean": 0, "l2_comp_stft_sum": 1, "l2_comp_stft_mean": 2}
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
        lib = _lib.load()
        # the handle's exponent / normalisation for THIS call, then back to what the operator's fused calls (hip_rec_loss / hip_optimize) were bound with
        _lib.check(lib.buddy_blindop_set_compression(h, self.compression_factor))
        _lib.check(lib.buddy_blindop_set_loss_norm(h, NORM_MODE[self.name]))
        try:
            return _StftLossFn.apply(a, b, h, self.weight, op)
        finally:
            _lib.check(lib.buddy_blindop_set_compression(h, float(getattr(op, "_comp_created", COMPRESSION))))
            _lib.check(lib.buddy_blindop_set_loss_norm(h, int(getattr(op, "_loss_norm", 0))))

    def __repr__(self):
        return f"LossSpec({self.name!r}, weight={self.weight}, compression_factor={self.compression_factor})"


class HybridLoss:
    """sum of member losses (reference utils/losses.py:22-23); callable like them.  The operator's FUSED calls take one member only and refuse this."""

    def __init__(self, parts):
        self.parts, self.name = parts, "hybrid"

    def __call__(self, x, x_hat):
        out = self.parts[0](x, x_hat)
        for p in self.parts[1:]:
            out = out + p(x, x_hat)
        return out


def get_loss(loss_args, operator=None):
    if loss_args.name == "none":
        return None
    if hasattr(loss_args, "loss_1"):        # a hybrid: the sum of its members (reference :22-23), through the callable only
        parts = [get_loss(loss_args[k], operator=operator) for k in loss_args.keys() if str(k).startswith("loss_")]
        return HybridLoss([p for p in parts if p is not None])
    name = loss_args.name
    if name not in SUPPORTED:
        raise NotImplementedError(f"rec_loss {name} not implemented in the HIP operator (supported: {SUPPORTED})")
    c = loss_args.get("compression_factor", None)
    if c is None or not (0.0 < float(c) <= 1.0):
        raise NotImplementedError(f"compression_factor {c}: the reference asserts 0 < factor <= 1 (utils/losses.py:60)")
    if loss_args.get("freq_weighting", None) is not None:
        raise NotImplementedError("freq_weighting is not built into the HIP loss (the shipped configs never set this key, appendix B.8)")
    return LossSpec(name, loss_args.get("weight", 1.0), c, operator)
