"""Reconstruction losses -- same factory surface as reference ``utils/losses.py:17-95`` (``get_loss(loss_args, operator)``).

The product evaluates the likelihood loss and its gradient INSIDE ``libbuddy_hip.so`` (``buddy_blindop_rec_loss_grad`` /
``buddy_blindop_fir_loss_grad`` / ``buddy_blindop_optimize``: STFT-1024/512/128, compressed-spectrum difference, frame mean, analytic
adjoints), so ``get_loss`` here resolves and VALIDATES a loss block of the config and returns its specification; the one supported loss is
the shipped ``l2_comp_stft_summean`` with compression factor 0.667 (``conf/tester/*.yaml``; reference ``losses.py:59-64``).  Anything else
raises ``NotImplementedError`` -- there is no torch-op evaluation path in the product (the formulas as torch expressions live in
``oracle/batched/losses.py``).  Per-utterance semantics: the library returns one loss per utterance and their sum, so gradients decouple
per utterance."""
from __future__ import annotations

SUPPORTED = ("l2_comp_stft_summean",)
COMPRESSION = 0.667


class LossSpec:
    """A validated loss block: name, weight, compression factor.  Calling it is an error: the value is computed by the library call the
    operator makes (``hip_rec_loss`` / ``hip_optimize``), never by torch ops."""

    def __init__(self, name, weight, compression_factor):
        self.name, self.weight, self.compression_factor = name, float(weight), float(compression_factor)

    def __call__(self, *a, **k):
        raise NotImplementedError(f"loss '{self.name}' is evaluated inside libbuddy_hip.so (operator.hip_rec_loss / hip_optimize), not through get_loss()()")

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
    if c is None or abs(float(c) - COMPRESSION) > 1e-9:
        raise NotImplementedError(f"compression_factor {c}: the HIP loss kernels are built for {COMPRESSION}")
    if loss_args.get("freq_weighting", None) is not None:
        raise NotImplementedError("freq_weighting is not built into the HIP loss (the shipped configs never set this key, appendix B.8)")
    return LossSpec(name, loss_args.get("weight", 1.0), c)
