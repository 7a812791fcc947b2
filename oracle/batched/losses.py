"""TEST INFRASTRUCTURE, not product: the reconstruction-loss formulas of reference ``utils/losses.py:17-95`` as torch expressions over
``operator.apply_stft`` (autograd), batched per utterance.  The product evaluates its one supported loss (``l2_comp_stft_summean``,
compression 0.667) and its gradient inside ``libbuddy_hip.so``; these formulas serve the CPU host-logic tests and the on-GPU cross-checks."""
from __future__ import annotations

import torch


def get_frequency_weighting(freqs, freq_weighting=None):     # reference losses.py:3-14
    if freq_weighting is None:
        return torch.ones_like(freqs)
    if freq_weighting == "sqrt":
        return torch.sqrt(freqs)
    if freq_weighting == "exp":
        freqs = torch.exp(freqs)
        return freqs - freqs[:, 0, :].unsqueeze(-2)
    if freq_weighting == "log":
        return torch.log(1 + freqs)
    if freq_weighting == "linear":
        return freqs
    raise NotImplementedError(freq_weighting)


def _comp(X, c):
    return (X.abs() + 1e-8) ** c * torch.exp(1j * X.angle())


def get_loss(loss_args, operator=None):
    if loss_args.name == "none":
        return None
    if hasattr(loss_args, "loss_1"):       # hybrid of several losses (losses.py:22-23)
        fns = [get_loss(getattr(loss_args, key), operator=operator) for key in list(loss_args.keys())]
        return lambda x, x_hat, per_utt=False: sum(f(x, x_hat, per_utt) for f in fns)
    name = loss_args.name
    weight = loss_args.get("weight", 1.0)
    if "stft" in name:
        def loss_fn(x, x_hat, per_utt=False):
            X = operator.apply_stft(x)
            X_hat = operator.apply_stft(x_hat)
            fw = loss_args.get("freq_weighting", None)   # NB key name: configs say frequency_weighting -> always None (appendix B.8)
            if fw is not None:
                freqs = torch.linspace(0, 1, X.shape[-2], device=X.device).unsqueeze(-1).unsqueeze(0).expand(X.shape) + 1
                w = get_frequency_weighting(freqs, fw)
                X, X_hat = X * w, X_hat * w
            if name == "l2_stft_sum":
                v = ((X - X_hat).abs() ** 2).sum(dim=(-2, -1))
            elif name == "l2_stft_mag_sum":
                v = ((X.abs() - X_hat.abs()) ** 2).sum(dim=(-2, -1))
            elif name == "l2_stft_logmag_sum":
                v = ((torch.log10(X.abs() + 1e-8) - torch.log10(X_hat.abs() + 1e-8)) ** 2).sum(dim=(-2, -1))
            elif name in ("l2_comp_stft_sum", "l2_comp_stft_mean", "l2_comp_stft_summean"):
                c = loss_args.get("compression_factor", None)
                assert c is not None and 0.0 < c <= 1.0, f"Compression factor weird: {c}"
                d = (_comp(X, c) - _comp(X_hat, c)).abs() ** 2
                if name == "l2_comp_stft_sum":
                    v = d.sum(dim=(-2, -1))
                elif name == "l2_comp_stft_mean":
                    v = d.mean(dim=(-2, -1))
                else:
                    v = d.sum(dim=-2).mean(dim=-1)      # mean over frames of the per-frame sum over bins (losses.py:59-64)
            elif name == "l2_log_stft_sum":
                v = ((torch.log(1 + X.abs()) * torch.exp(1j * X.angle()) - torch.log(1 + X_hat.abs()) * torch.exp(1j * X_hat.angle())).abs() ** 2).sum(dim=(-2, -1))
            else:
                raise NotImplementedError(f"rec_loss {name} not implemented")
            v = weight * v
            return v if per_utt else v.sum()
        return loss_fn
    if name == "l2_sum":
        return lambda x, x_hat, per_utt=False: (lambda v: v if per_utt else v.sum())(weight * ((x - x_hat) ** 2).sum(-1))
    if name == "l2_mean":
        return lambda x, x_hat, per_utt=False: (lambda v: v if per_utt else v.sum())(weight * ((x - x_hat) ** 2).mean(-1))
    raise NotImplementedError(f"rec_loss {name} not implemented")
