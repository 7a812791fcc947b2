"""SI-SDR (not in the reference): 10 log10(||a s||^2 / ||a s - s_hat||^2), a = <s_hat,s>/||s||^2, zero-mean."""
import torch


def si_sdr(est, ref):
    est = est.double() - est.double().mean(-1, keepdim=True)
    ref = ref.double() - ref.double().mean(-1, keepdim=True)
    a = (est * ref).sum(-1, keepdim=True) / (ref * ref).sum(-1, keepdim=True).clamp_min(1e-30)
    t = a * ref
    return 10 * torch.log10((t * t).sum(-1) / ((t - est) ** 2).sum(-1).clamp_min(1e-30))
