"""Sampler base -- same surface as reference ``testing/Sampler.py:5-86``."""
from __future__ import annotations

import abc

import torch


class Sampler:
    def __init__(self, model, diff_params, args):
        self.model = model.eval()
        self.diff_params = diff_params
        self.args = args
        if self.args.tester.sampling_params.same_as_training:
            self.sde_hp = diff_params.sde_hp
        else:
            self.sde_hp = self.args.tester.sampling_params.sde_hp
        self.T = self.args.tester.sampling_params.T
        self.step_counter = 0
        # optional injected noise: list (one per utterance) of objects with randn(shape) -> CPU tensor.  None = torch RNG,
        # drawn on the CPU generator and moved, like the reference (EulerHeunSampler.py:21,43).
        self.noise = None

    def _randn(self, shape, device):
        """(B, L) standard normal; with injected streams utterance b draws its own (1, L) in reference call order.
        Drawn on the CPU generator like the reference (same values for the same seed).  On a GPU the host copy is pinned and the transfer
        asynchronous in stream order: a pageable ``.to(device)`` makes the host wait for everything queued before it, i.e. drains the GPU at
        every step boundary (measured ~1 ms of a 107 ms step: copy + relaunch latency with an empty queue)."""
        cuda = torch.device(device).type == "cuda"
        if self.noise is None:
            n = torch.randn(shape, pin_memory=cuda)
        else:
            assert len(self.noise) == shape[0], "one noise stream per utterance"
            n = torch.cat([s.randn((1,) + tuple(shape[1:])) for s in self.noise], dim=0)
            if n.is_cuda:                # streams that draw on the device already (the float64 arbiter runs of the tests)
                return n.to(device)
            if cuda:
                n = n.pin_memory()
        if not cuda:
            return n.to(device)
        # The transfer runs on a side stream (round 6): the host is a step ahead of the GPU, so the 2 MB cross PCIe while the previous step's
        # kernels still run instead of sitting in the compute stream between two of them (a blit kernel of ~130 us per step); the compute stream
        # only waits for the copy's event.
        cur = torch.cuda.current_stream(device)
        side = self._copy_streams.get((str(device), cur.cuda_stream)) if hasattr(self, "_copy_streams") else None
        if side is None:
            if not hasattr(self, "_copy_streams"):
                self._copy_streams = {}
            side = self._copy_streams[(str(device), cur.cuda_stream)] = torch.cuda.Stream(device)
        with torch.cuda.stream(side):
            d = n.to(device, non_blocking=True)
        cur.wait_stream(side)
        d.record_stream(cur)
        return d

    @abc.abstractmethod
    def predict(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def predict_unconditional(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def predict_conditional(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def step(self, *args, **kwargs):
        pass

    def create_schedule(self, sigma_min=None, sigma_max=None, rho=None, T=None):
        """EDM rho-schedule exactly as the reference builds it (Sampler.py:39-56): a/(T-1) with a = 0..T, t[T] = 0."""
        sigma_min = self.sde_hp.sigma_min if sigma_min is None else sigma_min
        sigma_max = self.sde_hp.sigma_max if sigma_max is None else sigma_max
        rho = self.sde_hp.rho if rho is None else rho
        T = self.T if T is None else T
        if self.args.tester.sampling_params.schedule == "edm":
            a = torch.arange(0, T + 1)
            t = (sigma_max ** (1 / rho) + a / (T - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
            t[-1] = 0
            return t
        raise NotImplementedError(f"schedule {self.args.tester.sampling_params.schedule} not implemented")

    def Tweedie2score(self, tweedie, xt, t):
        return self.diff_params.Tweedie2score(tweedie, xt, t)

    def get_Tweedie_estimate(self, x, t_i):
        return self.diff_params.denoiser(x.unsqueeze(1), self.model, t_i).squeeze(1)


class NoSampler(Sampler):
    def predict(self, *args, **kwargs):
        return None

    def predict_unconditional(self, *args, **kwargs):
        return None

    def predict_conditional(self, *args, **kwargs):
        return None

    def step(self, *args, **kwargs):
        return None
