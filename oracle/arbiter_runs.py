"""One full blind DPS run of the CPU oracle, in fp32 (a given intra-op thread count) or in the fp64 arbiter mode.
TEST INFRASTRUCTURE (see ``oracle/__init__.py``, ``oracle/precision.py``); used by tools/arbiter.py and the -m gpu arbiter test."""
from __future__ import annotations

import contextlib

import torch

from . import ncsnpp_ref, operators_ref as O, precision, sampler_ref as S


@contextlib.contextmanager
def _on_device(device):
    """fp32 oracle with another default device (``"cuda"``: the same restated algorithm through torch's GPU kernels -- rocFFT / MIOpen / ATen:
    one more fp32 execution with its own summation orders, 50 full-size steps in seconds instead of minutes); ``None`` leaves the CPU"""
    if device is None:
        yield
        return
    # torch's GPU kernels are not run-to-run reproducible by default (atomic accumulation orders): two float64 executions of the informed T = 50
    # chain came out 74 dB apart (round 5), a noise floor right where the build is measured (69-72 dB).  The arbiter runs with torch's deterministic
    # algorithms: the float64 run is then bit-reproducible and the comparison means something.
    prev = torch.get_default_device()
    det, cd, cb = torch.are_deterministic_algorithms_enabled(), torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    warn = torch.is_deterministic_algorithms_warn_only_enabled()
    torch.set_default_device(device)
    torch.use_deterministic_algorithms(True, warn_only=True); torch.backends.cudnn.deterministic = True; torch.backends.cudnn.benchmark = False
    try:
        yield
    finally:
        torch.set_default_device(prev)
        torch.use_deterministic_algorithms(det, warn_only=warn); torch.backends.cudnn.deterministic = cd; torch.backends.cudnn.benchmark = cb


def overrides(T, updates, nf):
    return [f"tester.sampling_params.T={T}", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
            f"tester.posterior_sampling.blind_hp.op_updates_per_step={updates}", f"network.nf={nf}"]


def run_blind(seed, L, T, nf, updates, rir_taps, fp64=False, threads=8, weight_seed=0, device=None, perturb=0.0):
    """utterance ``seed`` (buddy_amd.synth clean/RIR), noise stream 9000 + seed -> (x_den per step (T, L) float32, clean (L,), n_draws)"""
    from buddy_amd.config import compose
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        args = compose(overrides=overrides(T, updates, nf))
        with (precision.fp64(device) if fp64 else _on_device(device)):
            dt, dev = torch.get_default_dtype(), torch.get_default_device()
            P = ncsnpp_ref.to_torch(synth_state_dict(weight_seed, nf))
            net = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
            c0 = torch.from_numpy(synth_clean(seed, L)).to(device=dev, dtype=dt)
            c0 = 0.05 * c0 / c0.std()
            if perturb:                                   # sensitivity probe: the same run with the input scaled by (1 + perturb)
                c0 = c0 * (1.0 + perturb)
            nr = S.NoiseStream(9000 + seed)
            ref = S.EulerHeunDPSRef(net, S.EDMRef(args.diff_params.sde_hp), args, nr)
            op_hp = args.tester.informed_dereverberation.op_hp
            oo = O.RIROperatorRef(op_hp)
            oo.update_params(torch.from_numpy(synth_rir(seed, rir_taps)).to(device=dev, dtype=dt))
            y0 = oo.degradation(c0[None])
            bo = O.BlindSubbandFilteringRef(op_hp, 16000, nr)
            bo.update_H(use_noise=True, noise=nr)
            tr = []
            ref.predict_conditional(y0, bo, shape=(1, L), blind=True, trace=tr)
        return torch.stack([t[1][0] for t in tr]).float().cpu(), c0.float().cpu(), nr.k
    finally:
        torch.set_num_threads(prev)


def run_blind_batched(seeds, L, T, nf, updates, rir_taps, fp64=False, weight_seed=0, device="cuda", perturb=0.0):
    """ALL the utterances ``seeds`` as ONE batch (round 6: the population gate samples 8 seeds; eight B = 1 float64 runs take eight minutes):
    the batched torch-op sampler / operator of oracle/batched (row b == the B = 1 oracle run of utterance b: tests/test_host_logic.py::
    test_batched_blind_dps_equals_oracle_per_utterance) around the oracle network, in float64 (the arbiter) or float32 (one more fp32
    execution through torch's GPU kernels), noise stream 9000 + seed per utterance -> (x_den per step (T, U, L) float32, clean (U, L), n_draws per utterance)"""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from .batched.operators import BlindSubbandFiltering
    from .batched.sampler import EulerHeunSamplerDPSTorch
    args = compose(overrides=overrides(T, updates, nf))
    with (precision.fp64(device) if fp64 else _on_device(device)), (_on_device(device) if fp64 else contextlib.nullcontext()):
        dt, dev = torch.get_default_dtype(), torch.get_default_device()
        P = ncsnpp_ref.to_torch(synth_state_dict(weight_seed, nf))

        class _OracleNet(torch.nn.Module):              # the product sampler's constructor calls model.eval()
            def forward(self, z, cn):
                return ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
        net = _OracleNet()
        op_hp = args.tester.informed_dereverberation.op_hp
        cs, ys = [], []
        for s in seeds:
            c0 = torch.from_numpy(synth_clean(s, L)).to(device=dev, dtype=dt)
            c0 = 0.05 * c0 / c0.std()
            if perturb:
                c0 = c0 * (1.0 + perturb)
            oo = O.RIROperatorRef(op_hp)
            oo.update_params(torch.from_numpy(synth_rir(s, rir_taps)).to(device=dev, dtype=dt))
            cs.append(c0); ys.append(oo.degradation(c0[None])[0])
        y0 = torch.stack(ys)
        ns = [S.NoiseStream(9000 + s) for s in seeds]
        smp = EulerHeunSamplerDPSTorch(net, instantiate(args.diff_params), args)
        smp.use_hip_update = False              # the elementwise tail as torch expressions in the run's dtype (the HIP kernels are fp32)
        smp.noise = ns
        bo = BlindSubbandFiltering(op_hp, 16000, num_utts=len(seeds), noise=ns, device=str(dev))
        bo.update_H(use_noise=True)
        smp.bind(y0, bo, True)
        sched = smp.create_schedule()
        tl, gl = sched.tolist(), smp.get_gamma(sched).tolist()
        x = smp.initialize_x(tuple(y0.shape), dev, sched)
        tr = []
        for i in range(T):
            x, xd = smp.step(x, tl[i], tl[i + 1], gl[i], blind=True)
            tr.append(xd.detach().float().cpu())
    return torch.stack(tr), torch.stack(cs).float().cpu(), [n.k for n in ns]


def run_informed(seed, L, T, nf, rir_taps, fp64=False, threads=8, weight_seed=0, device=None, order=2):
    """the non-chaotic chain: informed DPS (known RIR, no operator optimisation), order 2 -> (x_den per step (T, L) float32, clean, n_draws)"""
    from buddy_amd.config import compose
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        args = compose(tester="informed_dereverberation_DPS", overrides=[f"tester.sampling_params.T={T}", f"network.nf={nf}",
                                                                         f"tester.sampling_params.order={order}"])
        with (precision.fp64(device) if fp64 else contextlib.nullcontext()):
            dt, dev = torch.get_default_dtype(), torch.get_default_device()
            P = ncsnpp_ref.to_torch(synth_state_dict(weight_seed, nf))
            net = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
            c0 = torch.from_numpy(synth_clean(seed, L)).to(device=dev, dtype=dt)
            c0 = 0.05 * c0 / c0.std()
            nr = S.NoiseStream(9000 + seed)
            ref = S.EulerHeunDPSRef(net, S.EDMRef(args.diff_params.sde_hp), args, nr)
            oo = O.RIROperatorRef(args.tester.informed_dereverberation.op_hp)
            oo.update_params(torch.from_numpy(synth_rir(seed, rir_taps)).to(device=dev, dtype=dt))
            y0 = oo.degradation(c0[None])
            tr = []
            ref.predict_conditional(y0, oo, shape=(1, L), blind=False, trace=tr)
        return torch.stack([t[1][0] for t in tr]).float().cpu(), c0.float().cpu(), nr.k
    finally:
        torch.set_num_threads(prev)


def denoiser_eval(seed, L, nf, sigma, fp64=False, threads=8, weight_seed=0, device=None):
    """ONE denoiser evaluation D(x; sigma) and its input-VJP for a fixed cotangent (seeded) -> (D, J^T w) float64 on the CPU"""
    import numpy as np
    from buddy_amd.config import compose
    from buddy_amd.synth import synth_state_dict
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        args = compose(overrides=[f"network.nf={nf}"])
        with (precision.fp64(device) if fp64 else contextlib.nullcontext()):
            dt, dev = torch.get_default_dtype(), torch.get_default_device()
            P = ncsnpp_ref.to_torch(synth_state_dict(weight_seed, nf))
            net = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
            rs = np.random.RandomState(77 + seed)
            x = torch.from_numpy((sigma * rs.standard_normal((1, L))).astype(np.float32)).to(device=dev, dtype=dt).requires_grad_(True)
            w = torch.from_numpy(rs.standard_normal((1, L)).astype(np.float32)).to(device=dev, dtype=dt)
            edm = S.EDMRef(args.diff_params.sde_hp)
            d = edm.denoiser(x.unsqueeze(1), net, torch.tensor(float(sigma), dtype=dt, device=dev)).squeeze(1)
            g, = torch.autograd.grad(d, x, w)
        return d.detach().double().cpu(), g.double().cpu(), x.detach().float().cpu(), w.float().cpu()
    finally:
        torch.set_num_threads(prev)
