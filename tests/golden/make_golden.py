#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by importing the *reference* (read-only, from
/root/reference) in the build container.  The reference never travels to the GPU box; only the small
.npz vectors written here do (inputs + expected outputs -- no reference source).

Recipe: stub the third-party modules the hot path imports but does not need
(soundfile, wandb, torchaudio, nara_wpe, hydra, omegaconf, plotly, pandas, matplotlib), shim ``torchcde``
(linear interpolation; absent third-party, parity unpinned), patch ``torch.randn / randn_like / rand`` to
the deterministic ``NoiseStream`` so sampler runs can be replayed, load configs with the repo's own
YAML loader, use ``buddy_amd.synth`` seeded weights (the reference default init zeroes half the network).

Usage:  python tests/golden/make_golden.py [--only NAME ...]
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(8)


def _install_stubs():
    for name in ["soundfile", "wandb", "torchaudio", "nara_wpe", "nara_wpe.wpe", "nara_wpe.utils", "hydra",
                 "hydra.utils", "omegaconf", "plotly", "plotly.express", "plotly.graph_objects", "pandas",
                 "matplotlib", "matplotlib.pyplot"]:
        import importlib.util
        try:
            present = importlib.util.find_spec(name) is not None
        except (ImportError, ValueError):
            present = False
        if name not in sys.modules and not present:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    import torch._dynamo  # noqa: F401  (torch.optim imports it lazily; must happen before stubs confuse find_spec)
    sys.modules["nara_wpe.wpe"].wpe = None
    sys.modules["nara_wpe.utils"].stft = None
    sys.modules["nara_wpe.utils"].istft = None
    cde = types.ModuleType("torchcde")

    def linear_interpolation_coeffs(x):
        return x

    class LinearInterpolation:
        def __init__(self, coeffs, t):
            self.c, self.t = coeffs, t

        def evaluate(self, q):
            K = self.t.shape[0]
            idx = (torch.bucketize(q, self.t) - 1).clamp(0, K - 2)
            t0, t1 = self.t[idx], self.t[idx + 1]
            frac = ((q - t0) / (t1 - t0)).unsqueeze(-1)
            return self.c[..., idx, :] + frac * (self.c[..., idx + 1, :] - self.c[..., idx, :])

    cde.linear_interpolation_coeffs = linear_interpolation_coeffs
    cde.LinearInterpolation = LinearInterpolation
    sys.modules["torchcde"] = cde


_install_stubs()

from buddy_amd.config import compose, load_yaml, AttrDict, to_attrdict  # noqa: E402
from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir  # noqa: E402
from oracle.sampler_ref import NoiseStream  # noqa: E402  (only the deterministic noise stream)


class patched_noise:
    """Route torch.randn / randn_like / rand through a NoiseStream inside the reference."""

    def __init__(self, stream):
        self.s = stream

    def __enter__(self):
        self.orig = (torch.randn, torch.randn_like, torch.rand)
        s = self.s

        def randn(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            return s.randn(shape)

        def randn_like(t, **kw):
            return s.randn(tuple(t.shape))

        def rand(*shape, **kw):
            if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
                shape = tuple(shape[0])
            return s.rand(shape)

        torch.randn, torch.randn_like, torch.rand = randn, randn_like, rand
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like, torch.rand = self.orig


def build_ref_net(nf, n_fft, hop, seed, ch_mult=(1, 2, 2, 2), num_res_blocks=1):
    from networks.ncsnpp import NCSNppTime
    cfg = load_yaml(os.path.join(ROOT, "conf/network/ncsnpp.yaml"))
    cfg.pop("_target_")
    cfg.update(nf=nf, ch_mult=list(ch_mult), num_res_blocks=num_res_blocks,
               stft=AttrDict(n_fft=n_fft, hop_length=hop, center=True))
    net = NCSNppTime(**cfg)
    sd = synth_state_dict(seed, nf, tuple(ch_mult), num_res_blocks)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.eval()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------------------
def gen_edm_sched():
    from diff_params.edm import EDM
    from testing.EulerHeunSampler import EulerHeunSampler
    out = {}
    for tester in ["blind_dereverberation_BUDDy", "informed_dereverberation_DPS", "only_unconditional"]:
        for T in (10, 50, 201):
            args = compose(tester=tester, overrides=[f"tester.sampling_params.T={T}"])
            edm = EDM(args.diff_params.type, args.diff_params.sde_hp)
            s = EulerHeunSampler(torch.nn.Identity(), edm, args)
            t = s.create_schedule()
            out[f"{tester}.T{T}.t"] = t
            out[f"{tester}.T{T}.gamma"] = s.get_gamma(t)
    args = compose()
    edm = EDM(args.diff_params.type, args.diff_params.sde_hp)
    sig = torch.tensor([1e-4, 3e-3, 0.05, 0.5, 7.0])
    out["sigma"] = sig
    out["cskip"], out["cout"], out["cin"], out["cnoise"] = edm.cskip(sig), edm.cout(sig), edm.cin(sig), edm.cnoise(sig)
    save("edm_sched", **out)


def _net_fixture(name, nf, n_fft, hop, L, B, seed, with_taps, ch_mult=(1, 2, 2, 2), num_res_blocks=1, first_row_of=None):
    net = build_ref_net(nf, n_fft, hop, seed, ch_mult, num_res_blocks)
    rs = np.random.RandomState(seed + 100)
    x = torch.from_numpy((0.5 * rs.standard_normal((B, 1, L))).astype(np.float32))
    cn = torch.from_numpy(rs.uniform(-2.0, 0.3, size=(B,)).astype(np.float32))
    cot = torch.from_numpy(rs.standard_normal((B, 1, L)).astype(np.float32))
    if first_row_of is not None:      # row 0 = the B = 1 fixture's utterance (row independence: its results must not depend on the batch it is in)
        f1 = np.load(os.path.join(HERE, first_row_of + ".npz"))
        x[0], cn[0], cot[0] = torch.from_numpy(f1["x"][0]), torch.from_numpy(f1["cnoise"])[0], torch.from_numpy(f1["cot"][0])
    x.requires_grad_(True)
    taps = {}
    hooks = []
    if with_taps:
        from networks.ncsnpp_utils import layerspp
        for i, mod in enumerate(net.all_modules):
            if isinstance(mod, (layerspp.ResnetBlockBigGANpp, layerspp.AttnBlockpp)):
                hooks.append(mod.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(i, o.detach())))
    y = net(x, cn)
    g, = torch.autograd.grad(y, x, cot)
    for h in hooks:
        h.remove()
    arrs = dict(x=x.detach(), cnoise=cn, cot=cot, y=y.detach(), vjp=g,
                meta=np.array([nf, n_fft, hop, L, B, seed]))
    if tuple(ch_mult) != (1, 2, 2, 2) or num_res_blocks != 1:       # the shipped fixtures keep their exact bytes
        arrs["ch_mult"] = np.array(ch_mult)
        arrs["num_res_blocks"] = np.array(num_res_blocks)
    for i, t in taps.items():
        arrs[f"tap{i}_mean"] = t.mean()
        arrs[f"tap{i}_absmax"] = t.abs().max()
        arrs[f"tap{i}_std"] = t.std()
    save(name, **arrs)


def gen_net_small():
    _net_fixture("net_small", nf=32, n_fft=126, hop=32, L=4096, B=2, seed=3, with_taps=True)


def gen_net_arch():
    """the constructor is parametric (reference networks/ncsnpp.py:50-52,184-270): two members of the family besides the shipped (1,2,2,2) / 1 block"""
    _net_fixture("net_cm12_rb2", nf=32, n_fft=126, hop=32, L=4096, B=2, seed=21, with_taps=True, ch_mult=(1, 2), num_res_blocks=2)
    _net_fixture("net_cm1122_rb1", nf=32, n_fft=126, hop=32, L=4096, B=2, seed=22, with_taps=True, ch_mult=(1, 1, 2, 2), num_res_blocks=1)


def gen_net_full():
    _net_fixture("net_full", nf=128, n_fft=510, hop=128, L=16000, B=1, seed=5, with_taps=True)


def gen_net_full_64000():
    """the FULL size pinned against the reference itself (networks/ncsnpp.py:281-449, :498-506) -- nf = 128, STFT 510 / 128,
    L = 64 000 (BASELINE configs[1]'s utterance length): input, cnoise, cotangent, output, input-VJP and the per-module statistics; and a B = 2
    batch whose row 0 is that same utterance (row independence at the full size)."""
    _net_fixture("net_full_64000", nf=128, n_fft=510, hop=128, L=64000, B=1, seed=6, with_taps=True)
    _net_fixture("net_full_64000_B2", nf=128, n_fft=510, hop=128, L=64000, B=2, seed=6, with_taps=False, first_row_of="net_full_64000")


def gen_ops():
    import utils.reverb_utils as ru
    from utils.losses import get_loss
    from testing.operators.reverb import RIROperator
    from testing.operators.subband_filtering import BlindSubbandFiltering
    args = compose()
    op_hp = args.tester.informed_dereverberation.op_hp
    L = 16000
    x = torch.from_numpy(synth_clean(0, L))
    rir = torch.from_numpy(synth_rir(0, taps=3000))
    out = dict(x=x, rir=rir)
    # informed operator
    op = RIROperator(op_hp, time_kernel_size=rir.shape[-1], sample_rate=16000)
    op.update_params(rir)
    y = op.degradation(x[None])
    out["y_rir"] = y
    out["apply_stft_y"] = torch.view_as_real(op.apply_stft(y))
    loss = get_loss(args.tester.posterior_sampling.rec_loss, operator=op)
    xd = (x[None] * 0.9 + 0.01 * torch.from_numpy(synth_clean(1, L))[None]).requires_grad_(True)
    val = loss(y, op.degradation(xd))
    out["inf_loss"] = val.detach()
    out["inf_xd"] = xd.detach()
    out["inf_loss_grad"] = torch.autograd.grad(val, xd)[0]
    # DSP utils
    h = torch.from_numpy(synth_rir(2, taps=1500))
    out["minphase_in"] = h
    out["minphase_out"] = ru.minimum_phase_version(h)
    # blind operator
    ns = NoiseStream(11)
    with patched_noise(ns):
        bop = BlindSubbandFiltering(op_hp, 16000)
        bop.update_H(use_noise=True)
    out["blind_A"] = bop.design_filter().detach()
    out["blind_H"] = torch.view_as_real(bop.H.detach())
    out["blind_phases"] = bop.params_phases[0].detach()
    out["blind_deg"] = bop.degradation(x[None]).detach()
    out["blind_rir"] = bop.get_time_RIR().detach()
    out["blind_stft_x"] = torch.view_as_real(bop.apply_stft(x[None]))
    # one gradient of (rec_loss_params + RIR-noise reg) wrt params, as optimize_op forms it
    ps = args.tester.posterior_sampling
    lp = get_loss(ps.rec_loss_params, operator=bop)
    lr = get_loss(ps.RIR_noise_regularization.loss, operator=bop)
    for p in bop.params + bop.params_phases:
        p.requires_grad = True
    bop.update_H()
    l1 = lp(y, bop.degradation(x[None]))
    rt = bop.get_time_RIR()
    n = ns.randn(rt.shape)
    l2 = lr(rt, (rt + 0.005 * n).detach())
    gs = torch.autograd.grad(l1 + l2, bop.params + bop.params_phases)
    out["blind_l_rec"], out["blind_l_reg"] = l1.detach(), l2.detach()
    out["blind_g_decay"], out["blind_g_weights"], out["blind_g_phases"] = gs
    # project_params on out-of-range values
    with torch.no_grad():
        bop.params[0].copy_(torch.linspace(0.0, 0.8, 25)[None])
        bop.params[1].copy_(torch.linspace(0.2, 150.0, 25)[None])
    bop.project_params()
    out["proj_decay"], out["proj_weights"] = bop.params[0].detach(), bop.params[1].detach()
    save("ops", **out)


def _e2e(name, tester, blind, T, order, overrides=(), L=8192, nf=32, seed=7, utt=0):
    from diff_params.edm import EDM
    from testing.EulerHeunSamplerDPS import EulerHeunSamplerDPS
    from testing.operators.reverb import RIROperator
    from testing.operators.subband_filtering import BlindSubbandFiltering
    ov = [f"tester.sampling_params.T={T}", f"tester.sampling_params.order={order}"] + list(overrides)
    args = compose(tester=tester, overrides=ov)
    net = build_ref_net(nf, 510, 128, seed)
    edm = EDM(args.diff_params.type, args.diff_params.sde_hp)
    sampler = EulerHeunSamplerDPS(net, edm, args)
    clean = torch.from_numpy(synth_clean(utt, L))
    rir = torch.from_numpy(synth_rir(utt, taps=2000))
    op_hp = args.tester.informed_dereverberation.op_hp
    ns = NoiseStream(1000 + utt)
    import tqdm as _tq
    import testing.EulerHeunSamplerDPS as M
    M.tqdm = lambda it, *a, **k: it
    with patched_noise(ns):
        with torch.no_grad():
            op_ref = RIROperator(op_hp, time_kernel_size=rir.shape[-1], sample_rate=16000)
            op_ref.update_params(rir)
            y = op_ref.degradation(clean[None])
            if blind:
                op = BlindSubbandFiltering(op_hp, sample_rate=16000)
                op.update_H(use_noise=True)
        pred = sampler.predict_conditional(y, op if blind else op_ref, shape=(1, L), blind=blind)
    arrs = dict(clean=clean, rir=rir, y=y, pred=pred, n_draws=ns.k,
                meta=np.array([nf, L, T, order, seed, utt, 1000 + utt]))
    if blind:
        arrs["est_rir"] = sampler.operator.get_time_RIR().detach()
        arrs["decay"], arrs["weights"] = op.params[0].detach(), op.params[1].detach()
        arrs["phases"] = op.params_phases[0].detach()
    save(name, **arrs)


def gen_e2e_informed():
    _e2e("e2e_informed", "informed_dereverberation_DPS", blind=False, T=4, order=2)


def gen_e2e_blind():
    _e2e("e2e_blind", "blind_dereverberation_BUDDy", blind=True, T=3, order=1,
         overrides=["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                    "tester.posterior_sampling.blind_hp.op_updates_per_step=3"])


def gen_uncond():
    from diff_params.edm import EDM
    from testing.EulerHeunSampler import EulerHeunSampler
    import testing.EulerHeunSampler as M
    M.tqdm = lambda it, *a, **k: it
    args = compose(tester="only_unconditional", overrides=["tester.sampling_params.T=3"])
    net = build_ref_net(32, 510, 128, 7)
    edm = EDM(args.diff_params.type, args.diff_params.sde_hp)
    s = EulerHeunSampler(net, edm, args)
    ns = NoiseStream(2000)
    with patched_noise(ns):
        x = s.predict_unconditional((2, 8192), "cpu")
    save("e2e_uncond", pred=x, n_draws=ns.k, meta=np.array([32, 8192, 3, 2, 7, 2000]))


def gen_opt():
    """optimize_op itself (reference EulerHeunSamplerDPS.py:71-113): parameters and torch.optim.Adam state after ONE full iteration
    (update_H, both losses, backward, Adam step, projection), after three and after the shipped ten, plus the minimum-phase projection at the
    size cons() uses (12 928 samples) and the filter the next update_H builds from the updated parameters."""
    import utils.reverb_utils as ru
    from diff_params.edm import EDM
    from utils.losses import get_loss
    from testing.EulerHeunSamplerDPS import EulerHeunSamplerDPS
    from testing.operators.reverb import RIROperator
    from testing.operators.subband_filtering import BlindSubbandFiltering
    args = compose(overrides=["tester.posterior_sampling.blind_hp.op_updates_per_step=1"])
    ps = args.tester.posterior_sampling
    op_hp = args.tester.informed_dereverberation.op_hp
    L = 16000
    x = torch.from_numpy(synth_clean(0, L))
    rir = torch.from_numpy(synth_rir(0, taps=3000))
    op = RIROperator(op_hp, time_kernel_size=rir.shape[-1], sample_rate=16000)
    op.update_params(rir)
    y = op.degradation(x[None])
    x_den = (x[None] * 0.9 + 0.01 * torch.from_numpy(synth_clean(1, L))[None])
    out = dict(y=y, x_den=x_den, t=np.float32(0.02), meta=np.array([L, 21]))
    ns = NoiseStream(21)
    smp = EulerHeunSamplerDPS(torch.nn.Identity(), EDM(args.diff_params.type, args.diff_params.sde_hp), args)
    with patched_noise(ns):
        with torch.no_grad():
            bop = BlindSubbandFiltering(op_hp, 16000)
            bop.update_H(use_noise=True)
        out["phases0"] = bop.params_phases[0].detach().clone()
        smp.operator, smp.y = bop, y
        smp.rec_loss_params = get_loss(ps.rec_loss_params, operator=bop)
        smp.RIR_noise_regularization_loss = get_loss(ps.RIR_noise_regularization.loss, operator=bop)
        smp.optimizer_operator = torch.optim.Adam(bop.params + bop.params_phases, lr=ps.blind_hp.lr_op, weight_decay=ps.blind_hp.weight_decay,
                                                  betas=(ps.blind_hp.beta1, ps.blind_hp.beta2))
        t = torch.tensor(0.02)

        def snap(tag):
            out[f"{tag}_decay"], out[f"{tag}_weights"] = bop.params[0].detach().clone(), bop.params[1].detach().clone()
            out[f"{tag}_phases"] = bop.params_phases[0].detach().clone()
            st = smp.optimizer_operator.state
            for nm, p in (("decay", bop.params[0]), ("weights", bop.params[1]), ("phases", bop.params_phases[0])):
                out[f"{tag}_m_{nm}"], out[f"{tag}_v_{nm}"] = st[p]["exp_avg"].clone(), st[p]["exp_avg_sq"].clone()
            out[f"{tag}_H_stale"] = torch.view_as_real(bop.H.detach().clone())       # H built from the parameters BEFORE the last step
            out[f"{tag}_n_draws"] = ns.k

        smp.optimize_op(x_den.clone(), t)
        snap("it1")
        args.tester.posterior_sampling.blind_hp.op_updates_per_step = 2
        smp.optimize_op(x_den.clone(), t)
        snap("it3")
        args.tester.posterior_sampling.blind_hp.op_updates_per_step = 7
        smp.optimize_op(x_den.clone(), t)
        snap("it10")
        with torch.no_grad():
            bop.update_H()
        out["it10_H"] = torch.view_as_real(bop.H.detach())
        out["it10_rir"] = bop.get_time_RIR().detach()
    h = torch.from_numpy(synth_rir(2, taps=1500))
    hp = torch.nn.functional.pad(h * torch.exp(-torch.arange(1500) / 400.0), (0, 12928 - 1500))
    out["minphase_in"], out["minphase_out"] = hp, ru.minimum_phase_version(hp)
    save("opt", **out)


def gen_e2e_blind_o2():
    """blind, second-order (Heun) with the speech-magnitude constraint on: the corrector evaluation does NOT rescale x_den
    (reference EulerHeunSamplerDPS.py:139-149)."""
    _e2e("e2e_blind_o2", "blind_dereverberation_BUDDy", blind=True, T=3, order=2,
         overrides=["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                    "tester.posterior_sampling.blind_hp.op_updates_per_step=2"], utt=3)


def gen_e2e_blind10():
    """blind, T=10 with the shipped 10 operator updates per step (conf/tester/blind_dereverberation_BUDDy.yaml:72)."""
    _e2e("e2e_blind10", "blind_dereverberation_BUDDy", blind=True, T=10, order=1,
         overrides=["tester.posterior_sampling.warm_initialization.mode=reverb_scaled"], utt=5)


def gen_e2e_blind_full():
    """The blind sampler at the FULL size, recorded from the reference itself (round 6; the other blind fixtures are nf = 32, L = 8 192): nf = 128,
    L = 64 000 = BASELINE configs[1]'s utterance, order 1, a T = 3 schedule with 3 operator updates per step like e2e_blind (nine Adam updates: with the
    shipped ten per step the chain amplifies fp32 round-off by ~75 dB per step and two fp32 executions are 4e-2 apart after three steps -- measured on the
    first version of this fixture -- so nothing but the reference's own kernels could be held to it)."""
    _e2e("e2e_blind_full", "blind_dereverberation_BUDDy", blind=True, T=3, order=1,
         overrides=["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                    "tester.posterior_sampling.blind_hp.op_updates_per_step=3"], L=64000, nf=128, seed=0, utt=2)


def gen_config1():
    """BASELINE config 1: audio_examples/clean/p226/p226_003.wav + its RIR, informed DPS, order 2, T=10 (reference testing/tester.py:123-153,
    conf/tester/informed_dereverberation_DPS.yaml), full-width network (nf=128) on seeded weights.  The clip (133 829 samples: not a
    hop multiple; 8 083-tap RIR after the direct-path trim) goes through the reference's own preprocessing (datasets/vctk.py:209-214,
    tester.py:134-135)."""
    from scipy.io import wavfile
    from diff_params.edm import EDM
    from testing.EulerHeunSamplerDPS import EulerHeunSamplerDPS
    from testing.operators.reverb import RIROperator
    import testing.EulerHeunSamplerDPS as M
    M.tqdm = lambda it, *a, **k: it
    fs, d = wavfile.read(os.path.join(REF, "audio_examples/clean/p226/p226_003.wav"))
    fs2, r = wavfile.read(os.path.join(REF, "audio_examples/rir/p226/p226_003.wav"))
    assert fs == 16000 and fs2 == 16000
    conv = lambda a: a.astype(np.float64) / 32768.0 if a.dtype == np.int16 else a.astype(np.float64)     # soundfile's PCM16 scaling
    data, data_rir = conv(d), conv(r)
    data_rir = data_rir[np.argmax(np.abs(data_rir)):]
    data_rir = data_rir / np.abs(data_rir).max()
    args = compose(tester="informed_dereverberation_DPS", overrides=["tester.sampling_params.T=10"])
    assert args.tester.sampling_params.order == 2
    nf, seed, nseed = 128, 9, 4242
    net = build_ref_net(nf, 510, 128, seed)
    edm = EDM(args.diff_params.type, args.diff_params.sde_hp)
    sampler = EulerHeunSamplerDPS(net, edm, args)
    seg = torch.from_numpy(data).float()
    seg = args.tester.posterior_sampling.warm_initialization.scaling_factor * seg / seg.std()
    RIR = torch.Tensor(data_rir)
    ns = NoiseStream(nseed)
    with patched_noise(ns):
        with torch.no_grad():
            op = RIROperator(args.tester.informed_dereverberation.op_hp, time_kernel_size=RIR.shape[-1], sample_rate=16000)
            op.update_params(RIR)
            y = op.degradation(seg.unsqueeze(0))
        pred = sampler.predict_conditional(y, op, shape=(1, seg.shape[-1]), blind=False)
    save("config1", clean_raw=data.astype(np.float32), rir=data_rir.astype(np.float32), seg=seg, y=y, pred=pred, n_draws=ns.k,
         meta=np.array([nf, seg.shape[-1], 10, 2, seed, nseed, RIR.shape[-1]]))


def _enable_reference_fir():
    """The reference's fir=True path calls ``upfirdn2d`` (networks/ncsnpp_utils/up_or_down_sampling.py:220-257), but the import of
    it is commented out there (:10) and ``op/upfirdn2d.py`` JIT-compiles a CUDA extension on import.  Its pure-PyTorch twin
    ``upfirdn2d_native`` (op/upfirdn2d.py:171-215) is what the reference's own dispatcher runs on CPU (:145-156): lift exactly that function
    out of the file (no extension build) and bind the dispatcher's CPU branch under the missing name."""
    import ast
    import torch.nn.functional as F
    src = open(os.path.join(REF, "networks/ncsnpp_utils/op/upfirdn2d.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "upfirdn2d_native"][0]
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "upfirdn2d_native", "exec"), ns)
    native = ns["upfirdn2d_native"]
    from networks.ncsnpp_utils import up_or_down_sampling as U
    U.upfirdn2d = lambda input, kernel, up=1, down=1, pad=(0, 0): native(input, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return U


def gen_fir():
    """fir=True resampling: upsample_2d / downsample_2d with the (1,3,3,1) kernel and their input gradients, and a small
    network built with fir=True (forward, input-VJP, per-module statistics)."""
    U = _enable_reference_fir()
    rs = np.random.RandomState(77)
    x = torch.from_numpy(rs.standard_normal((2, 6, 10, 12)).astype(np.float32)).requires_grad_(True)
    k = (1, 3, 3, 1)
    up, dn = U.upsample_2d(x, k, factor=2), U.downsample_2d(x, k, factor=2)
    cu = torch.from_numpy(rs.standard_normal(tuple(up.shape)).astype(np.float32))
    cd = torch.from_numpy(rs.standard_normal(tuple(dn.shape)).astype(np.float32))
    gu, = torch.autograd.grad(up, x, cu, retain_graph=True)
    gd, = torch.autograd.grad(dn, x, cd)
    save("fir_ops", x=x.detach(), up=up.detach(), down=dn.detach(), cot_up=cu, cot_down=cd, vjp_up=gu, vjp_down=gd)
    # network with fir=True
    from networks.ncsnpp import NCSNppTime
    cfg = load_yaml(os.path.join(ROOT, "conf/network/ncsnpp.yaml"))
    cfg.pop("_target_")
    nf, n_fft, hop, L, B, seed = 32, 126, 32, 4096, 2, 13
    cfg.update(nf=nf, fir=True, stft=AttrDict(n_fft=n_fft, hop_length=hop, center=True))
    net = NCSNppTime(**cfg)
    sd = synth_state_dict(seed, nf)
    net.load_state_dict({k_: torch.from_numpy(v) for k_, v in sd.items()}, strict=True)
    net.eval()
    rs = np.random.RandomState(seed + 100)
    xi = torch.from_numpy((0.5 * rs.standard_normal((B, 1, L))).astype(np.float32)).requires_grad_(True)
    cn = torch.from_numpy(rs.uniform(-2.0, 0.3, size=(B,)).astype(np.float32))
    cot = torch.from_numpy(rs.standard_normal((B, 1, L)).astype(np.float32))
    taps, hooks = {}, []
    from networks.ncsnpp_utils import layerspp
    for i, mod in enumerate(net.all_modules):
        if isinstance(mod, (layerspp.ResnetBlockBigGANpp, layerspp.AttnBlockpp)):
            hooks.append(mod.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(i, o.detach())))
    y = net(xi, cn)
    g, = torch.autograd.grad(y, xi, cot)
    arrs = dict(x=xi.detach(), cnoise=cn, cot=cot, y=y.detach(), vjp=g, meta=np.array([nf, n_fft, hop, L, B, seed]))
    for i, t in taps.items():
        arrs[f"tap{i}_absmax"], arrs[f"tap{i}_std"] = t.abs().max(), t.std()
    save("net_small_fir", **arrs)


GENS = dict(edm_sched=gen_edm_sched, net_small=gen_net_small, net_arch=gen_net_arch, net_full=gen_net_full, net_full_64000=gen_net_full_64000, ops=gen_ops,
            e2e_informed=gen_e2e_informed, e2e_blind=gen_e2e_blind, e2e_uncond=gen_uncond,
            opt=gen_opt, e2e_blind_o2=gen_e2e_blind_o2, e2e_blind10=gen_e2e_blind10, e2e_blind_full=gen_e2e_blind_full, config1=gen_config1, fir=gen_fir)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    for k, fn in GENS.items():
        if a.only is None or k in a.only:
            print("==", k)
            fn()
