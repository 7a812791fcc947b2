"""GPU: the WPE warm start (`wpe_scaled`, reference testing/EulerHeunSamplerDPS.py:32-54) through the C-ABI (`buddy_wpe_dereverb`: hand-written
complex128 STFT -> WPE iterations -> iSTFT) against oracle/wpe_ref.py -- numpy, written independently from nara_wpe's published algorithm
(that package is absent: parity with nara_wpe itself stays unpinned) -- and the SHIPPED blind configuration (conf/tester/
blind_dereverberation_BUDDy.yaml: wpe_scaled, T = 201, 10 operator updates) for three steps against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _reverberant(u, L, taps):
    from buddy_amd.synth import synth_clean, synth_rir
    c = synth_clean(u, L).astype(np.float64)
    return np.convolve(0.05 * c / c.std(), synth_rir(u, taps).astype(np.float64))[:L].astype(np.float32)


@pytest.mark.parametrize("L,B,taps", [(64000, 2, 50), (12345, 3, 50), (700, 1, 3)])
def test_wpe_dereverb_vs_oracle(L, B, taps):
    """whole warm-start estimate, every utterance on its own; ragged lengths (not a multiple of the shift, shorter than two frames).
    50 taps on reverberant speech-like input: cond(R) ~ 1e9...1e10, Cholesky (kernel) vs LU (numpy) agree to cond * eps_fp64.  (The 700-sample
    clip has 11 frames: with more taps than frames R is singular and nara_wpe itself leaves the documented path for a least-squares fallback.)"""
    from buddy_amd.utils.wpe import wpe_dereverb
    from oracle.wpe_ref import wpe_warm_start_estimate
    y = np.stack([_reverberant(u, L, min(8000, L // 2)) for u in range(B)])
    out = wpe_dereverb(torch.from_numpy(y).cuda(), taps=taps, delay=2, iterations=5).cpu().numpy()
    assert out.shape == (B, L) and np.isfinite(out).all()
    for b in range(B):
        ref = wpe_warm_start_estimate(y[b:b + 1], taps=taps, delay=2, iterations=5)
        e = rel(out[b], ref[0])
        print(f"L={L} utterance {b}: buddy_wpe_dereverb vs oracle rel {e:.2e}")
        assert e < 1e-4, (L, b, e)
    # other filter orders / delays / iteration counts (few taps: well conditioned, agreement at float32 output precision)
    t2 = min(7, taps)
    out = wpe_dereverb(torch.from_numpy(y[:1]).cuda(), taps=t2, delay=3, iterations=2).cpu().numpy()
    assert rel(out[0], wpe_warm_start_estimate(y[:1], taps=t2, delay=3, iterations=2)[0]) < 1e-6
    out = wpe_dereverb(torch.from_numpy(y[:1]).cuda(), taps=5, delay=1, iterations=0).cpu().numpy()       # no iterations: istft(stft(y)) = y
    assert rel(out[0], y[0]) < 1e-6


def _stack(T, nf, L, B, seeds, updates=None):
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    ov = [f"tester.sampling_params.T={T}", f"network.nf={nf}"]           # everything else as shipped: wpe_scaled, 10 updates, order 1
    if updates is not None:
        ov.append(f"tester.posterior_sampling.blind_hp.op_updates_per_step={updates}")
    args = compose(tester="blind_dereverberation_BUDDy", overrides=ov)
    assert args.tester.posterior_sampling.warm_initialization.mode == "wpe_scaled"
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, nf).items()})
    net = net.cuda().eval()
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    ns = [NoiseStream(400 + s) for s in seeds]
    t.sampler.noise = ns
    items = [(synth_clean(s, L), synth_rir(s, 8000), f"u{s}.wav") for s in seeds]
    seg, y, op, _ = t.prepare_batch(items, blind=True, noise=ns)
    return args, t, ns, items, y, op


def _oracle(args, nf, item, seed, steps):
    """B = 1 oracle run of `steps` steps of the shipped configuration; returns (x0, [x_den per step])"""
    from buddy_amd.synth import synth_state_dict
    from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S
    torch.set_num_threads(32)
    P = ncsnpp_ref.to_torch(synth_state_dict(0, nf))
    onet = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
    nr = S.NoiseStream(400 + seed)
    ref = S.EulerHeunDPSRef(onet, S.EDMRef(args.diff_params.sde_hp), args, nr)
    op_hp = args.tester.informed_dereverberation.op_hp
    oo = O.RIROperatorRef(op_hp); oo.update_params(torch.from_numpy(item[1]))
    c0 = torch.from_numpy(item[0]); c0 = 0.05 * c0 / c0.std()
    y0 = oo.degradation(c0[None])
    bo = O.BlindSubbandFilteringRef(op_hp, 16000, nr); bo.update_H(use_noise=True, noise=nr)
    ref.operator, ref.y = bo, y0
    ps = args.tester.posterior_sampling
    ref.rec_loss = O.get_loss_ref(ps.rec_loss, bo); ref.rec_loss_params = O.get_loss_ref(ps.rec_loss_params, bo)
    ref.rir_reg_loss = O.get_loss_ref(ps.RIR_noise_regularization.loss, bo)
    ref.optim = torch.optim.Adam(bo.params + bo.params_phases, lr=ps.blind_hp.lr_op, betas=(ps.blind_hp.beta1, ps.blind_hp.beta2))
    ts = S.create_schedule(ref.sde_hp, ref.T); gm = S.get_gamma(ts, ref.sp)
    x = ref.initialize_x(y0.shape, ts)
    x0, tr = x.clone(), []
    for i in range(steps):
        x, xd = ref.step(x, ts[i], ts[i + 1], gm[i], True)
        tr.append(xd[0].clone())
    return x0, tr, nr.k


def test_initialize_x_wpe_scaled_vs_oracle():
    """the warm start itself: scaling_factor * x_wpe / std(x_wpe) + t_0 * noise, per utterance of a batch == the oracle's B = 1 value"""
    from oracle import sampler_ref as S
    L, seeds = 64000, [0, 5, 2]
    args, t, ns, items, y, op = _stack(201, 32, L, 3, seeds)
    smp = t.sampler
    smp.operator, smp.y = op, y
    x = smp.initialize_x(tuple(y.shape), "cuda", smp.create_schedule())
    for b, s in enumerate(seeds):
        k0 = ns[b].k
        x0, _, _ = _oracle(args, 32, items[b], s, 0)
        e = rel(x[b].cpu().numpy(), x0[0].numpy())
        print(f"initialize_x(wpe_scaled) utterance {s}: rel {e:.2e}")
        assert e < 1e-4


def test_shipped_blind_config_three_steps_vs_oracle():
    """conf/tester/blind_dereverberation_BUDDy.yaml exactly as shipped (wpe_scaled warm start, T = 201 schedule, order 1, 10 operator
    updates per step, speech-magnitude constraint; reference test_blind_dereverberation.sh:18), full-width network, 4 s utterances, B = 2:
    three diffusion steps of the second utterance against the oracle's single-utterance run with the same noise draws.  Ten scale-free Adam
    updates per step amplify fp32 round-off by tens of dB per step, so the agreement is asserted per step with the
    measured decay, the first step (no feedback yet) at round-off level."""
    from buddy_amd.utils.metrics import si_sdr
    L, seeds, steps = 64000, [3, 4], 3
    args, t, ns, items, y, op = _stack(201, 128, L, 2, seeds)
    assert args.tester.posterior_sampling.blind_hp.op_updates_per_step == 10 and args.tester.sampling_params.order == 1
    smp = t.sampler
    smp.bind(y, op, True)
    sched = smp.create_schedule()
    tl, gl = sched.tolist(), smp.get_gamma(sched).tolist()
    x = smp.initialize_x(tuple(y.shape), "cuda", sched)
    tr = []
    for i in range(steps):
        x, xd = smp.step(x, tl[i], tl[i + 1], gl[i], blind=True)
        tr.append(xd.cpu())
    assert torch.isfinite(x).all()
    _, tro, k = _oracle(args, 128, items[1], seeds[1], steps)
    assert k == ns[1].k, "noise streams out of step"
    sd = [float(si_sdr(tr[i][1:2], tro[i][None])) for i in range(steps)]
    print("shipped blind config (wpe_scaled, T=201, 10 updates), full size: per-step SI-SDR(build; oracle) dB:", [round(v, 1) for v in sd])
    assert sd[0] > 80.0 and sd[1] > 35.0 and sd[2] > 15.0, sd


def test_shipped_blind_config_to_the_end_sanity():
    """The shipped configuration run THROUGH: conf/tester/blind_dereverberation_BUDDy.yaml untouched -- wpe_scaled, all
    T = 201 steps, order 1, 10 operator updates per step -- full-width network, two 4 s utterances through Sampler.predict_conditional.
    The chain is chaotic (no sample-wise reference exists after a few steps), so this is the sanity gate a harness run
    needs: finite everywhere, every estimate at the level the speech-magnitude constraint pins (std = speech_scaling), the operator
    parameters inside their projection box, the estimated RIR finite with its direct path in place, the noise streams fully consumed in
    step.  The measured comparison against the float64 arbiter and the fp32 oracles is profiles/archive/r04_shipped_T201.json (tools/shipped_run.py)."""
    import time
    L, seeds = 64000, [3, 4]
    args, t, ns, items, y, op = _stack(201, 128, L, 2, seeds)
    ps = args.tester.posterior_sampling
    assert args.tester.sampling_params.T == 201 and ps.blind_hp.op_updates_per_step == 10 and ps.constraint_speech_magnitude.use
    torch.cuda.synchronize()
    t0 = time.time()
    pred = t.sampler.predict_conditional(y, op, shape=(2, L), blind=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    assert pred.shape == (2, L) and torch.isfinite(pred).all()
    lvl = pred.std(dim=1).cpu()
    assert torch.allclose(lvl, torch.full((2,), float(ps.constraint_speech_magnitude.speech_scaling)), rtol=1e-3), lvl
    d, w = (p.cpu() for p in op.params)
    assert torch.isfinite(d).all() and torch.isfinite(w).all()
    assert float(d.min()) >= op.min_decay * (1 - 1e-6) and float(d.max()) <= op.max_decay * (1 + 1e-6)
    assert float(w.min()) >= 10 ** (op.Amin / 20) * (1 - 1e-6) and float(w.max()) <= 10 ** (op.Amax / 20) * (1 + 1e-6)
    rir = op.get_time_RIR().cpu()
    assert torch.isfinite(rir).all() and float(rir.abs().max()) > 0
    assert ns[0].k == ns[1].k and ns[0].k > 201 * 11          # one churn draw + ten regulariser draws per step, plus the initial ones
    print(f"shipped blind config, T=201, B=2: {wall:.2f} s wall incl. WPE warm start ({wall / 201 * 1e3:.1f} ms/step); levels {lvl.tolist()}")
