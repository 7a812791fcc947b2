"""GPU end-to-end parity: the drop-in sampler stack (EDM + EulerHeun/DPS + operators) driving the HIP score network
against fixtures recorded from the reference on the same weights, inputs and injected noise draws.
Stated tolerance: relative-to-absmax 3e-3 on the sampler output over T=3..4 chained guided steps (fp32; the guidance
normalises by ||grad||, so round-off in the VJP is amplified by zeta/||g||), and |SI-SDR(build; reference)| > 40 dB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _setup(g, tester, extra=()):
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict
    meta = [int(v) for v in g["meta"]]
    nf, L, T, order, seed = meta[:5]
    args = compose(tester=tester, overrides=[f"tester.sampling_params.T={T}", f"tester.sampling_params.order={order}",
                                             f"network.nf={nf}"] + list(extra))
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(seed, nf).items()})
    net = net.cuda().eval()
    edm = instantiate(args.diff_params)
    return args, net, edm, meta


def _sisdr(a, b):
    from buddy_amd.utils.metrics import si_sdr
    return float(si_sdr(torch.as_tensor(a).reshape(1, -1), torch.as_tensor(b).reshape(1, -1)))


def test_informed_dps_vs_reference_fixture(golden):
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    g = golden("e2e_informed")
    args, net, edm, meta = _setup(g, "informed_dereverberation_DPS")
    nseed = meta[6]
    tester = Tester(args, net, edm, test_set=None, device="cuda", in_training=True)
    tester.sampler.noise = [NoiseStream(nseed)]
    seg, y, op, _ = tester.prepare_batch([(g["clean"], g["rir"], "u.wav")], blind=False)
    # the harness normalises the clean signal; the fixture's `clean` is already normalised -> compare y up to that scale
    sf = 0.05 / torch.from_numpy(g["clean"]).std()
    assert rel((y[0] / sf).cpu().numpy(), g["y"][0]) < 2e-5
    y_ref = torch.from_numpy(g["y"]).cuda()
    pred = tester.sampler.predict_conditional(y_ref, op, shape=(1, meta[1]), blind=False)
    assert tester.sampler.noise[0].k == int(g["n_draws"])
    p = pred.cpu().numpy()
    assert rel(p, g["pred"]) < 3e-3
    assert _sisdr(p, g["pred"]) > 40.0
    # north-star form of the tolerance: SI-SDR w.r.t. the clean signal differs by < 0.1 dB between build and reference
    assert abs(_sisdr(p, g["clean"]) - _sisdr(g["pred"], g["clean"])) < 0.1


def _run_blind(g, extra, backend):
    from buddy_amd.instantiate import instantiate
    from oracle.sampler_ref import NoiseStream
    args, net, edm, meta = _setup(g, "blind_dereverberation_BUDDy", extra)
    ns = [NoiseStream(meta[6])]
    if backend == "hip":                 # the product: sampler from the config's _target_, HIP operator
        from buddy_amd.testing.operators.subband_filtering import BlindSubbandFiltering
        smp = instantiate(args.tester.sampler, net, edm, args)
        op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, 16000, num_utts=1, noise=ns, device="cuda", length=meta[1])
        assert hasattr(op, "hip_optimize")
    else:                                # cross-check: the torch-op restatement (oracle/batched) around the same HIP network
        from oracle.batched.operators import BlindSubbandFiltering
        from oracle.batched.sampler import EulerHeunSamplerDPSTorch
        smp = EulerHeunSamplerDPSTorch(net, edm, args)
        op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, 16000, num_utts=1, noise=ns, device="cuda")
    smp.noise = ns
    op.update_H(use_noise=True)
    y = torch.from_numpy(g["y"]).cuda()
    pred = smp.predict_conditional(y, op, shape=(1, meta[1]), blind=True)
    assert ns[0].k == int(g["n_draws"])
    return pred.cpu().numpy(), op, smp


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_blind_dps_vs_reference_fixture(golden, backend):
    """backend "hip": the hand-written operator (csrc/operator.hip -- what the bench and the Tester run); "torch": the torch-op restatement
    of oracle/batched (test infrastructure) driving the same HIP network."""
    g = golden("e2e_blind")
    p, op, smp = _run_blind(g, ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                "tester.posterior_sampling.blind_hp.op_updates_per_step=3"], backend)
    assert rel(p, g["pred"]) < 3e-3
    assert _sisdr(p, g["pred"]) > 40.0
    assert abs(_sisdr(p, g["clean"]) - _sisdr(g["pred"], g["clean"])) < 0.1
    # operator parameters after 9 Adam updates: Adam's m/sqrt(v) is scale-free, so fp32 FFT round-off differences (25856-point
    # transforms inside the min-phase projection) move individual bands by ~1 %
    assert rel(op.params[0][0].detach().cpu().numpy(), g["decay"]) < 3e-2
    assert rel(op.params[1][0].detach().cpu().numpy(), g["weights"]) < 3e-2
    assert rel(smp.operator.get_time_RIR().detach().cpu().numpy(), g["est_rir"]) < 3e-2


def test_blind_dps_full_size_vs_reference_fixture(golden):
    """round 6: the blind sampler at the FULL size against the reference's own run, not only against the oracle: nf = 128,
    L = 64 000, 3 operator updates per step, T = 3 (e2e_blind_full.npz recorded by tests/golden/make_golden.py from
    testing/EulerHeunSamplerDPS.py:115-204)."""
    g = golden("e2e_blind_full")
    p, op, smp = _run_blind(g, ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                "tester.posterior_sampling.blind_hp.op_updates_per_step=3"], "hip")
    print(f"full-size blind run vs the reference: rel {rel(p, g['pred']):.2e}, SI-SDR {_sisdr(p, g['pred']):.1f} dB, decay {rel(op.params[0][0].detach().cpu().numpy(), g['decay']):.1e}")
    assert rel(p, g["pred"]) < 3e-3
    assert _sisdr(p, g["pred"]) > 40.0
    assert abs(_sisdr(p, g["clean"]) - _sisdr(g["pred"], g["clean"])) < 0.1
    assert rel(op.params[0][0].detach().cpu().numpy(), g["decay"]) < 5e-2
    assert rel(smp.operator.get_time_RIR().detach().cpu().numpy(), g["est_rir"]) < 5e-2


def test_reference_control_flow_on_product_operator_and_network(golden):
    """VERDICT r5 item 5 / SURVEY 8(b): the operator protocol holds under the REFERENCE's sampler, not only under the product's.  The torch-loop
    restatement of the reference's control flow (oracle/sampler_ref.EulerHeunDPSRef = testing/EulerHeunSamplerDPS.py:56-157 line for line: autograd
    through operator.degradation and get_loss(...)(y, y_hat) for the likelihood score, requires_grad -> update_H -> degradation / get_time_RIR ->
    loss.backward() -> torch.optim.Adam.step() -> project_params for the operator) drives the PRODUCT operator (degradation / update_H / get_time_RIR /
    apply_stft as autograd Functions over the library's VJP entries, persistent parameter tensors), the PRODUCT loss (utils.losses.get_loss: one library
    call, differentiable) and the PRODUCT network, and reproduces the reference's own run (e2e_blind.npz) at the tolerance of the product sampler."""
    import oracle.sampler_ref as S
    from oracle.arbiter_runs import _on_device
    from buddy_amd.testing.operators.subband_filtering import BlindSubbandFiltering
    from buddy_amd.utils.losses import get_loss
    g = golden("e2e_blind")
    args, net, edm, meta = _setup(g, "blind_dereverberation_BUDDy", ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                                                    "tester.posterior_sampling.blind_hp.op_updates_per_step=3"])
    saved = S.get_loss_ref
    try:
        S.get_loss_ref = get_loss                    # the product's loss factory in the reference's place (same signature, utils/losses.py:17)
        with _on_device("cuda"):                     # the injected noise stream draws on the default device
            ns = S.NoiseStream(meta[6])
            op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, 16000, num_utts=1, noise=[ns], device="cuda", length=meta[1])
            op.update_H(use_noise=True)
            ref = S.EulerHeunDPSRef(lambda z, cn: net(z, cn), S.EDMRef(args.diff_params.sde_hp), args, ns)
            y = torch.from_numpy(g["y"]).cuda()
            pred = ref.predict_conditional(y, op, shape=(1, meta[1]), blind=True)
    finally:
        S.get_loss_ref = saved
    assert ns.k == int(g["n_draws"])
    p = pred.cpu().numpy()
    print(f"reference control flow on the product operator + network vs the reference's run: rel {rel(p, g['pred']):.2e}, SI-SDR {_sisdr(p, g['pred']):.1f} dB")
    assert rel(p, g["pred"]) < 3e-3
    assert _sisdr(p, g["pred"]) > 40.0
    assert abs(_sisdr(p, g["clean"]) - _sisdr(g["pred"], g["clean"])) < 0.1
    assert rel(op.params[0][0].detach().cpu().numpy(), g["decay"]) < 3e-2
    assert rel(op.params[1][0].detach().cpu().numpy(), g["weights"]) < 3e-2
    assert rel(op.get_time_RIR().detach().cpu().numpy(), g["est_rir"]) < 3e-2


def test_blind_second_order_with_magnitude_constraint(golden):
    """order 2 + constraint_speech_magnitude on the HIP operator: the Heun corrector leaves x_den un-rescaled
    (reference EulerHeunSamplerDPS.py:139-149)."""
    g = golden("e2e_blind_o2")
    p, op, smp = _run_blind(g, ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                "tester.posterior_sampling.blind_hp.op_updates_per_step=2"], "hip")
    assert rel(p, g["pred"]) < 3e-3
    assert _sisdr(p, g["pred"]) > 40.0
    assert abs(_sisdr(p, g["clean"]) - _sisdr(g["pred"], g["clean"])) < 0.1


def test_blind_T10_shipped_updates_fp64_arbiter(golden):
    """T = 10 schedule with the shipped op_updates_per_step = 10 (conf/tester/blind_dereverberation_BUDDy.yaml:72), HIP operator.
    With ten scale-free Adam updates per step the reference ALGORITHM is chaotic in fp32: the CPU oracle (the reference's own torch
    kernels) leaves its float64 trajectory at 127 -> 45 -> 18 -> 13 -> 10 dB within four steps (profiles/archive/r02_arbiter_*.json), so a
    fixture recorded from one fp32 execution cannot be matched sample by sample by ANY other fp32 execution.  The arbiter is the
    algorithm run in float64 (oracle.precision): the build's deviation from that trajectory must stay of the order of the fp32 oracle's
    own (two thread counts = two summation orders) at every step, the first step (before any feedback) must agree to > 100 dB, and the
    reference fixture is reported (and loosely bounded) for the record.
    Margin: ONE denoiser evaluation of the build carries 7-8 dB more round-off than the oracle's (step 0: 120 dB against 127.6 dB to the
    float64 result -- the F(4x4,3x3) Winograd convolutions, unit tolerance 1e-4 instead of 2e-5), and the chain amplifies whatever it is
    given by the same factor per step until it saturates (~10 dB): the build therefore runs that constant 5-8 dB below the oracle through
    steps 1-3 (e.g. 56.5 / 26.9 dB against 62.5 / 25.0-33.2 dB) and level with it afterwards.  Bound: worst oracle - 10 dB."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.arbiter_runs import run_blind, overrides
    from oracle.sampler_ref import NoiseStream
    L, T, nf, up, taps, seeds = 8192, 10, 32, 10, 2000, [0, 1]
    args = compose(overrides=overrides(T, up, nf))
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, nf).items()})
    net = net.cuda().eval()
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    ns = [NoiseStream(9000 + s) for s in seeds]
    t.sampler.noise = ns
    seg, y, op, _ = t.prepare_batch([(synth_clean(s, L), synth_rir(s, taps), f"u{s}.wav") for s in seeds], blind=True, noise=ns)
    smp = t.sampler
    smp.bind(y, op, True)
    sched = smp.create_schedule()
    tl, gl = sched.tolist(), smp.get_gamma(sched).tolist()
    x = smp.initialize_x(tuple(y.shape), "cuda", sched)
    tr = []
    for i in range(T):
        x, xd = smp.step(x, tl[i], tl[i + 1], gl[i], blind=True)
        tr.append(xd.cpu())
    tr = torch.stack(tr)                                   # (T, B, L)
    for b, s in enumerate(seeds):
        x64, clean, k = run_blind(s, L, T, nf, up, taps, fp64=True)
        assert k == ns[b].k
        dev = {}
        for name, xx in (("build", tr[:, b]), ("fp32t8", run_blind(s, L, T, nf, up, taps, threads=8)[0]),
                         ("fp32t3", run_blind(s, L, T, nf, up, taps, threads=3)[0])):
            dev[name] = [_sisdr(xx[i], x64[i]) for i in range(T)]
            print(f"seed {s} {name:7s} SI-SDR to the fp64 trajectory per step:", [round(v, 1) for v in dev[name]])
        assert dev["build"][0] > 100.0
        for i in range(T):      # >= 100 dB is pure fp32 round-off (the F(4x4,3x3) convolutions carry about one more bit of it than direct ones)
            assert dev["build"][i] > min(100.0, min(dev["fp32t8"][i], dev["fp32t3"][i]) - 10.0), (s, i, dev)
    # the reference fixture of the same configuration (weights seed 7, utterance 5): reported, bounded loosely (see above)
    g = golden("e2e_blind10")
    p, op, smp = _run_blind(g, ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled"], "hip")
    s = _sisdr(p, g["pred"])
    print(f"T10 / 10 updates vs the reference's fp32 fixture: SI-SDR(build; reference) {s:.1f} dB, "
          f"delta SI-SDR to clean {_sisdr(p, g['clean']) - _sisdr(g['pred'], g['clean']):+.3f} dB")
    assert s > 5.0 and np.isfinite(p).all()


def test_config1_real_clip_informed(golden):
    """BASELINE config 1: audio_examples p226_003 + its RIR (133 829 samples, 8 083 taps), informed DPS, order 2, T = 10, full-width
    network, through the Tester's own preprocessing (reference testing/tester.py:123-153)."""
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    g = golden("config1")
    nf, L, T, order, seed, nseed, M = [int(v) for v in g["meta"]]
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict
    args = compose(tester="informed_dereverberation_DPS", overrides=[f"tester.sampling_params.T={T}"])
    assert args.tester.sampling_params.order == order and args.network.nf == nf
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(seed, nf).items()})
    net = net.cuda().eval()
    tester = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    tester.sampler.noise = [NoiseStream(nseed)]
    seg, y, op, _ = tester.prepare_batch([(g["clean_raw"].astype(np.float64), g["rir"].astype(np.float64), "p226_003.wav")], blind=False)
    assert rel(seg[0].cpu().numpy(), g["seg"]) < 1e-6
    assert rel(y[0].cpu().numpy(), g["y"][0]) < 2e-5
    pred = tester.sampler.predict_conditional(torch.from_numpy(g["y"]).cuda(), op, shape=(1, L), blind=False)
    assert tester.sampler.noise[0].k == int(g["n_draws"])
    p = pred.cpu().numpy()
    s = _sisdr(p, g["pred"])
    d = abs(_sisdr(p, g["seg"]) - _sisdr(g["pred"], g["seg"]))
    print(f"config 1: SI-SDR(build; reference) {s:.1f} dB, delta to clean {d:.5f} dB, rel {rel(p, g['pred']):.2e}")
    assert s > 40.0
    assert d < 0.1


def test_unconditional_vs_reference_fixture(golden):
    from buddy_amd.instantiate import instantiate
    from buddy_amd.config import compose
    from buddy_amd.synth import synth_state_dict
    from oracle.sampler_ref import NoiseStream
    g = golden("e2e_uncond")
    nf, L, T, order, seed, nseed = [int(v) for v in g["meta"]]
    args = compose(tester="only_unconditional", overrides=[f"tester.sampling_params.T={T}", f"network.nf={nf}"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(seed, nf).items()})
    net = net.cuda().eval()
    smp = instantiate(args.tester.sampler, net, instantiate(args.diff_params), args)

    class Both:   # the reference draws ONE (2, L) tensor per call; replay it as such
        def __init__(self, s): self.s = s
    ns = NoiseStream(nseed)
    smp._randn = lambda shape, device: ns.randn(tuple(shape)).to(device)
    x = smp.predict_unconditional((2, L), "cuda")
    assert ns.k == int(g["n_draws"])
    assert rel(x.cpu().numpy(), g["pred"]) < 2e-3


def test_batched_blind_rows_match_single_runs():
    """B=2 batched blind run: row b == separate B=1 run of utterance b (per-utterance semantics on the GPU path)."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    args = compose(overrides=["tester.sampling_params.T=3", "network.nf=32",
                              "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                              "tester.posterior_sampling.blind_hp.op_updates_per_step=2"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(1, 32).items()})
    net = net.cuda().eval()
    edm = instantiate(args.diff_params)
    L = 8192
    items = [(synth_clean(u, L), synth_rir(u, 1500), f"u{u}.wav") for u in range(2)]

    def run(sel):
        t = Tester(args, net, edm, test_set=None, device="cuda", in_training=True)
        ns = [NoiseStream(300 + u) for u in sel]
        t.sampler.noise = ns
        seg, y, op, _ = t.prepare_batch([items[u] for u in sel], blind=True, noise=ns)
        return t.sampler.predict_conditional(y, op, shape=(len(sel), L), blind=True).cpu().numpy()

    both = run([0, 1])
    for u in range(2):
        one = run([u])
        assert rel(both[u], one[0]) < 1e-3
