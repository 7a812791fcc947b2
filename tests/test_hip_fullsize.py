"""GPU tests at BASELINE.json's full size (NCSN++ nf=128, 4 s @ 16 kHz = 64 000 samples): direct parity against the CPU oracle
for one utterance, and size-independent properties (adjoint identity <J v, w> = <v, J^T w> by central differences, linearity of the
VJP, bit-exact batch independence, per-utterance magnitude constraint of the sampler)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
L = 64000


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def net():
    from tests.test_hip_network import build
    return build(128, 510, 128, 0)


def test_fullsize_forward_vjp_vs_oracle(net):
    from oracle import ncsnpp_ref
    from buddy_amd.synth import synth_state_dict
    torch.set_num_threads(32)
    P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
    rs = np.random.RandomState(5)
    x = torch.from_numpy((0.4 * rs.standard_normal((1, L))).astype(np.float32))
    cot = torch.from_numpy(rs.standard_normal((1, L)).astype(np.float32))
    cn = torch.tensor([-0.6])
    xr = x.clone().requires_grad_(True)
    yr = ncsnpp_ref.ncsnpp_time(P, xr, cn, 510, 128)
    gr, = torch.autograd.grad(yr, xr, cot)
    xg = x.cuda().requires_grad_(True)
    y = net(xg, cn.cuda())
    g, = torch.autograd.grad(y, xg, cot.cuda())
    assert rel(y, yr) < 5e-4
    assert rel(g, gr) < 5e-4


def test_longform_forward_vjp_vs_oracle(net):
    """Maximum documented size (BASELINE.json configs[4]: 30 s = 480 000 samples, 3 751 -> 3 760 frames, 15 040 attention tokens;
    the lowest level falls back to the direct 3x3 kernel because its 470 rows are not a multiple of 8)."""
    from oracle import ncsnpp_ref
    from buddy_amd.synth import synth_state_dict
    torch.set_num_threads(32)
    LL = 480000
    P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
    rs = np.random.RandomState(11)
    x = torch.from_numpy((0.4 * rs.standard_normal((1, LL))).astype(np.float32))
    cot = torch.from_numpy(rs.standard_normal((1, LL)).astype(np.float32))
    cn = torch.tensor([-0.6])
    xr = x.clone().requires_grad_(True)
    yr = ncsnpp_ref.ncsnpp_time(P, xr, cn, 510, 128)
    gr, = torch.autograd.grad(yr, xr, cot)
    xg = x.cuda().requires_grad_(True)
    y = net(xg, cn.cuda())
    g, = torch.autograd.grad(y, xg, cot.cuda())
    assert rel(y, yr) < 5e-4
    assert rel(g, gr) < 5e-4


def test_fullsize_adjoint_identity_and_linearity(net):
    rs = np.random.RandomState(6)
    B = 2
    x = torch.from_numpy((0.3 * rs.standard_normal((B, L))).astype(np.float32)).cuda()
    v = torch.from_numpy(rs.standard_normal((B, L)).astype(np.float32)).cuda()
    w1 = torch.from_numpy(rs.standard_normal((B, L)).astype(np.float32)).cuda()
    w2 = torch.from_numpy(rs.standard_normal((B, L)).astype(np.float32)).cuda()
    cn = torch.tensor([-0.4, -1.1]).cuda()

    def vjp(w):
        xg = x.clone().requires_grad_(True)
        y = net(xg, cn)
        return torch.autograd.grad(y, xg, w)[0]

    g1, g2, g12 = vjp(w1), vjp(w2), vjp(0.5 * w1 + w2)
    # linear in the cotangent: three separate VJP evaluations, each with the ~4e-6 round-off of the network backward (F(6x6,3x3) convolutions;
    # vs the oracle 3.8e-6) -- measured 1.0e-5 ... 1.1e-5, a missing or non-linear term would show at 1e-2
    assert rel(g12, 0.5 * g1 + g2) < 3e-5
    eps = 1e-2
    with torch.no_grad():
        jv = (net(x + eps * v, cn).double() - net(x - eps * v, cn).double()) / (2 * eps)
    lhs = (jv * w1.double()).sum(dim=1)
    rhs = (v.double() * g1.double()).sum(dim=1)
    assert float(((lhs - rhs).abs() / rhs.abs()).max()) < 2e-2      # central difference of a fp32 network: O(eps^2) + round-off/eps
    # rows are independent and bit-identical to B=1 calls
    with torch.no_grad():
        yb = net(x, cn)
        y0 = net(x[:1], cn[:1])
    assert torch.equal(yb[:1], y0)


def _two_blind_steps_vs_oracle(net, L, B, rir_taps, updates=3, check=(0,), floor_db=40.0, rel_tol=1e-2):
    """Two blind DPS steps (network forward + VJP, HIP operator optimisation, likelihood, fused update): utterances `check` of a batch of B
    against the oracle's B=1 runs (same noise draws)."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from buddy_amd.utils.metrics import si_sdr
    from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S
    ov = ["tester.sampling_params.T=50", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
          f"tester.posterior_sampling.blind_hp.op_updates_per_step={updates}"]
    args = compose(overrides=ov)
    edm = instantiate(args.diff_params)
    items = [(synth_clean(u, L), synth_rir(u, rir_taps), f"u{u}.wav") for u in range(B)]
    t = Tester(args, net, edm, test_set=None, device="cuda", in_training=True)
    ns = [S.NoiseStream(70 + u) for u in range(B)]
    t.sampler.noise = ns
    seg, y, op, _ = t.prepare_batch(items, blind=True, noise=ns)
    smp = t.sampler
    smp.bind(y, op, True)
    sched = smp.create_schedule().cuda(); gam = smp.get_gamma(sched).cuda()
    x = smp.initialize_x(tuple(y.shape), "cuda", sched)
    for i in range(2):
        x, x_den = smp.step(x, sched[i], sched[i + 1], gam[i], blind=True)
    assert torch.isfinite(x).all() and torch.isfinite(x_den).all()
    assert float((x_den.std(dim=1) - 0.05).abs().max()) < 1e-5          # constraint_speech_magnitude per utterance
    torch.set_num_threads(32)
    P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
    onet = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
    for u in check:                                                     # oracle, utterance u on its own (per-utterance semantics)
        nr = S.NoiseStream(70 + u)
        ref = S.EulerHeunDPSRef(onet, S.EDMRef(args.diff_params.sde_hp), args, nr)
        op_hp = args.tester.informed_dereverberation.op_hp
        oo = O.RIROperatorRef(op_hp); oo.update_params(torch.from_numpy(items[u][1]))
        c0 = torch.from_numpy(items[u][0]); c0 = 0.05 * c0 / c0.std()
        y0 = oo.degradation(c0[None])
        assert rel(y[u:u + 1], y0) < 1e-4
        bo = O.BlindSubbandFilteringRef(op_hp, 16000, nr); bo.update_H(use_noise=True, noise=nr)
        ref.operator, ref.y = bo, y0
        ps = args.tester.posterior_sampling
        ref.rec_loss = O.get_loss_ref(ps.rec_loss, bo); ref.rec_loss_params = O.get_loss_ref(ps.rec_loss_params, bo)
        ref.rir_reg_loss = O.get_loss_ref(ps.RIR_noise_regularization.loss, bo)
        ref.optim = torch.optim.Adam(bo.params + bo.params_phases, lr=ps.blind_hp.lr_op, betas=(ps.blind_hp.beta1, ps.blind_hp.beta2))
        ts = S.create_schedule(ref.sde_hp, ref.T); gm = S.get_gamma(ts, ref.sp)
        xr = ref.initialize_x(y0.shape, ts)
        for i in range(2):
            xr, xdr = ref.step(xr, ts[i], ts[i + 1], gm[i], True)
        assert nr.k == ns[u].k
        s = float(si_sdr(x_den[u:u + 1].cpu(), xdr))
        print(f"L={L} B={B} updates={updates} utterance {u}: two blind steps, SI-SDR(build; oracle) = {s:.1f} dB, rel {rel(x_den[u:u + 1], xdr):.2e}")
        assert s > floor_db, f"utterance {u}: SI-SDR(build; oracle) = {s:.1f} dB"
        assert rel(x_den[u:u + 1], xdr) < rel_tol


def test_fullsize_blind_sampler_two_steps_vs_oracle(net):
    _two_blind_steps_vs_oracle(net, 64000, 2, 4000)


def test_fullsize_blind_configs1_B8_10updates_vs_oracle(net):
    """BASELINE.json configs[1] as specified -- B = 8 utterances x 64 000 samples, the shipped 10 operator updates per step, T = 50
    schedule -- two steps, first and last utterance of the batch against the oracle's single-utterance runs.  With ten scale-free Adam
    updates per step the chain amplifies fp32 round-off by ~75 dB per step (the fp32 oracle itself sits at 53 dB of its
    float64 trajectory after step 2, the build at 54 dB), so two fp32 executions agree to ~50 dB after two steps, not to 87 dB as with 3 updates."""
    _two_blind_steps_vs_oracle(net, 64000, 8, 8000, updates=10, check=(0, 7), floor_db=35.0, rel_tol=3e-2)


def test_longform_blind_sampler_two_steps_vs_oracle(net):
    """BASELINE config 5 shape on one GPU: 30 s utterances (480 000 samples, 15 040 attention tokens through the flash kernel -- no
    905 MB attention matrix), B = 4, un-chunked."""
    _two_blind_steps_vs_oracle(net, 480000, 4, 8000)


def test_longform_f16_attention_blind_sampler_two_steps_vs_oracle():
    """BASELINE config 5 names an fp16 MFMA attention path: the same two blind steps at 480 000 samples, B = 4, with
    NCSNppTime(attention="f16") (16-bit MFMA operands in the attention kernels only, fp32 accumulate / softmax) against the fp32 oracle."""
    from tests.test_hip_network import build
    _two_blind_steps_vs_oracle(build(128, 510, 128, 0, attention="f16"), 480000, 4, 8000)


def test_longform_chunked_policy(net):
    """testing/longform.py through the Tester: a 10 s clip as overlapping 4 s chunks sampled as one batch (informed, 2 steps) and cross-faded;
    a clip that fits one chunk takes the un-chunked path bit for bit."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from buddy_amd.utils.metrics import si_sdr
    from oracle.sampler_ref import NoiseStream
    args = compose(tester="informed_dereverberation_DPS", overrides=["tester.sampling_params.T=3"])
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    LL = 160000
    clean, rir = synth_clean(3, LL), synth_rir(3, 4000)
    mk = lambda n: [NoiseStream(40 + u) for u in range(n)]
    seg, y, pred = t.dereverberate_long(clean, rir, blind=False, chunk_seconds=4.0, overlap_seconds=0.5, noise=mk)
    assert pred.shape == (LL,) and torch.isfinite(pred).all()
    seg1, y1, whole = t.dereverberate_long(clean, rir, blind=False, chunk_seconds=10.0, overlap_seconds=0.5, noise=mk)
    t.sampler.noise = mk(1)
    _, yb, op, _ = t.prepare_batch([(clean, rir, "long.wav")], blind=False)
    direct = t.sampler.predict_conditional(yb, op, shape=(1, LL), blind=False)
    assert torch.equal(whole, direct[0])
    print(f"chunked (4 s / 0.5 s overlap) vs un-chunked, 10 s clip, 3-step informed run: SI-SDR {float(si_sdr(pred[None].cpu(), whole[None].cpu())):.1f} dB")


def _sd(a, b):
    from buddy_amd.utils.metrics import si_sdr
    return float(si_sdr(torch.as_tensor(a).double().reshape(1, -1), torch.as_tensor(b).double().reshape(1, -1)))


def test_fullsize_blind_T50_population_fp64_arbiter(net):
    """BASELINE configs[1] as specified, through the WHOLE schedule, at the CURRENT arithmetic, as a POPULATION gate (rounds 3-5
    gated two seeds at |delta| < 3 dB): L = 64 000, nf = 128, T = 50, order 1, 10 operator updates per step, EIGHT utterances / noise seeds sampled as
    ONE B = 8 batch by the build -- the headline workload itself.  Arbiter = the restated algorithm in float64 through torch ops on the GPU, batched the
    same way (oracle.arbiter_runs.run_blind_batched: row b == the B = 1 oracle run of utterance b), executed TWICE (inputs scaled by 1 + 1e-13 the
    second time): the blind chain is chaotic in the reference's own arithmetic (reference testing/EulerHeunSamplerDPS.py:71-157: ten scale-free Adam
    steps per diffusion step), so two float64 executions separate, and their separation is the resolution "within 0.1 dB of the reference" can be
    tested at.  Asserted:
      * final estimates: median over the 8 utterances of |SI-SDR(build; clean) - SI-SDR(float64; clean)| <= the same median between the two float64
        executions + 0.1 dB; seven of the eight utterances <= the larger of 1.5 dB and the two float64 executions' own maximum + 0.5 dB; the eighth
        <= 12 dB.  Measured in round 6 on two builds that differ in fp32 rounding details only: medians 0.80 and 0.15 dB (two float64 runs: 0.74),
        maxima 2.20 and 4.68 dB (two float64 runs: 2.18; torch's own fp32 GPU kernels: 10.5) -- always utterance 7, whose chain is the most sensitive of
        the population: a maximum over eight chaotic chains is not a statistic one can hold to 1.5 dB, the median and the second largest are;
      * every step (first two utterances; this gate replaces the B = 1 T = 10 and two-seed T = 50 arbiter tests of rounds 2-5, 250 s of the suite): the build's SI-SDR to the float64 trajectory is not more than 10 dB below that of one more fp32 execution (the same batched
        algorithm through torch's own fp32 GPU kernels), capped at 100 dB = the fp32 round-off floor; the first step (before any feedback) is at
        that floor."""
    import json
    import os
    import time
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.arbiter_runs import run_blind_batched, overrides
    from oracle.sampler_ref import NoiseStream
    T, nf, up, taps, seeds = 50, 128, 10, 8000, list(range(8))
    args = compose(overrides=overrides(T, up, nf))
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
    t0 = time.time()
    for i in range(T):
        x, xd = smp.step(x, tl[i], tl[i + 1], gl[i], blind=True)
        tr.append(xd.cpu())
    torch.cuda.synchronize()
    t_build = time.time() - t0
    assert torch.isfinite(x).all()
    def batched(**kw):           # two half-batches: torch's autograd keeps every float64 activation of the 36-module network alive (~15 GB per utterance)
        parts = [run_blind_batched(seeds[lo:lo + 4], L, T, nf, up, taps, **kw) for lo in (0, 4)]
        torch.cuda.empty_cache()
        return torch.cat([p[0] for p in parts], dim=1), torch.cat([p[1] for p in parts]), parts[0][2] + parts[1][2]
    t0 = time.time()
    a64, clean, k = batched(fp64=True)
    t_a = time.time() - t0
    assert k == [n.k for n in ns], "noise streams out of step"
    b64, _, _ = batched(fp64=True, perturb=1e-13)
    t0 = time.time()
    o32, _, k32 = run_blind_batched(seeds[:2], L, T, nf, up, taps, fp64=False)      # the per-step floor on two utterances (80 s for all eight)
    t_o = time.time() - t0
    assert k32 == k[:2]
    d_build, d_64, d_o32, first = [], [], [], []
    for b, seed in enumerate(seeds):
        ref_c = _sd(a64[-1][b], clean[b])
        d_build.append(abs(_sd(tr[-1][b], clean[b]) - ref_c))
        d_64.append(abs(_sd(b64[-1][b], clean[b]) - ref_c))
        bd = [_sd(tr[i][b], a64[i][b]) for i in range(T)]
        first.append(bd[0])
        if b < o32.shape[1]:
            d_o32.append(abs(_sd(o32[-1][b], clean[b]) - ref_c))
            od = [_sd(o32[i][b], a64[i][b]) for i in range(T)]
            for i in range(T):
                assert bd[i] > min(100.0, od[i]) - 10.0, (seed, i, bd[i], od[i])
    med = lambda v: float(np.median(v))
    rep = {"what": "blind DPS, L = 64000, nf = 128, T = 50, order 1, 10 operator updates per step, seeds 0-7 as ONE B = 8 batch; |delta SI-SDR to clean| of the "
                   "final estimate against the float64 execution (dB), per utterance",
           "gemm": {0: "fp32", 1: "bf16x3", 2: "f16x2"}[net.get_option("gemm")],
           "build": [round(v, 3) for v in d_build], "second_float64_execution": [round(v, 3) for v in d_64], "fp32_torch_gpu_kernels": [round(v, 3) for v in d_o32],
           "median": {"build": med(d_build), "second_float64_execution": med(d_64), "fp32_torch_gpu_kernels": med(d_o32)},
           "max": {"build": max(d_build), "second_float64_execution": max(d_64), "fp32_torch_gpu_kernels": max(d_o32)},
           "first_step_SI_SDR_to_float64_dB": [round(v, 1) for v in first],
           "seconds": {"build_T50_B8": round(t_build, 1), "float64_batched": round(t_a, 1), "fp32_torch_batched": round(t_o, 1)}}
    print("population gate:", json.dumps(rep))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(rep, open(os.path.join(out, "r06_arbiter_L64000_T50.json"), "w"), indent=1)
    assert min(first) > 105.0, first
    assert med(d_build) <= med(d_64) + 0.1, rep["median"]
    srt = sorted(d_build)
    assert srt[-2] <= max(1.5, max(d_64) + 0.5) and srt[-1] <= 12.0, rep["build"]


def test_precision_budget_one_denoiser_evaluation_vs_fp64(net):
    """Gate on the round-off the kernels may spend: ONE denoiser evaluation D(x; sigma) and its input-VJP at full
    width / full length against the algorithm in float64.  Today: build 114 dB (F(6x6,3x3) + F(4x4,3x3) Winograd convolutions with fused
    GroupNorms), fp32 oracle 130 dB.  A kernel change that spends more than 4 dB of it turns this red."""
    from buddy_amd.instantiate import instantiate
    from buddy_amd.config import compose
    from oracle.arbiter_runs import denoiser_eval
    args = compose()
    edm = instantiate(args.diff_params)
    for sigma in (0.5, 0.02):
        d64, g64, x, w = denoiser_eval(0, L, 128, sigma, fp64=True, device="cuda")
        xg = x.cuda().requires_grad_(True)
        d = edm.denoiser(xg, net, sigma)
        g, = torch.autograd.grad(d, xg, w.cuda())
        sd_d, sd_g = _sd(d.detach().cpu(), d64), _sd(g.cpu(), g64)
        # what the network itself contributes: D = c_skip x + c_out F, so compare F through (D - c_skip x) as well
        print(f"sigma {sigma}: one denoiser evaluation vs float64: D {sd_d:.1f} dB, VJP {sd_g:.1f} dB")
        assert sd_d > 110.0 and sd_g > 108.0, (sigma, sd_d, sd_g)      # measured r03: 115.9 / 113.2 dB (sigma 0.5), 126.4 / 122.8 dB (0.02)


def test_precision_budget_informed_T50_vs_fp64(net):
    """The non-chaotic chain over the whole schedule: informed DPS (known RIR), order 2, T = 50 = 99 denoiser evaluations with VJP, full
    size, against the float64 execution of the same algorithm (torch ops on the GPU).  Round 2: 66.8 dB against the fp32 oracle (70.2 dB
    with F(4x4,3x3) convolutions only).  Floor 65 dB; |delta SI-SDR to clean| < 0.01 dB (north star: 0.1 dB)."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.arbiter_runs import run_informed
    from oracle.sampler_ref import NoiseStream
    T, seed, taps = 50, 0, 8000
    args = compose(tester="informed_dereverberation_DPS", overrides=[f"tester.sampling_params.T={T}"])
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    ns = [NoiseStream(9000 + seed)]
    t.sampler.noise = ns
    seg, y, op, _ = t.prepare_batch([(synth_clean(seed, L), synth_rir(seed, taps), "u.wav")], blind=False, noise=ns)
    pred = t.sampler.predict_conditional(y, op, shape=(1, L), blind=False)
    x64, clean, k = run_informed(seed, L, T, 128, taps, fp64=True, device="cuda")
    assert k == ns[0].k
    s = _sd(pred[0].cpu(), x64[-1])
    dc = _sd(pred[0].cpu(), clean) - _sd(x64[-1], clean)
    print(f"informed T=50 order 2, full size: SI-SDR(build; float64 run) {s:.1f} dB, delta SI-SDR to clean {dc:+.5f} dB")
    # measured r03: 72.7 dB; round 5 (f16x2 GEMMs, float64 arbiter made reproducible: it was 74 dB from ITSELF run to run): 70.5 dB; bf16x3 71.2, fp32 MFMA 72.1
    assert s > 68.0
    assert abs(dc) < 0.01
