"""CPU: pin the oracle (oracle/*.py) against fixtures recorded from the reference itself
(tests/golden/make_golden.py).  Float tolerance: the oracle and the reference run the same torch CPU
kernels in (almost) the same order, so agreement is at fp32 round-off: rtol 2e-4 of the tensor's abs-max
for network outputs, 1e-5 for scalar formulas."""
import numpy as np
import pytest
import torch

from buddy_amd.config import compose
from buddy_amd.synth import synth_state_dict
from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S

torch.set_num_threads(8)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_edm_schedule_gamma(golden):
    g = golden("edm_sched")
    for tester in ["blind_dereverberation_BUDDy", "informed_dereverberation_DPS", "only_unconditional"]:
        for T in (10, 50, 201):
            args = compose(tester=tester, overrides=[f"tester.sampling_params.T={T}"])
            t = S.create_schedule(args.tester.sampling_params.sde_hp, T)
            assert np.allclose(t.numpy(), g[f"{tester}.T{T}.t"], rtol=1e-6, atol=0)
            gam = S.get_gamma(t, args.tester.sampling_params)
            assert np.allclose(gam.numpy(), g[f"{tester}.T{T}.gamma"], rtol=1e-6, atol=0)
    args = compose()
    edm = S.EDMRef(args.diff_params.sde_hp)
    sig = torch.from_numpy(g["sigma"])
    for k in ["cskip", "cout", "cin", "cnoise"]:
        assert np.allclose(getattr(edm, k)(sig).numpy(), g[k], rtol=1e-6)


@pytest.mark.parametrize("name", ["net_small", "net_full", "net_cm12_rb2", "net_cm1122_rb1", "net_full_64000"])
def test_network_forward_and_vjp(golden, name):
    """net_full_64000 (round 6): the FULL size -- nf = 128, L = 64 000, BASELINE configs[1]'s utterance -- recorded from the reference
    itself: the oracle the 64 000-sample GPU tests lean on is pinned at that size too, not only at 16 000.
    net_cm12_rb2 / net_cm1122_rb1: other members of the architecture family the constructor accepts (reference networks/ncsnpp.py:184-270:
    ch_mult (1, 2) with two blocks per level, nf 32; ch_mult (1, 1, 2, 2) with one, nf 32), recorded from the reference like the shipped ones."""
    g = golden(name)
    nf, n_fft, hop, L, B, seed = [int(v) for v in g["meta"]]
    ch_mult = tuple(int(c) for c in g["ch_mult"]) if "ch_mult" in g.files else (1, 2, 2, 2)
    nrb = int(g["num_res_blocks"]) if "num_res_blocks" in g.files else 1
    P = ncsnpp_ref.to_torch(synth_state_dict(seed, nf, ch_mult, nrb))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    taps = {}
    y = ncsnpp_ref.ncsnpp_time(P, x, torch.from_numpy(g["cnoise"]), n_fft, hop, ch_mult=ch_mult, num_res_blocks=nrb, taps=taps)
    vjp, = torch.autograd.grad(y, x, torch.from_numpy(g["cot"]))
    assert rel(y.detach().numpy(), g["y"]) < 2e-4
    assert rel(vjp.numpy(), g["vjp"]) < 2e-4
    n = 0
    for k in g.files:
        if k.startswith("tap") and k.endswith("_absmax"):
            i = int(k[3:-7])
            assert abs(float(taps[i].abs().max()) - float(g[k])) < 2e-4 * float(g[k]) + 1e-6
            assert abs(float(taps[i].std()) - float(g[f"tap{i}_std"])) < 2e-4 * float(g[f"tap{i}_std"]) + 1e-6
            n += 1
    assert n >= (18 if len(ch_mult) == 4 and nrb == 1 else 12)


def test_operator_pieces(golden):
    g = golden("ops")
    args = compose()
    op_hp = args.tester.informed_dereverberation.op_hp
    x, rir = torch.from_numpy(g["x"]), torch.from_numpy(g["rir"])
    op = O.RIROperatorRef(op_hp)
    op.update_params(rir)
    y = op.degradation(x[None])
    assert rel(y, g["y_rir"]) < 1e-5
    assert rel(torch.view_as_real(op.apply_stft(y)), g["apply_stft_y"]) < 1e-5
    loss = O.get_loss_ref(args.tester.posterior_sampling.rec_loss, op)
    xd = torch.from_numpy(g["inf_xd"]).requires_grad_(True)
    val = loss(y, op.degradation(xd))
    assert abs(float(val) - float(g["inf_loss"])) < 1e-5 * abs(float(g["inf_loss"]))
    assert rel(torch.autograd.grad(val, xd)[0], g["inf_loss_grad"]) < 1e-4
    assert rel(O.minimum_phase_ref(torch.from_numpy(g["minphase_in"])), g["minphase_out"]) < 1e-5

    ns = S.NoiseStream(11)
    bop = O.BlindSubbandFilteringRef(op_hp, 16000, ns)
    bop.update_H(use_noise=True, noise=ns)
    assert rel(bop.design_filter().detach(), g["blind_A"]) < 1e-5
    assert rel(torch.view_as_real(bop.H.detach()), g["blind_H"]) < 1e-4
    assert rel(bop.params_phases[0], g["blind_phases"]) < 1e-3   # angles: wrap-sensitive near +-pi
    assert rel(bop.degradation(x[None]).detach(), g["blind_deg"]) < 1e-4
    assert rel(bop.get_time_RIR().detach(), g["blind_rir"]) < 1e-4
    assert rel(torch.view_as_real(bop.apply_stft(x[None])), g["blind_stft_x"]) < 1e-5
    ps = args.tester.posterior_sampling
    lp, lr = O.get_loss_ref(ps.rec_loss_params, bop), O.get_loss_ref(ps.RIR_noise_regularization.loss, bop)
    for p in bop.params + bop.params_phases:
        p.requires_grad = True
    bop.update_H()
    l1 = lp(y, bop.degradation(x[None]))
    rt = bop.get_time_RIR()
    n = ns.randn(rt.shape)
    l2 = lr(rt, (rt + 0.005 * n).detach())
    gs = torch.autograd.grad(l1 + l2, bop.params + bop.params_phases)
    assert abs(float(l1) - float(g["blind_l_rec"])) < 1e-4 * abs(float(g["blind_l_rec"]))
    assert abs(float(l2) - float(g["blind_l_reg"])) < 1e-4 * abs(float(g["blind_l_reg"]))
    assert rel(gs[0], g["blind_g_decay"]) < 2e-3
    assert rel(gs[1], g["blind_g_weights"]) < 2e-3
    assert rel(gs[2], g["blind_g_phases"]) < 2e-3
    with torch.no_grad():
        bop.params[0].copy_(torch.linspace(0.0, 0.8, 25)[None])
        bop.params[1].copy_(torch.linspace(0.2, 150.0, 25)[None])
    bop.project_params()
    assert np.array_equal(bop.params[0].detach().numpy(), g["proj_decay"])
    assert np.array_equal(bop.params[1].detach().numpy(), g["proj_weights"])


def _run_e2e(g, tester, blind, extra=()):
    nf, L, T, order, seed, utt, nseed = [int(v) for v in g["meta"]]
    args = compose(tester=tester, overrides=[f"tester.sampling_params.T={T}",
                                             f"tester.sampling_params.order={order}"] + list(extra))
    P = ncsnpp_ref.to_torch(synth_state_dict(seed, nf))
    net = lambda x, cn: ncsnpp_ref.ncsnpp_time(P, x, cn, 510, 128)
    ns = S.NoiseStream(nseed)
    smp = S.EulerHeunDPSRef(net, S.EDMRef(args.diff_params.sde_hp), args, ns)
    op_hp = args.tester.informed_dereverberation.op_hp
    clean, rir = torch.from_numpy(g["clean"]), torch.from_numpy(g["rir"])
    op_ref = O.RIROperatorRef(op_hp)
    op_ref.update_params(rir)
    y = op_ref.degradation(clean[None])
    assert rel(y, g["y"]) < 1e-5
    op = op_ref
    if blind:
        op = O.BlindSubbandFilteringRef(op_hp, 16000, ns)
        op.update_H(use_noise=True, noise=ns)
    pred = smp.predict_conditional(y, op, shape=(1, L), blind=blind)
    assert ns.k == int(g["n_draws"])
    return pred, op


def test_e2e_informed(golden):
    g = golden("e2e_informed")
    pred, _ = _run_e2e(g, "informed_dereverberation_DPS", False)
    assert rel(pred, g["pred"]) < 1e-3


def test_e2e_blind(golden):
    g = golden("e2e_blind")
    pred, op = _run_e2e(g, "blind_dereverberation_BUDDy", True,
                        ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                         "tester.posterior_sampling.blind_hp.op_updates_per_step=3"])
    assert rel(pred, g["pred"]) < 2e-3
    # Adam's update m/sqrt(v) is scale-free, so one band whose gradient is at round-off level moves by O(1 %) between runs of the
    # SAME torch code with different thread partitions (observed 0.3994 vs 0.4003); the output above is insensitive to it
    assert rel(op.params[0].detach(), g["decay"]) < 1e-2
    assert rel(op.params[1].detach(), g["weights"]) < 1e-2
    assert rel(op.get_time_RIR().detach(), g["est_rir"]) < 1e-2


def test_e2e_blind_full_size(golden):
    """round 6: the blind sampler at the FULL size against the reference's own run -- nf = 128, L = 64 000 (BASELINE configs[1]'s utterance), 3 operator
    updates per step, T = 3 (reference testing/EulerHeunSamplerDPS.py:115-204).  Nine scale-free Adam updates in: the parameters are held
    to 2e-2 (two executions of the reference's own code move single bands by a percent, see test_e2e_blind), the estimate to 2e-3 of its abs-max."""
    g = golden("e2e_blind_full")
    torch.set_num_threads(8)
    pred, op = _run_e2e(g, "blind_dereverberation_BUDDy", True, ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                                                    "tester.posterior_sampling.blind_hp.op_updates_per_step=3"])
    print("oracle vs the reference's full-size blind run:", rel(pred, g["pred"]), rel(op.params[0].detach(), g["decay"]), rel(op.get_time_RIR().detach(), g["est_rir"]))
    assert rel(pred, g["pred"]) < 2e-3
    assert rel(op.params[0].detach(), g["decay"]) < 2e-2 and rel(op.params[1].detach(), g["weights"]) < 2e-2
    assert rel(op.get_time_RIR().detach(), g["est_rir"]) < 2e-2


def test_e2e_unconditional(golden):
    g = golden("e2e_uncond")
    nf, L, T, order, seed, nseed = [int(v) for v in g["meta"]]
    args = compose(tester="only_unconditional", overrides=[f"tester.sampling_params.T={T}"])
    P = ncsnpp_ref.to_torch(synth_state_dict(seed, nf))
    net = lambda x, cn: ncsnpp_ref.ncsnpp_time(P, x, cn, 510, 128)
    ns = S.NoiseStream(nseed)
    smp = S.EulerHeunRef(net, S.EDMRef(args.diff_params.sde_hp), args, ns)
    x = smp.predict_unconditional((2, L))
    assert ns.k == int(g["n_draws"])
    assert rel(x, g["pred"]) < 1e-3


def test_e2e_blind_second_order_with_magnitude_constraint(golden):
    """order 2 + constraint_speech_magnitude: the Heun corrector leaves x_den un-rescaled (reference EulerHeunSamplerDPS.py:139-149)."""
    g = golden("e2e_blind_o2")
    pred, op = _run_e2e(g, "blind_dereverberation_BUDDy", True,
                        ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                         "tester.posterior_sampling.blind_hp.op_updates_per_step=2"])
    assert rel(pred, g["pred"]) < 2e-3
    assert rel(op.get_time_RIR().detach(), g["est_rir"]) < 1e-2


def test_optimize_op_one_and_ten_iterations(golden):
    """optimize_op (reference EulerHeunSamplerDPS.py:71-113): Adam state and parameters after one, three and ten full iterations (the
    oracle runs the reference's own torch CPU kernels, so it tracks the fixture far inside the fp32 divergence described in
    tests/test_hip_operator_golden.py::test_optimize_op_vs_reference)."""
    g = golden("opt")
    args = compose(overrides=["tester.posterior_sampling.blind_hp.op_updates_per_step=1"])
    ps, op_hp = args.tester.posterior_sampling, args.tester.informed_dereverberation.op_hp
    ns = S.NoiseStream(int(g["meta"][1]))
    bop = O.BlindSubbandFilteringRef(op_hp, 16000, ns)
    bop.update_H(use_noise=True, noise=ns)
    smp = S.EulerHeunDPSRef(None, S.EDMRef(args.diff_params.sde_hp), args, ns)
    smp.operator, smp.y = bop, torch.from_numpy(g["y"])
    smp.rec_loss_params = O.get_loss_ref(ps.rec_loss_params, bop)
    smp.rir_reg_loss = O.get_loss_ref(ps.RIR_noise_regularization.loss, bop)
    smp.optim = torch.optim.Adam(bop.params + bop.params_phases, lr=ps.blind_hp.lr_op, weight_decay=ps.blind_hp.weight_decay,
                                 betas=(ps.blind_hp.beta1, ps.blind_hp.beta2))
    x_den, t = torch.from_numpy(g["x_den"]), torch.tensor(float(g["t"]))

    def check(tag, tol):
        assert ns.k == int(g[f"{tag}_n_draws"])
        for nm, p in (("decay", bop.params[0]), ("weights", bop.params[1]), ("phases", bop.params_phases[0])):
            st = smp.optim.state[p]
            assert rel(st["exp_avg"], g[f"{tag}_m_{nm}"]) < tol
            assert rel(st["exp_avg_sq"], g[f"{tag}_v_{nm}"]) < tol
        assert rel(bop.params[0].detach(), g[f"{tag}_decay"]) < tol
        assert rel(bop.params[1].detach(), g[f"{tag}_weights"]) < tol
        A = bop.design_filter().detach()
        assert rel(torch.view_as_real(A * torch.exp(1j * bop.params_phases[0].detach())),
                   torch.view_as_real(A * torch.exp(1j * torch.from_numpy(g[f"{tag}_phases"])))) < tol

    smp.optimize_op(x_den.clone(), t)
    check("it1", 2e-3)
    ps.blind_hp.op_updates_per_step = 2
    smp.optimize_op(x_den.clone(), t)
    check("it3", 2e-3)
    ps.blind_hp.op_updates_per_step = 7
    smp.optimize_op(x_den.clone(), t)
    check("it10", 1e-2)
    assert rel(O.minimum_phase_ref(torch.from_numpy(g["minphase_in"])), g["minphase_out"]) < 1e-5


def test_fir_resampling_and_network(golden):
    """fir=True: upsample_2d / downsample_2d with the (1,3,3,1) kernel and a small network built with fir=True, against vectors
    recorded from the reference's pure-PyTorch upfirdn2d (op/upfirdn2d.py:171-215, the branch its dispatcher takes on CPU)."""
    g = golden("fir_ops")
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    up, dn = ncsnpp_ref.fir_up2(x), ncsnpp_ref.fir_down2(x)
    assert rel(up.detach(), g["up"]) < 1e-6 and rel(dn.detach(), g["down"]) < 1e-6
    gu, = torch.autograd.grad(up, x, torch.from_numpy(g["cot_up"]), retain_graph=True)
    gd, = torch.autograd.grad(dn, x, torch.from_numpy(g["cot_down"]))
    assert rel(gu, g["vjp_up"]) < 1e-6 and rel(gd, g["vjp_down"]) < 1e-6
    # the transposes are the other direction up to a constant (what the HIP VJP relies on)
    assert rel(4 * ncsnpp_ref.fir_down2(torch.from_numpy(g["cot_up"])), g["vjp_up"]) < 1e-6
    assert rel(0.25 * ncsnpp_ref.fir_up2(torch.from_numpy(g["cot_down"])), g["vjp_down"]) < 1e-6
    n = golden("net_small_fir")
    nf, n_fft, hop, L, B, seed = [int(v) for v in n["meta"]]
    P = ncsnpp_ref.to_torch(synth_state_dict(seed, nf))
    xi = torch.from_numpy(n["x"]).requires_grad_(True)
    taps = {}
    y = ncsnpp_ref.ncsnpp_time(P, xi, torch.from_numpy(n["cnoise"]), n_fft, hop, taps=taps, fir=True)
    vjp, = torch.autograd.grad(y, xi, torch.from_numpy(n["cot"]))
    assert rel(y.detach(), n["y"]) < 2e-4 and rel(vjp, n["vjp"]) < 2e-4
    for k in n.files:
        if k.startswith("tap") and k.endswith("_absmax"):
            i = int(k[3:-7])
            assert abs(float(taps[i].abs().max()) - float(n[k])) < 2e-4 * float(n[k]) + 1e-6
