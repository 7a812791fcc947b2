"""The SHIPPED blind configuration run to the end: conf/tester/blind_dereverberation_BUDDy.yaml untouched -- wpe_scaled warm
start, T = 201, order 1, 10 operator updates per step (reference test_blind_dereverberation.sh:18) -- on one 4 s synthetic utterance:

  * the MI355X build through the product classes (Tester.prepare_batch + sampler), wall time incl. the WPE warm start;
  * the oracle in float64 through the same torch ops on the GPU (the arbiter: the algorithm's trajectory for these inputs and noise draws);
  * the oracle in fp32 -- on the GPU through torch's kernels (rocFFT / MIOpen: "another fp32 execution") and, with --cpu-oracle, on the host
    cores (the reference's own CPU arithmetic; ~15 min for 201 steps);
  * a second float64 run with the input scaled by 1 + 1e-13 (the resolution of the arbiter for a chaotic chain).

Every execution is measured against the float64 trajectory: per-step SI-SDR of x_den, SI-SDR of the final estimate to clean.
    python tools/shipped_run.py > profiles/archive/r04_shipped_T201.json"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def oracle_run(args, nf, seed, L, rir_taps, mode, device, perturb=0.0):
    """mode: fp64 | fp32.  Returns (x_den trace (T, L) float32 cpu, clean, n_draws, seconds)"""
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from oracle import ncsnpp_ref, operators_ref as O, precision, sampler_ref as S

    @contextlib.contextmanager
    def on(dev):
        prev = torch.get_default_device()
        torch.set_default_device(dev)
        try:
            yield
        finally:
            torch.set_default_device(prev)
    ctx = precision.fp64(device) if mode == "fp64" else on(device)
    t0 = time.time()
    with ctx:
        dt, dev = torch.get_default_dtype(), torch.get_default_device()
        P = ncsnpp_ref.to_torch(synth_state_dict(0, nf))
        net = lambda z, cn: ncsnpp_ref.ncsnpp_time(P, z, cn, 510, 128)
        c0 = torch.from_numpy(synth_clean(seed, L)).to(device=dev, dtype=dt)
        c0 = 0.05 * c0 / c0.std()
        if perturb:
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
        if str(dev).startswith("cuda"):
            torch.cuda.synchronize()
    return torch.stack([t[1][0] for t in tr]).float().cpu(), c0.float().cpu(), nr.k, time.time() - t0


def build_run(args, nf, seed, L, rir_taps):
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, nf).items()})
    net = net.cuda().eval()
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    ns = [NoiseStream(9000 + seed)]
    t.sampler.noise = ns
    seg, y, op, _ = t.prepare_batch([(synth_clean(seed, L), synth_rir(seed, rir_taps), f"u{seed}.wav")], blind=True, noise=ns)
    smp = t.sampler
    torch.cuda.synchronize()
    t0 = time.time()
    smp.bind(y, op, True)
    sched = smp.create_schedule()
    gam = smp.get_gamma(sched).tolist()
    x = smp.initialize_x(tuple(y.shape), "cuda", sched)          # wpe_scaled: buddy_wpe_dereverb
    torch.cuda.synchronize()
    t_init = time.time() - t0
    tl = sched.tolist()
    tr = []
    for i in range(smp.T):
        x, xd = smp.step(x, tl[i], tl[i + 1], gam[i], blind=True)
        tr.append(xd)
    torch.cuda.synchronize()
    wall = time.time() - t0
    tr = torch.stack(tr)[:, 0].cpu()
    return tr, seg[0].cpu(), ns[0].k, wall, t_init


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=64000)
    ap.add_argument("--nf", type=int, default=128)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rir_taps", type=int, default=8000)
    ap.add_argument("--T", type=int, default=None, help="override the shipped T = 201 (smoke runs only)")
    ap.add_argument("--cpu-oracle", type=int, default=0, metavar="THREADS", help="also run the fp32 oracle on this many host threads")
    a = ap.parse_args()
    from buddy_amd.config import compose
    from buddy_amd.utils.metrics import si_sdr
    ov = [f"network.nf={a.nf}"] + ([f"tester.sampling_params.T={a.T}"] if a.T else [])
    args = compose(tester="blind_dereverberation_BUDDy", overrides=ov)        # the shipped yaml: wpe_scaled, T = 201, order 1, 10 updates
    ps = args.tester.posterior_sampling
    assert ps.warm_initialization.mode == "wpe_scaled" and ps.blind_hp.op_updates_per_step == 10 and (a.T or args.tester.sampling_params.T == 201)
    sd = lambda x, y: float(si_sdr(x.double()[None], y.double()[None]))
    log = lambda m: print(m, file=sys.stderr, flush=True)

    b_tr, clean, b_k, b_wall, b_init = build_run(args, a.nf, a.seed, a.L, a.rir_taps)
    log(f"build: {b_wall:.2f} s for {b_tr.shape[0]} steps (warm start + bind {b_init * 1e3:.0f} ms)")
    runs = {}
    g_tr, g_clean, g_k, g_s = oracle_run(args, a.nf, a.seed, a.L, a.rir_taps, "fp64", "cuda")
    log(f"float64 arbiter (torch ops on the GPU): {g_s:.0f} s")
    assert g_k == b_k, "noise streams out of step"
    runs["fp64_perturbed_1e-13"] = oracle_run(args, a.nf, a.seed, a.L, a.rir_taps, "fp64", "cuda", perturb=1e-13)
    log(f"float64, input x (1 + 1e-13): {runs['fp64_perturbed_1e-13'][3]:.0f} s")
    runs["fp32_oracle_torch_gpu"] = oracle_run(args, a.nf, a.seed, a.L, a.rir_taps, "fp32", "cuda")
    log(f"fp32 oracle through torch's GPU kernels: {runs['fp32_oracle_torch_gpu'][3]:.0f} s")
    if a.cpu_oracle:
        torch.set_num_threads(a.cpu_oracle)
        runs[f"fp32_oracle_cpu_{a.cpu_oracle}_threads"] = oracle_run(args, a.nf, a.seed, a.L, a.rir_taps, "fp32", "cpu")
        log(f"fp32 oracle on {a.cpu_oracle} host threads: {runs[f'fp32_oracle_cpu_{a.cpu_oracle}_threads'][3]:.0f} s")
    T = g_tr.shape[0]
    out = {"config": {"yaml": "conf/tester/blind_dereverberation_BUDDy.yaml (unchanged)", "warm_initialization": ps.warm_initialization.mode,
                      "T": T, "order": args.tester.sampling_params.order, "op_updates_per_step": ps.blind_hp.op_updates_per_step, "L": a.L, "nf": a.nf,
                      "utterance": f"synth_clean({a.seed}), synth_rir({a.seed}, {a.rir_taps})", "weights": "synth_state_dict(0, nf) (no trained checkpoint offline)",
                      "noise": f"NoiseStream({9000 + a.seed})", "n_noise_draws": int(g_k)},
           "build": {"wall_s_incl_wpe_warm_start": b_wall, "warm_start_and_bind_ms": b_init * 1e3, "ms_per_step": (b_wall - b_init) / T * 1e3,
                     "finite": bool(torch.isfinite(b_tr).all()), "final_std": float(b_tr[-1].std()),
                     "per_step_si_sdr_vs_fp64_dB": [round(sd(b_tr[i], g_tr[i]), 1) for i in range(T)],
                     "final_si_sdr_to_clean_dB": sd(b_tr[-1], clean)},
           "fp64_arbiter": {"seconds": g_s, "final_si_sdr_to_clean_dB": sd(g_tr[-1], g_clean), "where": "oracle/ in float64 through torch ops on the GPU"}}
    for name, (tr, c, k, secs) in runs.items():
        assert k == g_k
        out[name] = {"seconds": secs, "per_step_si_sdr_vs_fp64_dB": [round(sd(tr[i], g_tr[i]), 1) for i in range(T)],
                     "final_si_sdr_to_clean_dB": sd(tr[-1], c)}
    ref32 = [v for n, v in out.items() if n.startswith("fp32_oracle")]
    b = np.minimum(np.array(out["build"]["per_step_si_sdr_vs_fp64_dB"]), 100.0)
    o = np.minimum(np.min(np.array([r["per_step_si_sdr_vs_fp64_dB"] for r in ref32]), 0), 100.0)
    out["verdict"] = {"build_minus_worst_fp32_oracle_dB_min_over_steps": float((b - o).min()), "build_minus_worst_fp32_oracle_dB_median": float(np.median(b - o)),
                      "delta_si_sdr_to_clean_build_vs_fp64_dB": out["build"]["final_si_sdr_to_clean_dB"] - out["fp64_arbiter"]["final_si_sdr_to_clean_dB"],
                      "delta_si_sdr_to_clean_fp32_oracles_vs_fp64_dB": [r["final_si_sdr_to_clean_dB"] - out["fp64_arbiter"]["final_si_sdr_to_clean_dB"] for r in ref32],
                      "note": "the shipped blind chain is chaotic (DESIGN.md section 2): after a few steps any two executions, float64 ones included, are as "
                              "far apart as two samples of the posterior; the build's distance to the float64 trajectory must be of the order of the fp32 "
                              "oracle's own and the final estimate's quality (SI-SDR to clean) statistically the same"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
