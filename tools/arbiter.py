"""fp64-arbitrated parity of full blind DPS runs.

For each utterance / noise seed u the SAME run (blind DPS, T-step schedule, order 1, 10 operator updates per step, full-width network,
identical weights / inputs / injected noise draws) is executed
  * by the CPU oracle in float64           (``oracle.precision.fp64``: the algorithm's exact trajectory for these inputs),
  * by the CPU oracle in float32 at two intra-op thread counts (the reference arithmetic, two summation orders),
  * by the MI355X build (fp32),
  * once more in float64 with the input scaled by 1 + 1e-13 ("fp64b": how far two float64 executions of this chaotic chain separate, i.e. the
    resolution of the arbiter itself),
and every execution is measured against the fp64 trajectory: per-step SI-SDR of x_den, SI-SDR of the final estimate, and the
difference of SI-SDR-to-clean.  The build passes if its deviation from the fp64 trajectory is not larger than the fp32 oracle's own.

  python tools/arbiter.py oracle --seeds 0-7 [--L 64000 --T 50 --nf 128 --workers 2 --threads 8,4]     # CPU, writes traces
  python tools/arbiter.py build  --seeds 0-7 [...]                                                       # GPU, writes traces
  python tools/arbiter.py report --seeds 0-7 [...] > profiles/archive/r02_arbiter_*.json

Traces (x_den per step, float32) live under oracle/_ref/arbiter/ (git-ignored scratch that still travels to the GPU box)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_seeds(s):
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def tdir(a):
    d = os.path.join(ROOT, "oracle", "_ref", "arbiter", f"L{a.L}_T{a.T}_nf{a.nf}_up{a.updates}")
    os.makedirs(d, exist_ok=True)
    return d


def overrides(a):
    from oracle.arbiter_runs import overrides as ov
    return ov(a.T, a.updates, a.nf)


def oracle_job(a, seed, variant):
    """one CPU run; variant = fp64 | fp32tN"""
    import numpy as np
    from oracle.arbiter_runs import run_blind
    threads = int(variant.split("t")[1]) if variant.startswith("fp32t") else a.fp64_threads
    t0 = time.time()
    is64 = variant.startswith("fp64")
    xden, c0, k = run_blind(seed, a.L, a.T, a.nf, a.updates, a.rir_taps, fp64=is64, threads=threads,
                            device=(a.fp64_device if is64 and a.fp64_device != "cpu" else None), perturb=(1e-13 if variant == "fp64b" else 0.0))
    np.savez(os.path.join(tdir(a), f"seed{seed}_{variant}.npz"), xden=xden.numpy(), clean=c0.numpy(), n_draws=k, seconds=time.time() - t0, threads=threads)
    print(f"seed {seed} {variant}: {time.time() - t0:.0f} s ({threads} threads)", flush=True)


def phase_oracle(a):
    # fp64 + the first fp32 run of every seed first, the second fp32 thread count afterwards (a partial run still yields complete pairs)
    order = [(s, v) for s in a.seeds for v in a.variants[:2]] + [(s, v) for v in a.variants[2:] for s in a.seeds]
    jobs = [(s, v) for s, v in order if v in a.run_variants and not os.path.exists(os.path.join(tdir(a), f"seed{s}_{v}.npz"))]
    running = []
    while jobs or running:
        running = [p for p in running if p.poll() is None]
        while jobs and len(running) < a.workers:
            s, v = jobs.pop(0)
            cmd = [sys.executable, os.path.abspath(__file__), "job", "--seed", str(s), "--variant", v] + a.common
            running.append(subprocess.Popen(cmd))
        time.sleep(1.0)


def phase_build(a):
    """all seeds as ONE batched run on the GPU (per-utterance semantics: row b == the B=1 run of utterance b)"""
    import numpy as np
    import torch
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream          # deterministic noise stream only
    args = compose(overrides=overrides(a))
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, a.nf).items()})
    net = net.cuda().eval()
    edm = instantiate(args.diff_params)
    for lo in range(0, len(a.seeds), a.batch):
        seeds = a.seeds[lo:lo + a.batch]
        items = [(synth_clean(s, a.L), synth_rir(s, a.rir_taps), f"u{s}.wav") for s in seeds]
        t = Tester(args, net, edm, test_set=None, device="cuda", in_training=True)
        ns = [NoiseStream(9000 + s) for s in seeds]
        t.sampler.noise = ns
        seg, y, op, _ = t.prepare_batch(items, blind=True, noise=ns)
        smp = t.sampler
        smp.bind(y, op, True)
        sched = smp.create_schedule()
        gam = smp.get_gamma(sched).tolist()
        x = smp.initialize_x(tuple(y.shape), "cuda", sched)
        tl = sched.tolist()
        tr = []
        t0 = time.time()
        for i in range(a.T):
            x, xd = smp.step(x, tl[i], tl[i + 1], gam[i], blind=True)
            tr.append(xd.cpu())
        torch.cuda.synchronize()
        print(f"build: seeds {seeds} {time.time() - t0:.1f} s", flush=True)
        tr = torch.stack(tr)                              # (T, B, L)
        for b, s in enumerate(seeds):
            np.savez(os.path.join(tdir(a), f"seed{s}_{a.build_tag}.npz"), xden=tr[:, b].numpy(), clean=seg[b].cpu().numpy(), n_draws=ns[b].k)


def verdict_from_summary(summ, build_tag="build"):
    """build vs the fp32 oracle, both measured against the fp64 trajectory (median over seeds, per step).  Anything above 100 dB is the
    fp32 round-off floor (the first step, before any feedback) and counts as equal; fp64b tells how far two float64 executions separate."""
    import numpy as np
    o32 = [v for v in summ if v.startswith("fp32")]
    if build_tag not in summ or not o32:
        return None
    b = np.minimum(np.array(summ[build_tag]["median_per_step_dB"]), 100.0)
    o = np.minimum(np.min(np.array([summ[v]["median_per_step_dB"] for v in o32]), 0), 100.0)      # the worse fp32 oracle run, per step
    out = {"build_minus_worst_fp32_oracle_median_dB_per_step": [round(float(x), 1) for x in (b - o)], "min_margin_dB": float((b - o).min()),
           "build_no_worse_than_fp32_oracle_within_3dB_at_every_step": bool(((b - o) > -3.0).all())}
    if "fp64b" in summ:
        f = np.array(summ["fp64b"]["median_per_step_dB"])
        out["steps_until_two_fp64_runs_differ_by_more_than_40dB"] = int(np.argmax(f < 40.0)) + 1 if (f < 40.0).any() else None
        out["fp64b_final_vs_fp64_dB"] = summ["fp64b"]["final_vs_fp64_dB"]["median"]
    return out


def phase_report(a):
    import numpy as np
    import torch
    from buddy_amd.utils.metrics import si_sdr
    sd = lambda x, y: float(si_sdr(torch.from_numpy(np.asarray(x, dtype=np.float64))[None], torch.from_numpy(np.asarray(y, dtype=np.float64))[None]))
    others = [v for v in a.variants if v != "fp64"] + [a.build_tag]      # incl. fp64b = the resolution of the arbiter itself
    per_seed, steps = {}, {v: [] for v in others}
    final, dclean = {v: [] for v in others}, {v: [] for v in others}
    for s in a.seeds:
        p64 = os.path.join(tdir(a), f"seed{s}_fp64.npz")
        if not os.path.exists(p64):
            continue
        g = np.load(p64)
        row = {"fp64_si_sdr_to_clean_dB": sd(g["xden"][-1], g["clean"])}
        for v in others:
            pv = os.path.join(tdir(a), f"seed{s}_{v}.npz")
            if not os.path.exists(pv):
                continue
            h = np.load(pv)
            assert int(h["n_draws"]) == int(g["n_draws"]), "noise streams out of step"
            ps = [sd(h["xden"][i], g["xden"][i]) for i in range(g["xden"].shape[0])]
            row[v] = {"per_step_si_sdr_vs_fp64_dB": [round(x, 1) for x in ps], "final_si_sdr_vs_fp64_dB": ps[-1],
                      "delta_si_sdr_to_clean_dB": sd(h["xden"][-1], g["clean"]) - row["fp64_si_sdr_to_clean_dB"]}
            steps[v].append(ps); final[v].append(ps[-1]); dclean[v].append(row[v]["delta_si_sdr_to_clean_dB"])
        per_seed[str(s)] = row
    summ = {}
    for v in others:
        if not steps[v]:
            continue
        A = np.array(steps[v])
        summ[v] = {"n_seeds": int(A.shape[0]), "median_per_step_dB": [round(float(x), 1) for x in np.median(A, 0)],
                   "min_per_step_dB": [round(float(x), 1) for x in A.min(0)],
                   "final_vs_fp64_dB": {"median": float(np.median(final[v])), "min": float(np.min(final[v])), "max": float(np.max(final[v]))},
                   "abs_delta_si_sdr_to_clean_dB": {"median": float(np.median(np.abs(dclean[v]))), "max": float(np.max(np.abs(dclean[v]))),
                                                    "mean_signed": float(np.mean(dclean[v]))}}
    verdict = verdict_from_summary(summ, a.build_tag)
    print(json.dumps({"config": {"L": a.L, "T": a.T, "nf": a.nf, "op_updates_per_step": a.updates, "rir_taps": a.rir_taps, "seeds": a.seeds,
                                 "weights": "synth_state_dict(0, nf)", "noise": "NoiseStream(9000 + seed)"},
                      "summary": summ, "verdict": verdict, "per_seed": per_seed}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("phase", choices=["oracle", "build", "report", "job"])
    ap.add_argument("--seeds", default="0-7")
    ap.add_argument("--L", type=int, default=64000)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--nf", type=int, default=128)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--rir_taps", type=int, default=8000)
    ap.add_argument("--threads", default="8,4", help="thread counts of the two fp32 oracle runs")
    ap.add_argument("--fp64_threads", type=int, default=8)
    ap.add_argument("--fp64_device", default="cpu", help="cuda: run the float64 arbiter through the same torch ops on the GPU (an hour -> under a minute per utterance)")
    ap.add_argument("--only", default="", help="comma-separated subset of variants for the oracle phase (e.g. fp64 or fp32t8,fp32t4)")
    ap.add_argument("--workers", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--build_tag", default="build")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--variant", default="fp64")
    a = ap.parse_args()
    a.seeds = parse_seeds(a.seeds)
    a.variants = ["fp64"] + [f"fp32t{t}" for t in a.threads.split(",")] + ["fp64b"]     # fp64b: fp64 with the input scaled by 1 + 1e-13
    a.run_variants = [v for v in a.variants if not a.only or v in a.only.split(",")]
    a.common = ["--L", str(a.L), "--T", str(a.T), "--nf", str(a.nf), "--updates", str(a.updates), "--rir_taps", str(a.rir_taps),
                "--fp64_threads", str(a.fp64_threads), "--fp64_device", a.fp64_device]
    if a.phase == "job":
        oracle_job(a, a.seed, a.variant)
    elif a.phase == "oracle":
        phase_oracle(a)
    elif a.phase == "build":
        phase_build(a)
    else:
        phase_report(a)
