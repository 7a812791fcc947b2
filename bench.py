#!/usr/bin/env python3
"""Benchmark of the MI355X reverse-diffusion dereverberation sampler (BASELINE.json metric).

A "step" = ONE diffusion step of the blind Euler-Heun DPS sampler (order 1: stochastic churn, one score-network
forward + input-VJP through the hand-written HIP NCSN++, 10 operator Adam updates, likelihood score, Euler update;
reference testing/EulerHeunSamplerDPS.py:115-157) over one batch of B=8 synthetic 4 s @ 16 kHz utterances per GPU --
BASELINE.json configs[1] ("Batch=8 4 s@16 kHz synthetic STFT, 50-step EulerHeun blind sampler, NCSN++ HIP on 1 MI355X").
value = utterance-diffusion-steps per second over ALL ranks = n_gpus * B * K / max-over-ranks(elapsed).

Launch: `python bench.py` (1 GPU), `python bench.py --gpus N` (re-executes itself as N ranks under torch.distributed.run) or
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (one process per GPU; utterances sharded across ranks,
no collective inside the loop, ONE all_gather of the outputs at the end of the run, outside the timed region and reported as
gather_ms; --gpus must equal WORLD_SIZE).

Instrumentation: inside the timed region HIP events bracket only the dominant kernel (`roofline`), on every other step; the per-class
attribution (`conv3x3`, `roofline_hbm`, `other_matrix_kernels`, `operator_update`) is measured in two fully instrumented steps AFTER the
timed region (`attribution_pass`) -- with ~430 event pairs per step in the timed region the step was 3 ms (2.8 %) slower.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL across processes); must be set before HIP starts

_T_PROC = time.perf_counter()
import numpy as np
import torch
_IMPORT_TORCH_S = time.perf_counter() - _T_PROC     # first import on a fresh box pages the image in (1-2 min), later ones ~1.5 s

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

U_FWD = 1.2501e12   # algorithmic FLOPs of one score-network forward for a 4 s utterance (measured with
                    # torch.utils.flop_counter on the reference); forward + input-VJP = 2 * U_FWD
PEAK_HBM_GBS = 8000.0        # MI355X HBM3E, MI355X_MICROARCH.md
GUIDE_COPY_GBS = 6290.0      # the same guide's measured float4-copy rate ("8.0 TB/s spec; 6.29 TB/s measured"): the calibrated streaming ceiling
PEAK_FP32_MFMA = 157.3   # TFLOP/s, MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA = 2500.0  # TFLOP/s, MI355X dense bf16 matrix peak (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)


SETUP = {"import_torch_s": _IMPORT_TORCH_S}     # wall times of the one-off set-up pieces of the FIRST stack (rank-local): reported as `cold_start`


def build_stack(args_ns, device, B, first_utt, net=None, tester_cfg="blind_dereverberation_BUDDy", blind=True, T=None, length=None, extra=(), attention=None,
                shipped=False):
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    # shipped: the tester yaml exactly as it ships (blind: wpe_scaled warm start, T = 201) -- no overrides of the sampler's settings
    ov = ([] if shipped else [f"tester.sampling_params.T={T or args_ns.T}"]) + list(extra)
    if tester_cfg != "only_unconditional" and not shipped:
        ov.append("tester.posterior_sampling.warm_initialization.mode=reverb_scaled")
    if getattr(args_ns, "attention", None):
        ov.append(f"+network.attention={args_ns.attention}")
    if getattr(args_ns, "gemm", None):
        ov.append(f"+network.gemm={args_ns.gemm}")
    args = compose(tester=tester_cfg, overrides=ov)
    L = length or args_ns.length
    if net is None:
        t0 = time.perf_counter()
        from buddy_amd import _lib as _l
        _l.require_gpu()
        SETUP["lib_load_s"] = time.perf_counter() - t0            # dlopen of libbuddy_hip.so: registers the gfx950 code object with the HIP runtime
        t0 = time.perf_counter()
        net = instantiate(args.network)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(0, args.network.nf).items()})
        net = net.to(device).eval()
        torch.cuda.synchronize()
        SETUP["module_build_s"] = time.perf_counter() - t0        # torch side: module construction, synthetic state dict, H2D of 111 MB
        # cold start of the library handle: buddy_ncsnpp_create (parameter upload, small packs, 1x1 images) + the first forward, which
        # prepares on the GPU the one operand form each 3x3 convolution uses at this shape
        t0 = time.perf_counter()
        with torch.no_grad():
            net(torch.zeros(B, L, device=device), torch.zeros(B, device=device))
        torch.cuda.synchronize()
        SETUP["cold_start_s"] = time.perf_counter() - t0
    else:
        net = net.replica(attention=attention)   # same prepared weights (shared, read-only), own activation arena / VJP tape
    edm = instantiate(args.diff_params)
    tester = Tester(args, net, edm, test_set=None, device=device, in_training=True)
    if tester_cfg == "only_unconditional":
        return args, net, edm, tester, None, None, None
    t0 = time.perf_counter()
    items = [(synth_clean(first_utt + u, L), synth_rir(first_utt + u, 8000), f"utt{first_utt + u}.wav") for u in range(B)]
    SETUP.setdefault("synth_inputs_s", time.perf_counter() - t0)    # numpy: synthetic clean signals and RIRs of the first stack
    torch.manual_seed(1234 + first_utt)
    t0 = time.perf_counter()
    seg, y, op, _ = tester.prepare_batch(items, blind=blind)
    torch.cuda.synchronize()
    SETUP.setdefault("prepare_batch_s", time.perf_counter() - t0)   # y = clean * RIR through the HIP FIR, the blind operator handle (first stack)
    return args, net, edm, tester, seg, y, op


class StepRunner:
    """Drives the sampler one diffusion step at a time (what predict() does in its loop)."""

    def __init__(self, tester, y, op, device, blind=True):
        s = tester.sampler
        self.blind = blind
        s.bind(y, op, blind)           # what predict_conditional does before its loop: operator, observation, losses, fresh Adam state (HIP handle)
        # config 4 of BASELINE.json asks for the operator-update share: bracket optimize_op with events on the launch stream
        self.op_events = None
        if blind:
            inner = op.hip_optimize

            def timed_optimize(x_den, t):
                if self.op_events is None:
                    return inner(x_den, t)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); r = inner(x_den, t); e1.record()
                self.op_events.append((e0, e1))
                return r
            op.hip_optimize = timed_optimize
        self.s = s
        t = s.create_schedule()                            # host-side schedule, as in predict(): no device sync inside a step
        self.t, self.gamma = t.tolist(), s.get_gamma(t).tolist()
        self.x = s.initialize_x(tuple(y.shape), device, t)
        self.i = 0
        self.x_den = None

    def step(self):
        s = self.s
        if self.i >= s.T - 1:           # stay inside the schedule (the last step has t_{i+1} = 0): restart the trajectory
            self.i = 0
        self.x, self.x_den = s.step(self.x, self.t[self.i], self.t[self.i + 1], self.gamma[self.i], blind=self.blind)
        self.i += 1


class UncondRunner:
    """Steps of the unconditional Euler-Heun sampler (reference testing/EulerHeunSampler.py:47-72): score-network FORWARD evaluations only."""

    def __init__(self, tester, B, L, device):
        s = tester.sampler
        self.s = s
        t = s.create_schedule()
        self.t, self.gamma = t.tolist(), s.get_gamma(t).tolist()
        self.x = s.initialize_x((B, L), device, t)
        self.i = 0
        self.x_den = None

    def step(self):
        s = self.s
        if self.i >= s.T - 1:
            self.i = 0
        self.x, self.x_den = s.step(self.x, self.t[self.i], self.t[self.i + 1], self.gamma[self.i])
        self.i += 1


def cpu_baseline(length, n_threads, reps, utt=0, mode="blind"):
    """The CPU oracle (oracle/, reference-faithful restatement, torch fp32) on the host cores for ONE utterance -- the same unit of work as
    the GPU metric.  mode "blind": blind DPS steps (order 1, 10 operator updates; T=50 schedule); "informed": informed DPS steps (order 2 = two
    forward+VJP evaluations per step; T=10 schedule, BASELINE configs[0] shape).  1 warm-up step, then `reps` timed steps (per-step seconds)."""
    from buddy_amd.config import compose
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from oracle import ncsnpp_ref, operators_ref as O, sampler_ref as S
    torch.set_num_threads(n_threads)
    blind = mode == "blind"
    args = compose(tester="blind_dereverberation_BUDDy" if blind else "informed_dereverberation_DPS",
                   overrides=[f"tester.sampling_params.T={50 if blind else 10}", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled"])
    P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
    net = lambda x, cn: ncsnpp_ref.ncsnpp_time(P, x, cn, 510, 128)
    ns = S.NoiseStream(1 + utt)
    smp = S.EulerHeunDPSRef(net, S.EDMRef(args.diff_params.sde_hp), args, ns)
    op_hp = args.tester.informed_dereverberation.op_hp
    clean, rir = torch.from_numpy(synth_clean(utt, length)), torch.from_numpy(synth_rir(utt, 8000))
    op_ref = O.RIROperatorRef(op_hp)
    op_ref.update_params(rir)
    y = op_ref.degradation(clean[None])
    op = op_ref
    smp.rec_loss = O.get_loss_ref(args.tester.posterior_sampling.rec_loss, op_ref)
    if blind:
        op = O.BlindSubbandFilteringRef(op_hp, 16000, ns)
        op.update_H(use_noise=True, noise=ns)
        hp = args.tester.posterior_sampling.blind_hp
        smp.rec_loss = O.get_loss_ref(args.tester.posterior_sampling.rec_loss, op)
        smp.rec_loss_params = O.get_loss_ref(args.tester.posterior_sampling.rec_loss_params, op)
        smp.optim = torch.optim.Adam(op.params + op.params_phases, lr=hp.lr_op, weight_decay=hp.weight_decay, betas=(hp.beta1, hp.beta2))
        smp.rir_reg_loss = O.get_loss_ref(args.tester.posterior_sampling.RIR_noise_regularization.loss, op)
    smp.operator, smp.y = op, y
    t = S.create_schedule(smp.sde_hp, smp.T)
    gamma = S.get_gamma(t, smp.sp)
    x = smp.initialize_x(y.shape, t)
    times = []
    for i in range(1 + reps):
        t0 = time.time()
        x, _ = smp.step(x, t[i], t[i + 1], gamma[i], blind)
        times.append(time.time() - t0)
    return {"step_seconds": times[1:], "warmup_seconds": times[0], "threads": n_threads, "mode": mode}


def _effective_cores():
    """logical CPUs this process may actually use: the cgroup CPU quota (cpu.max) can be far below os.cpu_count() on shared hosts"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return n


def _cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def run_cpu_baseline(length, reps=3):
    """CPU leg: the oracle in child processes without a GPU context -- B=1 (one utterance on min(cores, 32) threads) and
    B=8 (eight independent utterances side by side, cores/8 threads each), 1 warm-up + `reps` timed steps each, median; bounded by hard
    timeouts so the bench always finishes in minutes."""
    import subprocess
    cores = _effective_cores()
    out = {"value": None, "unit": "utterance-steps/s", "cores": None, "kind": "port", "cpu_model": _cpu_model(), "host_logical_cores": os.cpu_count(),
           "usable_cores": cores}

    def spawn(threads, utt, mode="blind", nreps=reps):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", str(threads), "--length", str(length),
                                 "--cpu-reps", str(nreps), "--cpu-utt", str(utt), "--cpu-mode", mode], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                text=True, env=env)

    def collect(procs, timeout):
        res = []
        t_end = time.time() + timeout
        for pr in procs:
            try:
                so, _ = pr.communicate(timeout=max(1.0, t_end - time.time()))
                res.append(json.loads([l for l in so.splitlines() if l.startswith("{")][-1]))
            except Exception:
                pr.kill()
                res.append(None)
        return res

    t1 = min(cores, 32)            # oneDNN/OpenMP stop scaling (and can crawl) far beyond this on big hosts
    r1 = collect([spawn(t1, 0)], 240)[0]
    if r1:
        med = float(np.median(r1["step_seconds"]))
        out.update(value=1.0 / med, cores=t1)
        out["B1"] = {"utterance_steps_per_s": 1.0 / med, "threads": t1, "step_seconds": r1["step_seconds"], "warmup_seconds": r1["warmup_seconds"]}
    t8 = max(1, min(32, min(cores, 32 if cores < 64 else cores) // 8))     # B8 shares the same core budget as B1 unless the host really has >= 64 usable cores
    r8 = collect([spawn(t8, u) for u in range(8)], 300)
    if all(r8):
        per_step = np.max(np.array([r["step_seconds"] for r in r8]), axis=0)       # a batch step is done when its slowest utterance is
        med8 = float(np.median(per_step))
        out["B8"] = {"utterance_steps_per_s": 8.0 / med8, "threads": 8 * t8, "batch_step_seconds": [float(v) for v in per_step]}
        if out["value"] is None or 8.0 / med8 > out["value"]:
            out.update(value=8.0 / med8, cores=8 * t8)
    # the informed order-2 step: one utterance, two forward+VJP evaluations per step, no operator update
    ri = collect([spawn(t1, 0, "informed", 2)], 180)[0]
    if ri:
        medi = float(np.median(ri["step_seconds"]))
        out["informed_order2_B1"] = {"utterance_steps_per_s": 1.0 / medi, "score_evals_per_s": 2.0 / medi, "threads": t1, "step_seconds": ri["step_seconds"],
                                     "warmup_seconds": ri["warmup_seconds"], "what": "informed DPS, order 2, T=10 schedule: 1 warm-up + 2 timed steps, median"}
    out["sample"] = (f"oracle/ (torch fp32 restatement of the reference, parity-pinned): blind DPS steps (order 1, 10 operator updates) of {length / 16000:g} s "
                     f"utterances, 1 warm-up + {reps} timed steps, median; B1 = one utterance on {t1} threads, B8 = eight utterances side by side on "
                     f"{t8} threads each; value = the better of the two")
    if out["value"] is None:
        out["sample"] = "CPU leg failed / timed out"
    return out


def measure_peaks(lib, device):
    """On-box ceilings next to the nominal ones: a pure fp32-MFMA loop (no memory traffic) and plain HBM streaming kernels."""
    from buddy_amd import _lib
    out = {}
    try:
        blocks, iters = 256 * 8, 4096
        seed = torch.randn(1024, device=device); o = torch.empty(blocks * 256, device=device)
        clk = torch.zeros(2, dtype=torch.int64, device=device)
        S = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.buddy_mfma_ubench(seed.data_ptr(), o.data_ptr(), blocks, 64, clk.data_ptr(), S))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(lib.buddy_mfma_ubench(seed.data_ptr(), o.data_ptr(), blocks, iters, clk.data_ptr(), S)); e1.record()
        torch.cuda.synchronize()
        out["fp32_mfma_tflops"] = blocks * 4 * 4 * iters * 4096 / (e0.elapsed_time(e1) * 1e-3) / 1e12
        _lib.check(lib.buddy_mfma_ubench_bf16(seed.data_ptr(), o.data_ptr(), blocks, 64, clk.data_ptr(), S))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(lib.buddy_mfma_ubench_bf16(seed.data_ptr(), o.data_ptr(), blocks, iters, clk.data_ptr(), S)); e1.record()
        torch.cuda.synchronize()
        out["bf16_mfma_tflops"] = blocks * 4 * 12 * iters * 32768 / (e0.elapsed_time(e1) * 1e-3) / 1e12     # random operand bits, no memory traffic
    except Exception as e:
        out["fp32_mfma_tflops"] = None; out["mfma_error"] = str(e)[:100]
    # HBM: the library's own float4 streaming kernels (buddy_hbm_ubench: eight 16-byte requests in flight per thread, grid sweep, plain and non-temporal;
    # round 6 -- rounds 1-5 quoted torch's copy / sum kernels here, whose read-only figure was below their copy figure: a microbenchmark
    # artefact).  hbm_copy_GBps (read + write, the transforms' and the GEMM's own mix) is the calibrated denominator of every HBM-bound roofline.
    n = 256 * 2 ** 20
    x = torch.randn(n, device=device); y = torch.empty_like(x)
    S = torch.cuda.current_stream().cuda_stream

    def rate(fn, nbytes, reps=6):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
    for mode, name, moved in ((0, "hbm_copy_GBps", 8 * n), (1, "hbm_read_GBps", 4 * n), (2, "hbm_write_GBps", 4 * n)):
        best = 0.0
        for nt in (0, 1):
            for blocks in (-1, -2, -4, 512, 16384):
                best = max(best, rate(lambda: _lib.check(lib.buddy_hbm_ubench(x.data_ptr(), y.data_ptr(), 4 * n, mode, nt, blocks, S)), moved))
        out[name] = best
    out["hbm_torch_copy_GBps"] = rate(lambda: y.copy_(x), 8 * n)
    out["hbm_torch_add_GBps"] = rate(lambda: torch.add(x, 1.0, out=y), 8 * n)
    out["hbm_copy_guide_GBps"] = GUIDE_COPY_GBS
    out["hbm_note"] = ("hbm_copy / hbm_read / hbm_write: best of the library's float4 streaming kernel (buddy_hbm_ubench; plain and non-temporal; one-shot grids of 1 / 2 / "
                       "4 words per thread and grid-stride forms of 512 / 16384 workgroups) over 1 GiB on THIS box; hbm_copy_guide = the 6.29 TB/s MI355X_MICROARCH.md records for a float4 copy (not reached on "
                       "the boxes of this pool: profiles/r06_hbm_ubench.json)")
    del x, y
    return out


def f16x2_roofline(gemm_tf, alg_bytes, ms_total, peaks):
    """roofline object of the dominant kernel in f16x2 arithmetic (the default): the level-0 launches (K = 128: 56 % of the FLOPs) are HBM-bound on the V read +
    M write of the three-pass form, the 256-channel ones matrix-bound; `bound` is the side with the larger fraction of its nominal peak, both are in the object."""
    hbm = alg_bytes / (ms_total * 1e-3) / 1e9 if ms_total > 0 else 0.0
    mf = 3.0 * gemm_tf
    side_h = {"achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS}
    side_m = {"achieved": mf, "peak": PEAK_BF16_MFMA, "unit": "TFLOP/s", "frac": mf / PEAK_BF16_MFMA}
    use_h = side_h["frac"] >= side_m["frac"]
    r = {"bound": "hbm" if use_h else "mfma",
         "kernel": "wgemm_f16x2_kernel -- the batched GEMMs M[pos] = V[pos] U[pos]^T of the three-pass Winograd F(6x6,3x3) convolutions (64 positions; 94 % of the "
                   "network's algorithmic FLOPs) in f16x2 arithmetic: both fp32 operands scaled by a power of two and split into two f16 terms, three "
                   "v_mfma_f32_32x32x16_f16 products per 16 k, fp32 accumulate"}
    r.update(side_h if use_h else side_m)
    r["achieved_note"] = ("ALGORITHMIC bytes per launch (V read once + M written once + weights once = 4 * positions * tiles * (Cin + Cout) + the stage images) / "
                          "average launch duration" if use_h else
                          "EXECUTED f16 MFMA FLOPs per launch (3 x 2 * positions * tiles * Cin * Cout) / average launch duration") + \
                         ", HIP events on the launch stream inside the timed region (every launch of every other step)"
    r["mfma_side" if use_h else "hbm_side"] = dict(side_m if use_h else side_h, note=(
        "executed f16 MFMA FLOPs (3 x 2 * positions * tiles * Cin * Cout) / time against the 2.5 PFLOP/s dense f16 peak" if use_h else
        "algorithmic bytes (V read once, M written once, weights once) / time"))
    r["peak_measured_on_box"] = peaks.get("hbm_copy_GBps") if use_h else peaks.get("bf16_mfma_tflops")
    r["frac_of_measured_peak"] = (r["achieved"] / r["peak_measured_on_box"]) if r["peak_measured_on_box"] else None
    if use_h:
        r["frac_of_guide_copy"] = r["achieved"] / GUIDE_COPY_GBS       # against the 6.29 TB/s float4 copy of MI355X_MICROARCH.md
    r["peak_measured_note"] = ("the library's float4 copy kernel on this box (buddy_hbm_ubench: read + write, the GEMM's own mix; best of plain / non-temporal over a grid sweep)" if use_h else
                               "a pure 16-bit MFMA loop on random operand bits on this box: the chip clocks to its power budget")
    r["fp32_equivalent_tflops"] = gemm_tf
    r["frac_of_fp32_matrix_peak"] = gemm_tf / PEAK_FP32_MFMA
    return r


def conv_source_stamp():
    """sha1 over the sources of the dominant kernel group: a PMC summary is only quoted if it was measured on these exact kernels"""
    import hashlib
    h = hashlib.sha1()
    for f in ("igemm.hip", "wgemm.hip", "wino4.hip", "wino6.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "buddy_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--length", type=int, default=64000)
    ap.add_argument("--T", type=int, default=50)
    ap.add_argument("--sub-batches", type=int, default=1, help="sample the B utterances of a GPU as this many concurrent sub-batches on their own HIP "
                    "streams in the MAIN timed region (buddy_amd/testing/concurrent.py).  Default 1: one batch, one stream, so that per-kernel event "
                    "times and the rocprofv3 summary attribute cleanly (co-running kernels stretch each other's durations)")
    ap.add_argument("--also-concurrent", type=int, default=2, help="after the main region, time the same workload as this many concurrent sub-batches "
                    "in a second region and report it as `concurrent_sub_batches` (0 = skip)")
    ap.add_argument("--attention", default=None, choices=["auto", "flash", "matrix", "bf16", "f16"], help="attention core of the network (default: the library's, "
                    "auto: fp32, materialised for T <= 4096, online softmax beyond; bf16 / f16 = the opt-in fast mode that passed the 0.1 dB gate, profiles/archive/r02_attention_modes.json)")
    ap.add_argument("--gemm", default=None, choices=["bf16x3", "fp32", "f16x2"], help="arithmetic of the Winograd-domain GEMMs (default: the library's, f16x2; bf16x3 = exact "
                    "three-way bf16 split of the fp32 operands, six bf16 MFMA products, fp32 accumulate; fp32 = v_mfma_f32_32x32x2_f32, the reference run)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rccl-selftest", action="store_true", help="skip the one-rank RCCL bring-up at the end of an N=1 run")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU smoke tests of the N>1 path)")
    ap.add_argument("--cpu-baseline-only", type=int, default=0, metavar="THREADS", help="internal: run only the CPU leg and print its JSON")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--cpu-utt", type=int, default=0)
    ap.add_argument("--cpu-mode", default="blind", choices=["blind", "informed"])
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (--backend, default nccl = RCCL) even with one rank and run the "
                    "end-of-run gather through it: exercises communicator set-up and the collective on a 1-GPU box")
    ap.add_argument("--legs", default="auto", help="extra untimed-from-`value` legs (BASELINE.md section 3): comma list of informed,informed_b1,blind_b1,"
                    "forward_only,longform,full_run,shipped,gemm_modes,realclip or 'all' / 'none'; auto = all at N=1 with the default workload, full_run only otherwise")
    a = ap.parse_args()
    if a.cpu_baseline_only:
        print(json.dumps(cpu_baseline(a.length, a.cpu_baseline_only, a.cpu_reps, a.cpu_utt, a.cpu_mode)))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher -- one rank per GPU under torch.distributed.run (RCCL), same
        # arguments; rank 0 of the children prints the ONE JSON line, which passes through this process's stdout.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    # stdout carries exactly ONE line (the JSON).  RCCL prints a version banner to the C-level stdout when a communicator comes up, so file
    # descriptor 1 points at stderr for the whole run and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the sampler path has no CPU fallback")
    if world > 1 and a.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {torch.cuda.device_count()} (--backend gloo lets ranks share a GPU for smoke tests)")
    from buddy_amd import dist as bdist_
    dev_index = bdist_.device_index(local_rank)            # == local_rank on a real node (visible device r of the launcher's list); gloo smoke tests wrap around one GPU
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # each rank drives ~2 000 launches per step from one Python thread: its own cores, next to its GPU (sysfs numa_node / local_cpulist), disjoint from its peers'
    placement = bdist_.pin_rank(local_rank, local_world, torch.cuda.device_count()) if world > 1 else {"cpus": None, "numa_cpus": None, "physical_device": None}
    t0 = time.perf_counter()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    torch.cuda.init()
    torch.zeros(1, device=device).add_(1.0)                # HIP context + torch's own code objects + the first kernel launch of the process
    torch.cuda.synchronize()
    SETUP["hip_context_s"] = time.perf_counter() - t0
    dist = None
    dist_init_ms = 0.0
    rccl_env = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        rccl_env = bdist_.rccl_env_defaults()          # before the process group exists: dmabuf IPC, the RCCL version banner (NCCL_DEBUG=VERSION)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                   # --force-dist on one rank: a private rendezvous
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
        t0 = time.perf_counter()
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
        dist_init_ms = (time.perf_counter() - t0) * 1e3
    coll_dev = device if a.backend == "nccl" else torch.device("cpu")

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)
    placements = [placement]
    if dist is not None:
        placements = [None] * world
        dist.all_gather_object(placements, placement)
        log("rccl / device environment: " + ", ".join(f"{k}={v}" for k, v in rccl_env.items() if v is not None))
        for r_, p_ in enumerate(placements):
            log(f"rank {r_}: physical device {p_['physical_device']}, cpus {p_['cpus']} (GPU-local NUMA cpus: {p_['numa_cpus']})")

    from buddy_amd import _lib
    lib = _lib.require_gpu()
    B = a.batch
    net0 = [None]

    def make_runners(S):
        rr, ss = [], []
        for k in range(S):
            st = torch.cuda.Stream() if S > 1 else torch.cuda.current_stream()
            with torch.cuda.stream(st):
                _, net, _, tester, _, y, op = build_stack(a, device, B // S, rank * B + k * (B // S), net0[0])
                net0[0] = net0[0] or net
                rr.append(StepRunner(tester, y, op, device))
            ss.append(st)
        return rr, ss

    class _All:                      # one diffusion step of the whole batch = one step of every sub-batch, issued from this host thread
        def __init__(self, rr, ss):
            self.rr, self.ss = rr, ss

        def step(self):
            for r, st in zip(self.rr, self.ss):
                with torch.cuda.stream(st):
                    r.step()

    S = a.sub_batches if (a.sub_batches > 1 and B % a.sub_batches == 0) else 1
    log(f"building stack: B={B}/GPU as {S} sub-batch(es), L={a.length}, world={world}")
    t_sb = time.perf_counter()
    runs, streams = make_runners(S)
    torch.cuda.synchronize()
    stack_build_s = time.perf_counter() - t_sb
    run = _All(runs, streams)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # every set-up piece goes to stderr too (the driver keeps the tail of stderr; the JSON line's `cold_start` object lies beyond its 2000 characters)
    log("stack ready: " + ", ".join(f"{k} {v:.2f}" for k, v in SETUP.items()) + f", stack_build_s {stack_build_s:.2f} (= lib_load + module_build + cold_start "
        "[create + first forward incl. the arena hipMalloc] + synth_inputs + prepare_batch + Tester/replica objects)")
    log("warmup")
    first_step_ms = None
    for k in range(a.warmup):
        t0 = time.perf_counter()
        run.step()
        if k == 0:                       # first sampler step of the process: arena sizing dry run, data-gradient weight forms, operator graph capture
            torch.cuda.synchronize()
            first_step_ms = (time.perf_counter() - t0) * 1e3
        log("warmup step done")
    barrier()
    # Timed region.  A HIP event pair around a kernel costs a dispatch bubble (~7 us: the next kernel's launch is no longer overlapped with the
    # tail of the previous one); with every instrumented class bracketed (~430 pairs per step) that was 3 ms of a 109 ms step.  So inside the
    # timed region only the DOMINANT kernel is bracketed (the roofline), and only on every other step -- all 80 of its launches on those steps,
    # so every layer shape is sampled equally --; the per-class attribution (convolution passes, GroupNorm groups, other GEMMs, operator update)
    # comes from a separate, untimed pass of ATTR_STEPS fully instrumented steps right after.
    log("timed region")
    dom_level = int(os.environ.get("BUDDY_BENCH_PROF", "1"))
    t0 = time.perf_counter()
    sampled_steps = 0
    for i in range(a.steps):
        on = dom_level if i % 2 == 0 else 0
        lib.buddy_prof_enable(on)
        sampled_steps += 1 if on else 0
        run.step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.buddy_prof_enable(0)
    log(f"timed region done: {elapsed:.3f} s")
    dom_ms = (C.c_double * 3)(); dom_fl, dom_bi, dom_bo, dom_bg, dom_n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_longlong()
    _lib.check(lib.buddy_prof_collect_wino4(dom_ms, C.byref(dom_fl), C.byref(dom_bi), C.byref(dom_bo), C.byref(dom_bg), C.byref(dom_n)))
    el = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
    per_rank_ms = [elapsed / a.steps * 1e3]
    if dist is not None:
        each = [torch.empty_like(el) for _ in range(world)]
        dist.all_gather(each, el.clone())
        per_rank_ms = [float(e.item()) / a.steps * 1e3 for e in each]          # the spread `ms_per_step` is the maximum of
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    weight_store_main = net0[0].weight_bytes()       # before the legs: the other arithmetics' legs add their own operand forms to the shared store
    ATTR_STEPS = 2
    log(f"attribution pass: {ATTR_STEPS} fully instrumented steps (untimed)")
    lib.buddy_prof_enable(2)
    for r in runs:
        r.op_events = []
    t0 = time.perf_counter()
    for _ in range(ATTR_STEPS):
        run.step()
    barrier()
    attr_elapsed = time.perf_counter() - t0
    lib.buddy_prof_enable(0)
    op_ms = sum(e0.elapsed_time(e1) for r in runs for e0, e1 in r.op_events)
    for r in runs:
        r.op_events = None
    ms = (C.c_double * 2)(); fl = (C.c_double * 2)(); ln = (C.c_longlong * 2)(); by = (C.c_double * 2)(); xf = (C.c_double * 2)()
    _lib.check(lib.buddy_prof_collect(ms, fl, ln, by, xf))
    w4_ms = (C.c_double * 3)(); w4_fl, w4_bi, w4_bo, w4_bg, w4_n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_longlong()
    _lib.check(lib.buddy_prof_collect_wino4(w4_ms, C.byref(w4_fl), C.byref(w4_bi), C.byref(w4_bo), C.byref(w4_bg), C.byref(w4_n)))
    hb_ms, hb_by, hb_n = C.c_double(), C.c_double(), C.c_longlong()
    _lib.check(lib.buddy_prof_collect_hbm(C.byref(hb_ms), C.byref(hb_by), C.byref(hb_n)))

    # second region: the same workload as concurrent sub-batches (identical per-utterance results, better occupancy; not used for the rooflines)
    conc = None
    S2 = a.also_concurrent
    if S2 > 1 and S == 1 and B % S2 == 0:
        log(f"second region: {S2} concurrent sub-batches")
        runs2, streams2 = make_runners(S2)
        run2 = _All(runs2, streams2)
        for _ in range(a.warmup):
            run2.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run2.step()
        barrier()
        e2 = torch.tensor([time.perf_counter() - t0], device=coll_dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        e2 = float(e2.item())
        assert all(torch.isfinite(r.x_den).all() for r in runs2), "sampler diverged"
        conc = {"sub_batches_per_gpu": S2, "value": world * B * a.steps / e2, "unit": "utterance-steps/s", "ms_per_step": e2 / a.steps * 1e3,
                "what": "the same B utterances per GPU sampled as concurrent sub-batches on their own HIP streams (buddy_amd/testing/concurrent.py, "
                        "`tester.sub_batches` / `bench.py --sub-batches`): the latency-bound operator update and bottleneck layers of one sub-batch run "
                        "beside the large kernels of the other; separate timed region, same steps / warm-up"}
        del runs2, run2

    # ---- further legs, each its own untimed-from-`value` region on a replica of the same network -------------------------
    default_workload = (B == 8 and a.length == 64000)
    ALL_LEGS = {"informed", "informed_b1", "blind_b1", "forward_only", "longform", "full_run", "shipped", "gemm_modes", "realclip"}
    gemm_mode_now = {0: "fp32", 1: "bf16x3", 2: "f16x2"}[int(net0[0].get_option("gemm"))]
    if a.legs == "auto":
        want = set(ALL_LEGS) if (world == 1 and default_workload) else {"full_run"}
    elif a.legs == "all":
        want = set(ALL_LEGS)
    else:
        want = {w for w in a.legs.split(",") if w and w != "none"}
    legs = {}

    def time_leg(name, runner, units_per_step, n_steps, n_warm, extra):
        for _ in range(n_warm):
            runner.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            runner.step()
        barrier()
        e = torch.tensor([time.perf_counter() - t0], device=coll_dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
        e = float(e.item())
        assert torch.isfinite(runner.x_den).all(), f"leg {name} diverged"
        legs[name] = dict({"ms_per_step": e / n_steps * 1e3, "value": world * units_per_step * n_steps / e, "unit": "utterance-steps/s", "steps": n_steps,
                           "warmup": n_warm}, **extra)
        log(f"leg {name}: {legs[name]['ms_per_step']:.2f} ms/step")

    def stack_runner(tester_cfg, Bl, blind, T, length=None, extra=(), attention=None, gemm=None):
        _, net_l, _, tester, _, y, op = build_stack(a, device, Bl, rank * Bl, net0[0], tester_cfg=tester_cfg, blind=blind, T=T, length=length, extra=extra,
                                                    attention=attention)
        if gemm is not None:             # per-handle option of the replica: the other arithmetics' operand forms are prepared on first use (shared store)
            net_l.set_option("gemm", {"fp32": 0, "bf16x3": 1, "f16x2": 2}[gemm])
        return StepRunner(tester, y, op, device, blind=blind)

    if "informed" in want:       # (ii) of BASELINE.md section 3: informed sampler, order 2 (two forward+VJP evaluations per step), T=10 schedule
        r_ = stack_runner("informed_dereverberation_DPS", B, False, 10)
        time_leg("informed_order2", r_, B, 6, 2, {"config": f"informed DPS (known RIR, 8000 taps), order 2, T=10 schedule, B={B} x {a.length} samples "
                                                          "(conf/tester/informed_dereverberation_DPS.yaml; BASELINE configs[0] is its B=1 real-clip case)",
                                                "score_evals_per_step": 2})
        legs["informed_order2"]["score_evals_per_s"] = 2 * legs["informed_order2"]["value"]
        del r_
    if "gemm_modes" in want and gemm_mode_now == "f16x2":
        # the exact-arithmetic numbers under the driver -- the SAME workload as `value` (blind, B utterances, order 1, 10 updates) with the
        # Winograd-domain GEMMs as exact three-way bf16 splits (six products: the fp32 kernel's accuracy) and on the fp32 MFMA (bit-exact FMA chains)
        for gm in ("bf16x3", "fp32"):
            r_ = stack_runner("blind_dereverberation_BUDDy", B, True, a.T, gemm=gm)
            time_leg(f"gemm_{gm}", r_, B, 6, 2, {"config": f"the headline workload (blind step, B={B} x {a.length}) with the Winograd-domain GEMMs in {gm} arithmetic "
                                                           "(per-handle option `gemm`; default f16x2 = `value`)", "gemm": gm})
            del r_
            torch.cuda.empty_cache()
    if "realclip" in want:       # BASELINE configs[0]'s own geometry: the audio_examples clip has 133 829 samples -> 1 056 frames, 132 rows at the lowest level
        r_ = stack_runner("informed_dereverberation_DPS", 1, False, 10, length=133829)
        time_leg("informed_order2_B1_L133829", r_, 1, 6, 2, {"config": "informed DPS, order 2, T=10 schedule, ONE utterance of 133 829 samples = the length of "
                                                                      "audio_examples/clean/p226 (BASELINE configs[0]; reference conf/tester/informed_dereverberation_DPS.yaml:24), "
                                                                      "synthetic signal and weights: a geometry that is not a power of two (1 056 frames; tiles overhang on "
                                                                      "every level)", "score_evals_per_step": 2, "length": 133829})
        del r_
        torch.cuda.empty_cache()
    if "informed_b1" in want:    # the reference's own shape: one utterance at a time (testing/tester.py:132-153)
        r_ = stack_runner("informed_dereverberation_DPS", 1, False, 10)
        time_leg("informed_order2_B1", r_, 1, 6, 2, {"config": f"as informed_order2 with B=1 (latency of one utterance; BASELINE configs[0] shape)", "score_evals_per_step": 2})
        del r_
    if "blind_b1" in want:
        r_ = stack_runner("blind_dereverberation_BUDDy", 1, True, a.T)
        time_leg("blind_B1", r_, 1, 6, 2, {"config": f"the headline blind step with B=1 x {a.length} samples: per-utterance latency, the reference's own shape "
                                                     "(testing/tester.py:132-153 samples one utterance at a time)",
                                           "attention": a.attention or os.environ.get("BUDDY_ATTN", "auto (materialised form at this length)")})
        del r_
        if not a.attention and not os.environ.get("BUDDY_ATTN"):
            # the attention default is a function of T alone (rows must not depend on their batch): the materialised form it picks at 4 s wins at B = 8 and
            # loses at B = 1 (four of its six products have 32 output tiles per utterance); +network.attention=flash is the single-utterance setting
            r_ = stack_runner("blind_dereverberation_BUDDy", 1, True, a.T, attention="flash")
            time_leg("blind_B1_flash", r_, 1, 6, 2, {"config": "as blind_B1 with attention=flash (online-softmax kernels, 8-way loop split): the single-utterance latency setting",
                                                     "attention": "flash"})
            del r_
    if "forward_only" in want:   # score-network forward evaluations only (unconditional Euler-Heun sampler, order 2: two forwards per step)
        _, _, _, tester_u, _, _, _ = build_stack(a, device, B, 0, net0[0], tester_cfg="only_unconditional", T=a.T)
        r_ = UncondRunner(tester_u, B, a.length, device)
        time_leg("forward_only", r_, B, 6, 2, {"config": f"unconditional Euler-Heun sampler (reference testing/EulerHeunSampler.py:47-72), order 2, B={B} x {a.length}: "
                                                         "two score-network FORWARD evaluations per utterance-step, no VJP, no operator", "score_evals_per_step": 2})
        legs["forward_only"]["forward_evals_per_s"] = 2 * legs["forward_only"]["value"]
        del r_, tester_u
    if "longform" in want:       # BASELINE configs[4], one-GPU slice: B=4 x 30 s un-chunked, fp32 flash attention unless --attention says otherwise
        r_ = stack_runner("blind_dereverberation_BUDDy", 4, True, a.T, length=480000)
        time_leg("longform_480000_B4", r_, 4, 3, 1, {"config": "blind step, B=4 x 480000 samples (30 s@16 kHz), un-chunked (BASELINE configs[4], one-GPU slice of "
                                                               "batch=32 across 8)", "attention": a.attention or os.environ.get("BUDDY_ATTN", "auto (fp32; T = 15 008 > 4096: online-softmax kernels)"),
                                                    "value_in_4s_units": None})
        legs["longform_480000_B4"]["value_in_4s_units"] = legs["longform_480000_B4"]["value"] * 7.5
        del r_
        torch.cuda.empty_cache()
        if not a.attention:      # ... and as configs[4] words it: "fp16 MFMA attention path" (opt-in fast mode, DESIGN.md section 4.3: inside the 0.1 dB gate)
            r_ = stack_runner("blind_dereverberation_BUDDy", 4, True, a.T, length=480000, attention="f16")
            time_leg("longform_480000_B4_f16", r_, 4, 3, 1, {"config": "as longform_480000_B4 with the attention kernels on f16 MFMA operands (fp32 accumulation and softmax "
                                                                       "statistics) -- the 'fp16 MFMA attention path' BASELINE configs[4] names; everything else fp32", "attention": "f16"})
            legs["longform_480000_B4_f16"]["value_in_4s_units"] = legs["longform_480000_B4_f16"]["value"] * 7.5
            del r_
            torch.cuda.empty_cache()

    if "shipped" in want:        # the reference's own shape (testing/tester.py:132-153, test_blind_dereverberation.sh:18): ONE utterance, the shipped yaml unchanged
        t0 = time.perf_counter()
        args_s, _, _, tester_s, _, y_s, op_s = build_stack(a, device, 1, rank, net0[0], shipped=True)
        ps_ = args_s.tester.posterior_sampling
        assert ps_.warm_initialization.mode == "wpe_scaled" and args_s.tester.sampling_params.T == 201 and ps_.blind_hp.op_updates_per_step == 10
        assert tester_s.sampler.noise is None            # torch RNG, no injected streams
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
        barrier()
        t0 = time.perf_counter()
        pred_s = tester_s.sampler.predict_conditional(y_s, op_s, shape=(1, a.length), blind=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        assert torch.isfinite(pred_s).all(), "shipped run diverged"
        legs["full_run_B1_shipped"] = {"wall_s": wall, "T": 201, "ms_per_step": wall / 201 * 1e3, "value": world * 201 / wall, "unit": "utterance-steps/s",
                                       "setup_s": setup_s, "length": a.length,
                                       "config": "conf/tester/blind_dereverberation_BUDDy.yaml UNCHANGED (wpe_scaled warm start, T = 201, order 1, 10 operator updates per "
                                                 "step), ONE utterance, torch RNG: Sampler.predict_conditional wall time incl. bind, the WPE warm start on the GPU and "
                                                 "the final sync; setup_s = prepare_batch + replica handle before it"}
        log(f"leg full_run_B1_shipped: {wall:.2f} s for T=201 = {wall / 201 * 1e3:.2f} ms/step (set-up {setup_s:.2f} s)")
        del tester_s, y_s, op_s, pred_s

    # ---- one REAL run end to end: predict_conditional over the whole T-step schedule, init included; then the end-of-run gather ---------------
    full_run = None
    if "full_run" in want:
        t0 = time.perf_counter()
        _, net_f, _, tester_f, _, y_f, op_f = build_stack(a, device, B, rank * B, net0[0])
        arena_b = net_f.arena_bytes(B, a.length, True)     # a harness keeps its network handle across batches: the activation arena (hipMalloc of
        torch.cuda.synchronize()                           # tens of GB, 0.1 ... 3 s depending on the box's memory state) is set-up, not run
        setup_s = time.perf_counter() - t0
        barrier()
        t0 = time.perf_counter()
        pred_f = tester_f.sampler.predict_conditional(y_f, op_f, shape=(B, a.length), blind=True)
        torch.cuda.synchronize()
        run_s = time.perf_counter() - t0
        fr = torch.tensor([run_s], device=coll_dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(fr, op=dist.ReduceOp.MAX)
        assert torch.isfinite(pred_f).all(), "full run diverged"
        full_run = {"T": a.T, "batch_per_gpu": B, "wall_s": float(fr.item()), "operator_and_batch_setup_s": setup_s, "arena_bytes": int(arena_b),
                    "utterance_steps_per_s": world * B * a.T / float(fr.item()), "ms_per_step": float(fr.item()) / a.T * 1e3,
                    "what": "Sampler.predict_conditional (bind + initialize_x + all T steps + final sync) of one batch on one stream through the product classes; "
                            "operator_and_batch_setup_s = Tester.prepare_batch (synthetic y = clean * RIR through the HIP FIR, blind operator handle) and the "
                            "replica handle's activation arena (hipMalloc) before it"}
        log(f"full run: {full_run['wall_s']:.2f} s for T={a.T}")
        del tester_f, y_f, op_f, pred_f

    # end-of-run gather of the (B_local, L) outputs: the only collective on this path (RCCL over xGMI)
    gather_ms = gather_first_ms = 0.0
    out = torch.cat([r.x_den for r in runs]).contiguous()
    if dist is not None:
        # twice: the first call carries the communicator's lazy set-up (ring / channel construction), the second is what a long-running
        # harness pays per run; both are outside the timed region
        for k in range(2):
            torch.cuda.synchronize(); dist.barrier(); tg = time.perf_counter()
            out_c = out.to(coll_dev)
            bufs = [torch.empty_like(out_c) for _ in range(world)]
            dist.all_gather(bufs, out_c)
            torch.cuda.synchronize(); g = (time.perf_counter() - tg) * 1e3
            gather_first_ms, gather_ms = (g, g) if k == 0 else (gather_first_ms, g)
        assert torch.isfinite(torch.stack(bufs)).all()
        assert torch.equal(bufs[rank].to(out.device), out), "gather returned another rank's rows in this rank's slot"
        # the harness's own gather (buddy_amd/dist.py: lengths, then zero-padded rows) on the same communicator
        from buddy_amd import dist as bdist
        rows = bdist.gather_ragged([out[i] for i in range(out.shape[0])], world * out.shape[0], rank, world, device=device)
        assert len(rows) == world * out.shape[0] and all(torch.equal(rows[rank + world * i].to(out.device), out[i]) for i in range(out.shape[0]))
    rccl_selftest = None
    if dist is None and world == 1 and not a.no_rccl_selftest:
        # N=1 default run: bring RCCL up once anyway (one rank, this GPU) after everything that is timed, so that communicator set-up, the
        # IPC-mode setting and the nccl side of buddy_amd/dist.py have run on this box before an 8-GPU node appears.  Failure is reported, not fatal.
        rccl_selftest = {"ok": False}
        try:
            import socket
            import torch.distributed as tdist
            from buddy_amd import dist as bdist
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            t0 = time.perf_counter()
            tdist.init_process_group("nccl", device_id=device)
            rccl_selftest["init_ms"] = (time.perf_counter() - t0) * 1e3
            tms = []
            for k in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                full = bdist.gather_rows(out, out.shape[0], 0, 1)
                torch.cuda.synchronize(); tms.append((time.perf_counter() - t0) * 1e3)
            rows = bdist.gather_ragged([out[i, : a.length - 17 * i] for i in range(out.shape[0])], out.shape[0], 0, 1, device=device)
            rccl_selftest.update(ok=bool(torch.equal(full, out) and all(torch.equal(r, out[i, : a.length - 17 * i]) for i, r in enumerate(rows))),
                                 gather_first_call_ms=tms[0], gather_ms=tms[1], backend=tdist.get_backend(), world=1,
                                 what="init_process_group('nccl') on one rank + buddy_amd.dist.gather_rows / gather_ragged on device tensors (all_gather through RCCL)")
            tdist.destroy_process_group()
        except Exception as e:       # noqa: BLE001 -- reported in the line
            rccl_selftest["error"] = f"{type(e).__name__}: {str(e)[:300]}"
    assert torch.isfinite(out).all(), "sampler diverged"

    if rank == 0:
        n_utt_steps = world * B * a.steps
        step_s = elapsed / a.steps                                                           # wall time of one step in the timed region
        attr_step_s = attr_elapsed / ATTR_STEPS                                              # ... in the fully instrumented pass (slower: event bubbles)
        n36 = max(1, int(dom_n.value))
        gemm_tf = dom_fl.value / (dom_ms[1] * 1e-3) / 1e12 if dom_ms[1] > 0 else 0.0         # executed FLOPs of the 36 batched GEMMs / their time
        gemm_ms_per_step = dom_ms[1] / max(1, sampled_steps)
        conv_alg_tf = fl[0] / (ms[0] * 1e-3) / 1e12 if ms[0] > 0 else 0.0                    # direct-convolution FLOPs / three-launch group time
        exec_step_tf = (xf[0] + xf[1]) / ATTR_STEPS / step_s / 1e12                          # every matrix-core FLOP executed per step / wall time of a step
        a36 = max(1, int(w4_n.value))
        # HBM bytes of the dominant kernel: separate rocprofv3 --pmc passes of this same command (counters cannot be read in-process),
        # summarised by tools/pmc_summary.py; quoted only if measured on the kernels that are running now (source stamp)
        gemm_mode = {0: "fp32", 1: "bf16x3", 2: "f16x2"}[int(net0[0].get_option("gemm"))]      # what the handle ran with (--gemm, BUDDY_GEMM, or the library default)
        traffic, traffic_group, traffic_src = None, None, "no PMC summary for the current kernel sources (tools/pmc_summary.py writes profiles/conv_traffic_pmc.json)"
        tp = os.path.join(ROOT, "profiles", "conv_traffic_pmc.json")
        stamp = conv_source_stamp()
        if os.path.exists(tp) and B == 8 and a.length == 64000:
            pj = json.load(open(tp))
            if pj.get("source_stamp") == stamp:
                k36 = [v for k, v in pj["per_kernel_bytes_per_convolution"].items() if "GEMM" in k or "36>" in k][0]
                traffic = k36["fetch"] + k36["write"]
                traffic_group = pj.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/conv_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction; source stamp {stamp})"
            else:
                traffic_src = f"profiles/conv_traffic_pmc.json is stale (stamp {pj.get('source_stamp')} != {stamp} of the current igemm.hip / wino4.hip): not quoted"
        peaks = measure_peaks(lib, device)
        res = {
            "metric": f"diffusion steps/sec ({a.length / 16000:g} s@16 kHz utterance, blind Euler-Heun DPS, order 1, 10 operator updates/step)",
            "value": n_utt_steps / elapsed, "unit": "utterance-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "f32 (bf16x3 exact split in the Winograd-domain GEMMs)", "fp32": "f32",
                      "f16x2": "f32 (GEMM operands -- Winograd-domain and 1x1 / NIN -- as two-term f16 splits, 2^-22, fp32 accumulation)"}[gemm_mode], "data": "synthetic (seeded clean/RIR/weights; random-init NCSN++ 27.7 M params)",
            "config": {"workload": f"blind DPS sampler step, B={B} utterances/GPU x {a.length} samples ({a.length / 16000:g} s@16 kHz), T={a.T}-step schedule, "
                                   f"NCSN++ nf=128 STFT 510/128" + (" (BASELINE.json configs[1])" if (B == 8 and a.length == 64000) else ""),
                       "batch_per_gpu": B, "length": a.length, "T": a.T, "order": 1, "op_updates_per_step": 10,
                       "parallelism": f"utterance-sharded x{world}", "sub_batches_per_gpu": S, "attention": a.attention or os.environ.get("BUDDY_ATTN", "auto (fp32: materialised T x T form for T <= 4096, online-softmax kernels beyond)")},
            "score_evals_per_s": n_utt_steps / elapsed,   # forward + input-VJP evaluations per second (order 1: one per utterance-step); forward-only: legs.forward_only
            "value_mode": ("one batch of B utterances on one stream (per-kernel attribution is clean); the harness default (Tester, groups of >= 4 "
                           "utterances) samples them as two concurrent sub-batches = `concurrent_sub_batches`") if S == 1 else f"{S} concurrent sub-batches",
            "cold_start": {"cold_start_s": SETUP.get("cold_start_s"), "first_step_ms": first_step_ms, "module_build_s": SETUP.get("module_build_s"),
                           "stack_build_s": stack_build_s, "weight_store_bytes": weight_store_main, "weight_store_bytes_after_legs": net0[0].weight_bytes(),
                           "import_torch_s": SETUP.get("import_torch_s"), "hip_context_s": SETUP.get("hip_context_s"), "lib_load_s": SETUP.get("lib_load_s"),
                           "synth_inputs_s": SETUP.get("synth_inputs_s"), "prepare_batch_s": SETUP.get("prepare_batch_s"),
                           "what": "cold_start_s = buddy_ncsnpp_create (parameter upload, small packs) + the first forward of the batch, which prepares on the "
                                   "GPU the one operand form each 3x3 convolution uses (wprep.hip); first_step_ms = the first sampler step (arena dry run, "
                                   "data-gradient forms, operator hipGraph capture); module_build_s = torch-side module construction + synthetic state dict "
                                   "+ .to(device); replicas (concurrent sub-batches, legs, full_run) share the weight store"},
            "legs": legs, "full_run": full_run, "rccl_selftest": rccl_selftest, "dist_init_ms": dist_init_ms,
            "network_algorithmic_tflops": n_utt_steps * 2 * U_FWD * (a.length / 64000.0) / elapsed / 1e12,
            "gather_ms": gather_ms, "gather_first_call_ms": gather_first_ms, "gather_bytes_per_rank": int(out.numel() * 4),
            "gather_backend": (a.backend if dist is not None else None), "per_rank_ms_per_step": per_rank_ms,
            "per_rank_placement": placements if world > 1 else None, "rccl_env": rccl_env,
            # dominant kernel: the batched Winograd-domain GEMMs of the 3x3 convolutions.  bf16x3 (default): every fp32 multiply-add is SIX bf16 MFMA
            # multiply-adds -> achieved = 6 x the fp32-equivalent rate, against the bf16 matrix peak; the fp32-equivalent rate against the fp32 matrix
            # peak is beside it (the kernel replaces v_mfma_f32_32x32x2_f32 at equal accuracy; --gemm fp32 is the reference run)
            "roofline": (f16x2_roofline(gemm_tf, dom_bg.value, dom_ms[1], peaks) if gemm_mode == "f16x2" else
                         {"bound": "mfma", "kernel": "wgemm_bf16x3_kernel -- the batched GEMMs M[pos] = V[pos] U[pos]^T of the three-pass Winograd 3x3 convolutions "
                                                     "(64 positions, F(6x6,3x3), on the large layers; 36, F(4x4,3x3), on the small ones; 94 % of the network's "
                                                     "algorithmic FLOPs) in bf16x3 arithmetic: exact three-way bf16 split of both fp32 operands, six "
                                                     "v_mfma_f32_32x32x16_bf16 products per 16 k, fp32 accumulate",
                          "achieved": 6.0 * gemm_tf, "peak": PEAK_BF16_MFMA, "unit": "TFLOP/s", "frac": 6.0 * gemm_tf / PEAK_BF16_MFMA,
                          "achieved_note": "EXECUTED bf16 MFMA FLOPs per launch (6 x 2 * positions * tiles * Cin * Cout) / average launch duration, HIP events on the "
                                           "launch stream inside the timed region (every launch of every other step)",
                          "peak_measured_on_box": peaks.get("bf16_mfma_tflops"),
                          "frac_of_measured_peak": (6.0 * gemm_tf / peaks["bf16_mfma_tflops"]) if peaks.get("bf16_mfma_tflops") else None,
                          "peak_measured_note": "a pure v_mfma_f32_32x32x16_bf16 loop on random operand bits (no memory traffic) on this box: the chip clocks "
                                                "to its power budget, so this -- not 2.5 PFLOP/s -- is what the matrix pipe sustains here",
                          "fp32_equivalent_tflops": gemm_tf, "frac_of_fp32_matrix_peak": gemm_tf / PEAK_FP32_MFMA,
                          "fp32_equivalent_note": "the same launches counted as the fp32 multiply-adds they replace (2 * positions * tiles * Cin * Cout) against the "
                                                  "157.3 TFLOP/s fp32 matrix peak: what an exact-fp32 GEMM could reach at most on v_mfma_f32_32x32x2_f32",
                          "hbm_side": {"achieved_GBps": dom_bg.value / (dom_ms[1] * 1e-3) / 1e9 if dom_ms[1] > 0 else 0.0, "peak_GBps": PEAK_HBM_GBS,
                                       "frac": (dom_bg.value / (dom_ms[1] * 1e-3) / 1e9 / PEAK_HBM_GBS) if dom_ms[1] > 0 else 0.0,
                                       "note": "algorithmic bytes (V read once, M written once, weights once) / time: the K = 128 layers of level 0 sit nearer this roofline "
                                               "than the matrix one (192 bf16 FLOP per byte)"}}
                         if gemm_mode == "bf16x3" else
                         {"bound": "mfma", "kernel": "igemm_kernel<1,false,false,2,2,36> -- the batched GEMMs M[pos] = V[pos] U[pos]^T of the three-pass "
                                                     "Winograd 3x3 convolutions on v_mfma_f32_32x32x2_f32 (--gemm fp32: the reference run)",
                          "achieved": gemm_tf, "peak": PEAK_FP32_MFMA, "unit": "TFLOP/s", "frac": gemm_tf / PEAK_FP32_MFMA,
                          "achieved_note": "EXECUTED FLOPs per launch (2 * positions * tiles * Cin * Cout) / average launch duration, HIP events on the launch stream inside "
                                           "the timed region (every launch of every other step: an event pair costs a ~7 us dispatch bubble); <= 1 by construction",
                          "peak_measured_on_box": peaks.get("fp32_mfma_tflops"),
                          "frac_of_measured_peak": (gemm_tf / peaks["fp32_mfma_tflops"]) if peaks.get("fp32_mfma_tflops") else None}),
            # the whole 3x3 convolution (three launches) and the whole step, for context
            "conv3x3": {"algorithmic_tflops": conv_alg_tf, "algorithmic_speedup": (fl[0] / xf[0]) if xf[0] > 0 else None,
                        "note": "direct-convolution FLOPs (2*M*N*9*Cin) / time of the three-launch group; the Winograd forms execute 64/(36*9) "
                                "(F(6x6,3x3), plus tile overhang) or 1/4 (F(4x4,3x3)) of them -- algorithmic_speedup = direct / executed FLOPs over all "
                                "convolutions -- so this is NOT a roofline fraction: the matrix-pipe utilisation is roofline.frac",
                        "avg_conv_ms": ms[0] / max(1, ln[0]), "convolutions": int(ln[0]), "share_of_step": ms[0] * 1e-3 / attr_elapsed,
                        "transform_passes": {"input_GBps": w4_bi.value / (w4_ms[0] * 1e-3) / 1e9 if w4_ms[0] > 0 else 0.0,
                                             "output_GBps": w4_bo.value / (w4_ms[2] * 1e-3) / 1e9 if w4_ms[2] > 0 else 0.0,
                                             "peak_GBps": PEAK_HBM_GBS, "copy_GBps_on_this_box": peaks.get("hbm_copy_GBps"), "copy_GBps_guide": GUIDE_COPY_GBS,
                                             "share_of_step": (w4_ms[0] + w4_ms[2]) * 1e-3 / attr_elapsed,
                                             "note": "HBM-bound: input read once + 64/36 (F(6x6,3x3)) or 36/16 (F(4x4,3x3)) transformed values written; the same read + output (and residual) once"},
                        "fused_form_bytes_per_conv": by[0] / max(1, ln[0]),
                        "three_pass_bytes_per_conv": (w4_bi.value + w4_bg.value + w4_bo.value) / a36},
            "step_executed": {"tflops": exec_step_tf, "frac": exec_step_tf / PEAK_FP32_MFMA,
                              "note": "all matrix-core FLOPs executed per step (Winograd-domain GEMMs, 1x1 / attention / DFT GEMMs; counted in the attribution "
                                      "pass) / wall time of a step of the timed region / fp32 matrix peak"},
            "other_matrix_kernels": {"tflops": fl[1] / (ms[1] * 1e-3) / 1e12 if ms[1] > 0 else 0.0, "ms_per_step": ms[1] / ATTR_STEPS, "launches": int(ln[1])},
            # second roofline SURVEY 8(d) asks for: the HBM-bound GroupNorm(+SiLU, +2x resample) kernels and their backward
            "roofline_hbm": {"bound": "hbm", "kernel": "GroupNorm statistics / apply(+SiLU,+resample) / backward (chan_reduce, gn_apply, gn_bwd_apply)",
                             "achieved": hb_by.value / (hb_ms.value * 1e-3) / 1e9 if hb_ms.value > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": (hb_by.value / (hb_ms.value * 1e-3) / 1e9 / PEAK_HBM_GBS) if hb_ms.value > 0 else 0.0,
                             "frac_of_measured_copy": (hb_by.value / (hb_ms.value * 1e-3) / 1e9 / peaks["hbm_copy_GBps"]) if (hb_ms.value > 0 and peaks.get("hbm_copy_GBps")) else None,
                             "frac_of_guide_copy": (hb_by.value / (hb_ms.value * 1e-3) / 1e9 / GUIDE_COPY_GBS) if hb_ms.value > 0 else None,
                             "peak_measured_on_box": {k: v for k, v in peaks.items() if k.startswith("hbm_")},
                             "launch_groups": int(hb_n.value), "kernel_time_share_of_step": hb_ms.value * 1e-3 / attr_elapsed,
                             "note": "algorithmic bytes (every pass reads its inputs and writes its output once) / HIP-event time"},
            "operator_update": {"ms_per_step": op_ms / ATTR_STEPS, "share_of_step": op_ms * 1e-3 / attr_elapsed,
                                "what": "optimize_op: 10 x (design filter, min-phase projection, subband FIR, loss, analytic backward, Adam, clamps) per step, "
                                        "HIP events on the launch stream (rank 0)"},
            "attribution_pass": {"steps": ATTR_STEPS, "ms_per_step": attr_step_s * 1e3,
                                 "note": "conv3x3 / step_executed FLOP counts / other_matrix_kernels / roofline_hbm / operator_update come from these fully instrumented "
                                         "steps run right after the timed region (shares are of THIS pass's time); value, ms_per_step and roofline from the timed region"},
            "peaks": {"nominal": {"fp32_mfma_tflops": PEAK_FP32_MFMA, "bf16_mfma_tflops": PEAK_BF16_MFMA, "hbm_GBps": PEAK_HBM_GBS}, "measured_on_this_box": peaks},
        }
        res["roofline"].update({"avg_launch_ms": dom_ms[1] / n36, "launches": n36, "sampled_steps": sampled_steps, "share_of_step": gemm_ms_per_step * 1e-3 / step_s,
                                "flops_per_launch_fp32_equivalent": dom_fl.value / n36, "algorithmic_bytes_per_launch": dom_bg.value / n36,
                                "traffic": traffic, "traffic_source": traffic_src,
                                # the whole three-launch convolution against the bytes a fused form would move: what the three-pass structure costs
                                "traffic_conv_group": traffic_group, "fused_form_bytes_per_conv": by[0] / max(1, ln[0]),
                                "traffic_conv_group_over_fused_form": (traffic_group / (by[0] / max(1, ln[0]))) if (traffic_group and ln[0]) else None})
        res["config"]["gemm"] = gemm_mode
        if conc is not None:
            res["concurrent_sub_batches"] = conc
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle on host cores)")
            res["cpu_baseline"] = run_cpu_baseline(a.length)
        C.CDLL(None).fflush(None)          # anything buffered by C stdio (the RCCL banner) leaves through stderr before the line is written
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
