"""GPU: the reference command line end to end (test.py drop-in): paired clean/RIR wav tree -> harness -> sampler -> wav tree
(reference testing/tester.py:155-203), blind mode with the shipped `wpe_scaled` warm start and informed mode, 2 steps, small net."""
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

pytestmark = pytest.mark.gpu


def _dataset(root, n=2, L=16000):
    from buddy_amd.synth import synth_clean, synth_rir
    for u in range(n):
        for sub, data in (("clean", synth_clean(u, L)), ("rir", np.concatenate([np.zeros(37, np.float32), synth_rir(u, 3000)]))):
            d = os.path.join(root, sub, "p001")
            os.makedirs(d, exist_ok=True)
            wavfile.write(os.path.join(d, f"p001_{u:03d}.wav"), 16000, data.astype(np.float32))


@pytest.mark.parametrize("tester,mode", [("blind_dereverberation_BUDDy", "blind_dereverberation"), ("informed_dereverberation_DPS", "informed_dereverberation")])
def test_cli_end_to_end(tmp_path, tester, mode):
    import test as cli
    data = str(tmp_path / "data")
    _dataset(data)
    out = str(tmp_path / "exp")
    cli.main(["--config-name=conf_VCTK.yaml", f"tester={tester}", "tester.sampling_params.T=2", f"model_dir={out}", "+gpu=0",
              f"dset.test.path={data}", "dset.test.num_examples=2", "network.nf=32", "+batch_size=2", "tester.overriden_name=run", "+allow_random_init=true"])
    base = os.path.join(out, "run", mode, "VCTK_16k_4s_time")
    subs = ["original", "degraded", "reconstructed", "true_rir"] + (["estimated_rir"] if "blind" in mode else [])
    for s in subs:
        files = sorted(os.listdir(os.path.join(base, s)))
        assert len(files) == 2, (s, files)
        sr, a = wavfile.read(os.path.join(base, s, files[0]))
        assert sr == 16000 and np.isfinite(a).all() and np.abs(a).max() > 0
    assert os.path.exists(os.path.join(base, ".argv"))
    # RIR preprocessing of the paired loader: direct path first, peak-normalised (reference datasets/vctk.py:211-214)
    _, r = wavfile.read(os.path.join(base, "true_rir", sorted(os.listdir(os.path.join(base, "true_rir")))[0]))
    assert abs(np.abs(r).max() - 1.0) < 1e-6 and abs(r[0]) == np.abs(r).max()


def test_harness_ragged_utterance_lengths(tmp_path):
    """Utterances of different lengths (one not a multiple of the hop) in one batch: the harness groups equal lengths, every output keeps
    its own length, and each result equals the single-utterance run with the same per-utterance noise stream."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    args = compose(tester="informed_dereverberation_DPS", overrides=["tester.sampling_params.T=2", "network.nf=32"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(3, 32).items()})
    net = net.cuda().eval()
    edm = instantiate(args.diff_params)
    lengths = [16000, 12345, 16000]
    items = [(synth_clean(u, L), synth_rir(u, 2000), f"u{u}.wav") for u, L in enumerate(lengths)]

    def run(sel):
        t = Tester(args, net, edm, test_set=[items[i] for i in sel], device="cuda", in_training=True)
        t.batch_size = len(sel)
        t.noise_factory = lambda names: [NoiseStream(100 + int(n[1:-4])) for n in names]
        t.test_dereverberation("informed_dereverberation", blind=False)
        return {n: p for n, p in t.results}

    allr = run([0, 1, 2])
    assert {k: v.shape[-1] for k, v in allr.items()} == {"u0": 16000, "u1": 12345, "u2": 16000}
    for i in range(3):
        single = run([i])[f"u{i}"]
        a, b = allr[f"u{i}"].double(), single.double()
        assert float((a - b).abs().max() / b.abs().max()) < 1e-3      # same bound as the batched-vs-single sampler test (guidance normalisation amplifies round-off)
        assert torch.isfinite(a).all()


def test_bench_gpus2_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it must become two ranks itself (the driver's own command for the scaling curve),
    shard the utterances, time the end-of-run gather and print ONE line with n_gpus == 2.  One GPU here: gloo lets the two ranks share it
    (RCCL needs one device per rank); the N-rank RCCL path differs only in the backend string."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "2", "--also-concurrent", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["batch_per_gpu"] == 2
    assert j["gather_ms"] > 0 and j["gather_bytes_per_rank"] == 2 * 64000 * 4 and j["gather_backend"] == "gloo"
    assert abs(j["value"] - 2 * 2 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    # a launcher / --gpus mismatch is refused instead of silently printing n_gpus = 1
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_gpus8_gloo_rank_placement():
    """`python bench.py --gpus 8 --backend gloo` on the one GPU of the test box (round 6): the one-command 8-rank run of the day a node
    exists, minus RCCL (one device).  Every rank pins itself to its own cores next to its GPU (buddy_amd.dist.pin_rank) -- eight pairwise DISJOINT sets
    when the lease has at least eight cores --, the line carries each rank's placement and step time, and the ranks' step times agree within 5 %
    (they are timed between the same barriers)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "1", "--also-concurrent", "0", "--no-cpu-baseline", "--legs", "none"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["batch_per_gpu"] == 1 and j["gather_backend"] == "gloo" and len(j["per_rank_ms_per_step"]) == 8
    pl = j["per_rank_placement"]
    assert len(pl) == 8 and all(p["cpus"] for p in pl)
    n_cores = len(os.sched_getaffinity(0))
    if n_cores >= 8:
        allc = [c for p in pl for c in p["cpus"]]
        assert len(allc) == len(set(allc)), f"ranks share cores: {[p['cpus'] for p in pl]}"
    t = j["per_rank_ms_per_step"]
    assert (max(t) - min(t)) / max(t) < 0.05, t
    assert "rank 7: physical device" in r.stderr and "rccl / device environment" in r.stderr and j["rccl_env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_longform_chunked_blind_keeps_the_level_profile():
    """blind chunked sampling rescales every chunk to one standard deviation (constraint_speech_magnitude acts per utterance = per chunk), which
    would bring a pause back as loud as speech.  Tester.dereverberate_long(blind=True) level-matches the chunks to the observation before the cross-fade:
    on a clip whose middle third is 26 dB quieter, the estimate's pause-to-speech level ratio follows the input's (small network, 3 steps)."""
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    args = compose(tester="blind_dereverberation_BUDDy", overrides=["tester.sampling_params.T=3", "network.nf=32",
                                                                      "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                                                                      "tester.posterior_sampling.blind_hp.op_updates_per_step=2"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(3, 32).items()})
    net = net.cuda().eval()
    t = Tester(args, net, instantiate(args.diff_params), test_set=None, device="cuda", in_training=True)
    L = 96000
    clean = synth_clean(2, L).copy()
    clean[32000:64000] *= 0.05
    seg, y, pred = t.dereverberate_long(clean, synth_rir(2, 2000), blind=True, chunk_seconds=2.0, overlap_seconds=0.25)
    assert pred.shape == (L,) and torch.isfinite(pred).all()
    ratio = lambda z: float(z[36000:60000].std() / (z[:28000].std() + 1e-12))
    ry, rp = ratio(y), ratio(pred)
    print(f"pause / speech level: observation {ry:.3f}, blind chunked estimate {rp:.3f}")
    assert rp < 3.0 * ry + 0.05          # without the level match every chunk comes back at std 0.05: ratio ~ 1


def test_bench_line_contract():
    """The ONE JSON line `python bench.py` prints (driver contract): metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better /
    scaling / vs_baseline / dtype / data / config.workload naming BASELINE.json configs[1], and the `roofline` object of the dominant kernel with
    bound / achieved / peak / unit / frac / traffic; value consistent with ms_per_step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--also-concurrent", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert "configs[1]" in j["config"]["workload"] and j["config"]["batch_per_gpu"] == 8 and j["config"]["gemm"] in ("f16x2", "bf16x3", "fp32")
    assert abs(j["value"] - 8 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches"):
        assert k in rf, k
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] == {"mfma": "TFLOP/s", "hbm": "GB/s"}[rf["bound"]] and 0.0 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["peak"] == {"mfma": 2500.0, "hbm": 8000.0}[rf["bound"]]
    other = rf["mfma_side" if rf["bound"] == "hbm" else "hbm_side"]            # the other roofline of the same launches is beside it
    assert 0.0 < other["frac"] <= rf["frac"]
    assert rf["launches"] == 80                      # all 80 launches of the dominant kernel on the one sampled step
    if rf["traffic"] is not None:                    # quoted only when the stamped PMC summary matches the kernel sources
        assert 0.5e9 < rf["traffic"] < 3e9 and rf["traffic_conv_group_over_fused_form"] > 1.0
    # round 4: cold start, the BASELINE.md section 3 legs, one full run, the one-rank RCCL bring-up
    cs = j["cold_start"]
    assert 0.0 < cs["cold_start_s"] < 1.5 and 0.0 < cs["first_step_ms"] < 3000.0, cs
    ws = cs["weight_store_bytes"]
    assert ws["conv3_forms"] < 3.2e9 and ws["params"] < 1.2e8, ws          # shared by every replica (round 3: 5.7 GB per handle); the sub-pixel up forms carry 4 phase kernels each
    assert "one stream" in j["value_mode"]
    # round 6: bandwidth calibrated with the library's own float4 streaming kernel (a read-only stream is not slower than a copy),
    # every HBM-bound fraction quoted against 8.0 TB/s nominal, the box's copy rate and the guide's 6.29 TB/s; the exact-arithmetic legs and the real-clip
    # geometry in the default line
    pk = j["peaks"]["measured_on_this_box"]
    assert pk["hbm_read_GBps"] > pk["hbm_copy_GBps"] > 3000.0 and pk["hbm_write_GBps"] > 3000.0 and pk["hbm_copy_guide_GBps"] == 6290.0, pk
    if rf["bound"] == "hbm":
        assert abs(rf["frac_of_measured_peak"] - rf["achieved"] / pk["hbm_copy_GBps"]) < 1e-9 and abs(rf["frac_of_guide_copy"] - rf["achieved"] / 6290.0) < 1e-9
    assert j["roofline_hbm"]["frac_of_measured_copy"] > j["roofline_hbm"]["frac"]
    for name in ("gemm_bf16x3", "gemm_fp32"):
        assert j["legs"][name]["gemm"] == name[5:] and j["legs"][name]["ms_per_step"] > j["ms_per_step"] * 0.98, name     # the exact forms cost more than f16x2
    assert j["legs"]["gemm_fp32"]["ms_per_step"] > j["legs"]["gemm_bf16x3"]["ms_per_step"]
    assert j["legs"]["informed_order2_B1_L133829"]["length"] == 133829
    assert "leg gemm_bf16x3" in r.stderr and "leg gemm_fp32" in r.stderr and "leg informed_order2_B1_L133829" in r.stderr
    for name in ("informed_order2", "informed_order2_B1", "blind_B1", "blind_B1_flash", "forward_only", "longform_480000_B4", "longform_480000_B4_f16",
                 "gemm_bf16x3", "gemm_fp32", "informed_order2_B1_L133829"):
        leg = j["legs"][name]
        assert leg["ms_per_step"] > 0 and leg["value"] > 0 and leg["unit"] == "utterance-steps/s" and leg["config"], name
    assert j["legs"]["longform_480000_B4"]["attention"]
    assert j["legs"]["forward_only"]["forward_evals_per_s"] > j["score_evals_per_s"]          # a forward costs less than forward + VJP + operator update
    # round 5: the reference's own shape -- ONE utterance through the shipped yaml (wpe_scaled, T = 201, 10 updates), torch RNG --
    # and every set-up piece of the cold start, in the line and (for the driver's tail) on stderr
    sh = j["legs"]["full_run_B1_shipped"]
    assert sh["T"] == 201 and 0.5 < sh["wall_s"] < 20.0 and abs(sh["ms_per_step"] - sh["wall_s"] / 201 * 1e3) < 1e-6 and "UNCHANGED" in sh["config"]
    for k in ("import_torch_s", "hip_context_s", "lib_load_s", "module_build_s", "cold_start_s", "synth_inputs_s", "prepare_batch_s", "stack_build_s"):
        assert cs[k] is not None and cs[k] >= 0.0, k
    assert "stack ready: import_torch_s" in r.stderr and "leg full_run_B1_shipped" in r.stderr
    fr = j["full_run"]
    assert fr["T"] == 50 and fr["batch_per_gpu"] == 8 and 1.0 < fr["wall_s"] < 30.0 and abs(fr["utterance_steps_per_s"] - 400 / fr["wall_s"]) < 1e-6 * 400
    rs = j["rccl_selftest"]
    assert rs["ok"] is True and rs["backend"] == "nccl" and rs["init_ms"] > 0 and rs["gather_first_call_ms"] > 0, rs


def test_bench_force_dist_one_rank_rccl():
    """`bench.py --gpus 1 --force-dist`: the N>1 code path (process group, barriers, all_reduce of the elapsed time, all_gather of the outputs,
    buddy_amd.dist.gather_ragged) on ONE rank with backend nccl = RCCL, so the distributed branch has run on RCCL before an 8-GPU node appears."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--also-concurrent", "0", "--no-cpu-baseline", "--legs", "none"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["gather_backend"] == "nccl" and j["gather_first_call_ms"] > 0 and j["gather_ms"] > 0 and j["dist_init_ms"] > 0
    assert j["rccl_selftest"] is None and j["full_run"] is None and j["legs"] == {}
