"""GPU: utterance-sharded data parallelism through the harness (Tester + dist.gather_ragged) with TWO ranks sharing the one GPU of the
test box (gloo control plane, HIP compute): after the end-of-run gather rank 0 holds every prediction, bit-identical to the single-process
run with the same injected noise streams (results are independent of the world size)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENGTHS = [8192, 6000, 8192]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(rank, world, lengths=None):
    lengths = LENGTHS if lengths is None else lengths
    sys.path.insert(0, ROOT)
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    args = compose(overrides=["tester.sampling_params.T=2", "network.nf=32", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                              "tester.posterior_sampling.blind_hp.op_updates_per_step=2"])
    net = instantiate(args.network)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(2, 32).items()})
    net = net.cuda().eval()
    items = [(synth_clean(u, L), synth_rir(u, 1500), f"u{u}.wav") for u, L in enumerate(lengths)]
    t = Tester(args, net, instantiate(args.diff_params), test_set=items, device="cuda", in_training=True, batch_size=1, rank=rank, world_size=world)
    t.noise_factory = lambda names: [NoiseStream(700 + int(n[1:-4])) for n in names]
    t.test_dereverberation("blind_dereverberation", blind=True)
    return t.gathered


def _worker(rank, world, port, out_path, lengths=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from buddy_amd import dist as bd
    bd.init(backend="gloo")
    g = _run(rank, world, lengths)
    if rank == 0:
        torch.save(g, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    out_path = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path)
    ref = _run(0, 1)
    assert [n for n, _ in got] == [n for n, _ in ref] == ["u0", "u1", "u2"]
    for (n, a), (_, b), L in zip(got, ref, LENGTHS):
        assert a.shape == (L,) and torch.isfinite(a).all()
        assert torch.equal(a, b), (n, float((a - b).abs().max()))


def test_eight_ranks_64_utterances_one_gpu_equal_single_process(tmp_path):
    """BASELINE configs[2] at its real rank count: 64 utterances of ragged lengths over EIGHT ranks (u -> rank u mod 8, eight per rank) sharing
    the one GPU of the test box (gloo control plane; RCCL wants a device per rank), one end-of-run gather: rank 0 holds all 64 rows in utterance
    order, bit-identical to the single-process run.  nf = 32 keeps eight activation arenas small; the partition, the ragged gather and the
    per-utterance noise streams are the full-size ones."""
    lengths = [4096 + 512 * ((5 * u) % 9) + 37 * (u % 4) for u in range(64)]          # 4096 .. 8303 samples, 33 distinct lengths
    out_path = str(tmp_path / "g8.pt")
    mp.spawn(_worker, args=(8, _free_port(), out_path, lengths), nprocs=8, join=True)
    got = torch.load(out_path)
    ref = _run(0, 1, lengths)
    assert [n for n, _ in got] == [n for n, _ in ref] == [f"u{u}" for u in range(64)]
    for (n, a), (_, b), L in zip(got, ref, lengths):
        assert a.shape == (L,) and torch.isfinite(a).all()
        assert torch.equal(a, b), (n, float((a - b).abs().max()))


def test_bench_eight_gloo_ranks_on_one_gpu():
    """`python bench.py --gpus 8 --backend gloo ...`: the driver's N = 8 launch path (self-launcher, RANK / WORLD_SIZE environment, barrier + max over
    ranks timing, end-of-run gather) with eight ranks sharing the one GPU: ONE JSON line, n_gpus 8, the per-rank step times it is the max of."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--batch", "1", "--length", "16000", "--steps", "2", "--warmup", "1",
           "--legs", "none", "--no-cpu-baseline", "--also-concurrent", "0", "--no-rccl-selftest"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    per = j["per_rank_ms_per_step"]
    assert len(per) == 8 and abs(max(per) - j["ms_per_step"]) < 1e-6 * j["ms_per_step"] + 1e-9
    assert abs(j["value"] - 8 * 1 * 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    print(f"8 gloo ranks on one GPU: {j['ms_per_step']:.1f} ms/step (max), per rank {[round(v, 1) for v in per]}")


def test_concurrent_sub_batches_equal_single_batch():
    """tester.sub_batches = 2 (testing/concurrent.py: two sub-batches on two HIP streams, own network replica each) returns what the single-batch
    run returns -- rows never interact; a different batch size only changes tile shapes / reduction chunking of a few kernels, i.e. fp32
    round-off.  Informed mode (no operator optimisation, not chaotic): 1e-4 of the peak.  Blind mode: the six scale-free Adam updates of this
    short run amplify that round-off (oracle/precision.py analysis: 3e-2 at iteration 6), so only 5e-2 is asserted there."""
    sys.path.insert(0, ROOT)
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from buddy_amd.synth import synth_state_dict, synth_clean, synth_rir
    from buddy_amd.testing.tester import Tester
    from oracle.sampler_ref import NoiseStream
    L = 8192
    items = [(synth_clean(u, L), synth_rir(u, 1500), f"u{u}.wav") for u in range(4)]

    def run(sub, blind):
        ov = ["tester.sampling_params.T=3", "network.nf=32", f"+tester.sub_batches={sub}"]
        if blind:
            ov += ["tester.posterior_sampling.warm_initialization.mode=reverb_scaled", "tester.posterior_sampling.blind_hp.op_updates_per_step=2"]
        args = compose(tester="blind_dereverberation_BUDDy" if blind else "informed_dereverberation_DPS", overrides=ov)
        net = instantiate(args.network)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(2, 32).items()})
        net = net.cuda().eval()
        t = Tester(args, net, instantiate(args.diff_params), test_set=items, device="cuda", in_training=True, batch_size=4)
        assert t.sub_batches == sub
        t.noise_factory = lambda names: [NoiseStream(800 + int(n[1:-4])) for n in names]
        t.test_dereverberation("blind_dereverberation" if blind else "informed_dereverberation", blind=blind)
        torch.cuda.synchronize()
        assert (t._concurrent is not None) == (sub > 1)
        return t.gathered

    for blind, tol in ((False, 1e-4), (True, 5e-2)):
        one, two = run(1, blind), run(2, blind)
        for (n1, a), (n2, b) in zip(one, two):
            assert n1 == n2 and torch.isfinite(a).all()
            err = float((a.double() - b.double()).abs().max() / b.double().abs().max())
            print("blind" if blind else "informed", n1, f"{err:.2e}")
            assert err < tol, (blind, n1, err)


def _rccl_one_rank(port, out_path):
    """child process: RCCL (backend "nccl") with ONE rank on the one GPU -- communicator set-up, the dmabuf IPC mode, and the nccl branch of
    buddy_amd/dist.py (device tensors through all_gather) -- which no multi-rank test on a 1-GPU box can take (RCCL wants one device per rank)."""
    import json
    import sys
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from buddy_amd import dist as bd
    torch.cuda.set_device(0)
    t0 = time.time()
    r, _, w = bd.init(backend="nccl", device=torch.device("cuda", 0), force=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    rows = torch.arange(3 * 1000, dtype=torch.float32, device="cuda").reshape(3, 1000)
    full = bd.gather_rows(rows, 3, 0, 1)                       # all_gather on device tensors through RCCL
    rag = bd.gather_ragged([rows[0], rows[1, :777], rows[2, :5]], 3, 0, 1, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    ok = bool(torch.equal(full, rows)) and [int(v.shape[0]) for v in rag] == [1000, 777, 5] and all(v.is_cuda for v in rag) \
        and bool(torch.equal(rag[1], rows[1, :777]))
    dist.barrier()
    dist.destroy_process_group()
    json.dump({"ok": ok, "seconds": time.time() - t0}, open(out_path, "w"))


def test_rccl_one_rank_gathers_on_device(tmp_path):
    """`init_process_group("nccl")` has run at least once on this code before an 8-GPU node appears."""
    import json
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl.json")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_rccl_one_rank, args=(port, out))
    p.start(); p.join(300)
    assert p.exitcode == 0, f"RCCL one-rank child failed (exit {p.exitcode})"
    j = json.load(open(out))
    assert j["ok"], j
    print(f"RCCL one-rank init + gathers: {j['seconds']:.2f} s")
