"""World-size-2 gloo test (CPU) of the utterance-sharded path: shard u -> rank u mod 2, sample independently, one
all_gather at the end; the gathered result must equal the single-process run (results independent of world size)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _sample(utts, L):
    """blind DPS (toy score function, batched operator) for the given utterance ids -> (len(utts), L)"""
    sys.path.insert(0, ROOT)
    from buddy_amd.config import compose
    from buddy_amd.instantiate import instantiate
    from oracle.batched.operators import BlindSubbandFiltering
    from oracle.batched.sampler import EulerHeunSamplerDPSTorch
    from oracle.sampler_ref import NoiseStream
    from tests.test_host_logic import _ToyNet
    torch.set_num_threads(2)
    args = compose(overrides=["tester.sampling_params.T=2", "tester.posterior_sampling.warm_initialization.mode=reverb_scaled",
                              "tester.posterior_sampling.blind_hp.op_updates_per_step=1"])
    smp = EulerHeunSamplerDPSTorch(_ToyNet(), instantiate(args.diff_params), args)
    ns = [NoiseStream(900 + u) for u in utts]
    smp.noise = ns
    y = torch.stack([torch.from_numpy((0.05 * np.random.RandomState(u).standard_normal(L)).astype(np.float32)) for u in utts])
    op = BlindSubbandFiltering(args.tester.informed_dereverberation.op_hp, 16000, num_utts=len(utts), noise=ns, device="cpu")
    op.update_H(use_noise=True)
    return smp.predict_conditional(y, op, shape=(len(utts), L), blind=True)


def _worker(rank, world, port, n_utts, L, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from buddy_amd import dist as bd
    r, _, w = bd.init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = bd.shard_indices(n_utts, rank, world)
    local = _sample(mine, L)
    full = bd.gather_rows(local, n_utts, rank, world)
    if rank == 0:
        torch.save(full, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    n_utts, L = 3, 4096        # ragged shards: rank 0 gets utterances {0, 2}, rank 1 gets {1}
    out_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_utts, L, out_path), nprocs=2, join=True)
    gathered = torch.load(out_path)
    single = torch.cat([_sample([u], L) for u in range(n_utts)])
    assert gathered.shape == (n_utts, L)
    err = float((gathered - single).abs().max() / single.abs().max())
    assert err < 1e-4, err


def test_eight_ranks_64_utterance_partition(tmp_path):
    """BASELINE configs[2]'s partition at its real rank count on the CPU: 64 utterances over eight gloo ranks (u -> rank u mod 8, eight per rank,
    sampled as ONE batch of eight per rank like a GPU rank does), one all_gather; every row equals the utterance sampled alone (per-utterance
    semantics: results do not depend on the world size or on which utterances share a batch)."""
    n_utts, L = 64, 2048
    out_path = str(tmp_path / "gathered8.pt")
    mp.spawn(_worker, args=(8, _free_port(), n_utts, L, out_path), nprocs=8, join=True)
    gathered = torch.load(out_path)
    assert gathered.shape == (n_utts, L) and torch.isfinite(gathered).all()
    for u in (0, 7, 8, 13, 42, 63):        # spot rows of different ranks / batch positions against the utterance on its own
        single = _sample([u], L)[0]
        err = float((gathered[u] - single).abs().max() / single.abs().max())
        assert err < 1e-4, (u, err)


def _ragged_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from buddy_amd import dist as bd
    bd.init(backend="gloo")
    lengths = [5000, 4096, 7001, 300, 4096]
    mine = bd.shard_indices(len(lengths), rank, world)
    rows = [torch.full((lengths[i],), float(i + 1)) + torch.arange(lengths[i]) * 1e-4 for i in mine]
    full = bd.gather_ragged(rows, len(lengths), rank, world)
    ok = all(full[i].shape[0] == lengths[i] and torch.equal(full[i], torch.full((lengths[i],), float(i + 1)) + torch.arange(lengths[i]) * 1e-4)
             for i in range(len(lengths)))
    torch.save(ok, out_path + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_gather_ragged_rows_two_ranks(tmp_path):
    """the harness's end-of-run gather: utterances of different lengths, uneven shards, every rank ends up with all rows in order"""
    out_path = str(tmp_path / "ok")
    mp.spawn(_ragged_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    assert torch.load(out_path + ".0") is True and torch.load(out_path + ".1") is True


def _fake_sysfs(root, cards):
    """cards: list of (vendor, local_cpulist or None, numa_node or None) in drm card order; node cpulists: node0 = 0-7, node1 = 8-15"""
    for i, (vendor, cpulist, node) in enumerate(cards):
        d = root / "class" / "drm" / f"card{i}" / "device"
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        if cpulist is not None:
            (d / "local_cpulist").write_text(cpulist + "\n")
        if node is not None:
            (d / "numa_node").write_text(f"{node}\n")
    for n, cl in ((0, "0-7"), (1, "8-15")):
        nd = root / "devices" / "system" / "node" / f"node{n}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    (root / "class" / "drm" / "card0-DP-1").mkdir()          # connector entries sit next to the cards in /sys/class/drm
    return str(root)


def test_rank_placement_device_mapping_and_affinity(tmp_path, monkeypatch):
    """Round 6: what a rank needs on an 8-GPU node before it samples -- its device under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    re-mapping, and its own cores next to its GPU (sysfs numa_node / local_cpulist), disjoint from the other ranks' (each rank drives ~2 000 launches per
    step from one Python thread).  Pure host logic on a fabricated /sys tree: two sockets x four GPUs, an integrated non-AMD card in between."""
    sys.path.insert(0, ROOT)
    from buddy_amd import dist as bd
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert bd.visible_device_ids() is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert bd.visible_device_ids() == [4, 5, 6, 7]
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "7,6,5,4,3,2,1,0")           # HIP indices count inside the ROCr list
    assert bd.visible_device_ids() == [3, 2, 1, 0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert bd.visible_device_ids() == [7, 6, 5, 4, 3, 2, 1, 0]
    assert [bd.device_index(r, 8) for r in range(8)] == list(range(8)) and [bd.device_index(r, 1) for r in range(3)] == [0, 0, 0]
    # plan: disjoint, inside the allowed set, GPU-local when the kernel says which cores those are
    allowed = list(range(16))
    sets = [bd.plan_affinity(r, 4, allowed, list(range(8, 16))) for r in range(4)]
    assert all(set(s) <= set(range(8, 16)) and len(s) == 2 for s in sets) and len(set().union(*map(set, sets))) == 8
    sets = [bd.plan_affinity(r, 8, allowed, None) for r in range(8)]
    assert len(set().union(*map(set, sets))) == 16 and all(len(s) == 2 for s in sets)
    assert bd.plan_affinity(0, 8, [3, 5], [3, 5, 7]) == [3] and bd.plan_affinity(1, 8, [3, 5], [0, 1]) in ([5], [3, 5])   # fewer cores than ranks: still a valid set
    # sysfs: AMD cards only, in card order; local_cpulist first, numa_node as the fall-back
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    sysfs = _fake_sysfs(tmp_path, [("0x1002", "0-7", 0), ("0x1002", None, 0), ("0x8086", "0-15", 0), ("0x1002", "0-7", 0), ("0x1002", "0-7", 0),
                                   ("0x1002", "8-15", 1), ("0x1002", None, 1), ("0x1002", "8-15", 1), ("0x1002", None, -1)])
    assert bd._gpu_numa_cpus(0, sysfs) == list(range(8)) and bd._gpu_numa_cpus(1, sysfs) == list(range(8))        # node fall-back
    assert bd._gpu_numa_cpus(4, sysfs) == list(range(8, 16)) and bd._gpu_numa_cpus(7, sysfs) is None and bd._gpu_numa_cpus(99, sysfs) is None
    applied = []
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(16)), raising=False)
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cpus: applied.append(sorted(cpus)), raising=False)
    recs = [bd.pin_rank(r, 8, 8, sysfs) for r in range(8)]
    assert [r["physical_device"] for r in recs] == list(range(8)) and applied == [r["cpus"] for r in recs]
    for r in range(4):
        assert set(recs[r]["cpus"]) <= set(range(8)) and recs[r]["numa_cpus"] == 8
    for r in (4, 5, 6):
        assert set(recs[r]["cpus"]) <= set(range(8, 16))
    for grp in (recs[:4], recs[4:7]):
        allc = [c for r in grp for c in r["cpus"]]
        assert len(allc) == len(set(allc)), "ranks of one socket share cores"
    assert recs[7]["numa_cpus"] is None and recs[7]["cpus"]                                  # a GPU without a NUMA node: a slice of the allowed set
    # eight ranks sharing ONE visible GPU (the gloo smoke test of the test box): eight disjoint slices of that GPU's cores
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4")
    recs = [bd.pin_rank(r, 8, 1, sysfs) for r in range(8)]
    assert all(r["physical_device"] == 4 for r in recs) and sorted(c for r in recs for c in r["cpus"]) == list(range(8, 16))
    env = bd.rccl_env_defaults()
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] and os.environ["NCCL_DEBUG"]
